"""Shared parity machinery: run a refcases case through any StageAPI (oracle or HIP) and compare with a golden."""
import os

import numpy as np

from linrad_amd import abi
from refcases import case_params, lrh_config

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

RINGS = [(abi.RING_FFT1_FLOAT, "fft1_float"), (abi.RING_FFT1_SUMSQ, "fft1_sumsq"),
         (abi.RING_FFT1_SLOWSUM, "fft1_slowsum"), (abi.RING_TIMF2_FLOAT, "timf2_float"),
         (abi.RING_TIMF2_PWR, "timf2_pwr_float"), (abi.RING_FFT2_FLOAT, "fft2_float"),
         (abi.RING_FFT2_POWER, "fft2_power_float"), (abi.RING_FFT2_POWERSUM, "fft2_powersum_float"),
         (abi.RING_TIMF3_FLOAT, "timf3_float")]


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))


def run_case(open_fn, name, golden=None, stupid=None, batch=1, params=None, **cfg_kw):
    """Drive the case block by block in the reference call pattern (ref_harness.c main loop).
    params: a case dictionary of refcases' form instead of the named one (tests/test_gpu_random_configs.py; `golden` then carries iq and liminfo)."""
    d = dict(params) if params is not None else case_params(name)
    g = golden if golden is not None else load_golden(name)
    iq, lim = g["iq"], g["liminfo"]
    if stupid is not None:
        d["stupid"] = stupid
    cfg = lrh_config(d, iq, **cfg_kw)
    api = open_fn(cfg)
    api.timf1_write(iq)
    api.set_liminfo(lim)
    if "foldcorr" in g:
        api.set_foldcorr(g["foldcorr"])
    api.set_mix1_selfreq(d["fq"])
    if d["fft3_n"]:
        n3 = 1 << d["fft3_n"]
        # stand-in for make_bg_filter's output; the golden carries the table the compiled reference ran with
        api.set_bg_filterfunc(g["bg_filterfunc"] if "bg_filterfunc" in g else
                              np.exp(-((np.arange(n3) - n3 / 2) / (n3 / 6.0)) ** 2).astype(np.float32))
        if d["mixer_mode"] == 2:                     # bg.mixer_mode = 2: the FIR the compiled reference ran with (stand-in for make_bg_filter's)
            api.set_basebraw_fir(g["basebraw_fir"])
    itrace, wf_lines, mixtrace, cleared = [], [], [], []
    afc, afc_t = None, [0]
    if d["afc"]:
        # the reference harness supplied these per-transform frequencies (its AFC_SUPPLY macro); replay the same supply
        afc = abi.AfcTables(cfg.max_fft2n if d["second_fft"] else cfg.max_fft1n, d["afc_bw"])

    def mix1_afc(nx, mask, call):
        t = afc_t[0]
        if t == 0:
            afc.mid[nx] = g["afc_fq0"][0]
        afc.mid[(nx + 1) & mask] = g["afc_supplied"][t]
        afc_t[0] += 1
        call(afc, 1)
    nblk = d["nblk"]
    b = 0
    while b < nblk and not d["second_fft"]:
        # second fft disabled: fft1_b -> fft1_c -> fft1_mix1_fixed (ref_harness.c, !second branch)
        B = min(batch, nblk - b)
        api.fft1_b(B)
        api.fft1_c(B)
        for _ in range(B):
            if afc is not None:
                mix1_afc(api.p.fft1_nx, cfg.max_fft1n - 1, api.fft1_mix1_afc)
            else:
                api.fft1_mix1_fixed(1)
            ms = api.mix1_state()
            mixtrace.append([ms.mix1_point, ms.mix1_phase, ms.mix1_phase_rot, ms.mix1_phase_step,
                             ms.mix1_old_phase, ms.mix1_old_point, api.p.timf3_pa, api.p.fft1_nx])
        p = api.p
        itrace.append([0, 0, 0, 0, 0, 0, p.fft1_nx, p.fft1_sumsq_pa, p.fft1_sumsq_counter, 0, p.fft1_liminfo_cnt])
        b += B
    while b < nblk:
        B = min(batch, nblk - b)
        api.fft1_b(B)
        api.fft1_c(B)
        api.make_timf2(B)
        api.first_noise_blanker()
        if d["blockpower_block"]:
            api.compute_timf2_powersum()
        k = api.fft2_available()
        for _ in range(k):
            wptr = api.p.wg_waterf_ptr
            api.make_fft2(1)
            if api.p.wg_waterf_ptr != wptr:
                wf_lines.append(api.export(abi.RING_WG_WATERF, wptr, cfg.wf_xpixels))
            if afc is not None:
                mix1_afc(api.p.fft2_nx, cfg.max_fft2n - 1, api.fft2_mix1_afc)
            else:
                api.fft2_mix1_fixed(1)
            if d["fft3_n"]:
                k3 = api.fft3_available()
                if k3:
                    api.make_fft3_all(k3)
                    api.fft3_mix2(k3)
            ms = api.mix1_state()
            mixtrace.append([ms.mix1_point, ms.mix1_phase, ms.mix1_phase_rot, ms.mix1_phase_step,
                             ms.mix1_old_phase, ms.mix1_old_point, api.p.timf3_pa, api.p.fft2_nx])
        bs = api.blanker_state()
        p = api.p
        itrace.append([p.timf2_pa, p.timf2p_fit, p.timf2_pn2, p.timf2_px, bs.timf2_noise_floor,
                       bs.stupid_bln_limit, p.fft2_na, p.fft1_sumsq_pa, p.fft1_sumsq_counter,
                       p.fft1_lowlevel_points, p.fft1_liminfo_cnt])
        cleared.append(bs.timf2_cleared_points)              # (a clearing decision that went the other way shows here even after the ring has moved on)
        b += B
    out = {key: api.export(ring) for ring, key in RINGS}
    if d["fft3_n"]:
        out["fft3"] = api.export(abi.RING_FFT3)
        out["baseb_raw"] = api.export(abi.RING_BASEB_RAW)
        out["fft3_ptrs"] = np.array([api.p.fft3_pa, api.p.timf3_px, api.fft3_interleave_points])
        out["baseb_ptrs"] = np.array([api.p.baseb_pa, api.p.fft3_px])
        out["timf3_py"] = api.p.timf3_py
    if d["blockpower_block"]:
        out["timf2_blockpower"] = api.export(abi.RING_TIMF2_BLOCKPOWER)
        out["blockpower_ptrs"] = np.array([api.p.timf2_blockpower_pa, api.p.timf2_pb])
    if afc is not None:
        out["afc_tables"] = np.stack([afc.mid, afc.slope, afc.curv, afc.start])
    out["itrace"] = np.array(itrace, np.int64)
    out["cleared_trace"] = np.array(cleared, np.int64)
    out["wf_lines"] = np.array(wf_lines, np.int16).reshape(-1, cfg.wf_xpixels)
    out["mixtrace"] = np.array(mixtrace, np.float64).reshape(-1, 8)
    out["lowlevel_fraction"] = api.p.fft1_lowlevel_fraction
    out["cfg"] = cfg
    out["api"] = api
    return out


def relerr(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    ct = np.complex128 if (np.iscomplexobj(a) or np.iscomplexobj(b)) else np.float64
    a, b = a.astype(ct), b.astype(ct)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def waterfall_itab(cfg):
    """yfac index per pixel as lrh_open builds it (float-accumulated like fft2.c:713-729)"""
    n1, n2 = 1 << cfg.fft1_n, 1 << cfg.fft2_n
    r = max(1, n2 // n1)
    mode = cfg.wf_mode
    hx, hp = (1, 1) if mode == 1 else ((mode, 0) if mode > 1 else (0, -mode))
    if hx > 0 and hx >= r:
        wx, wp = hx // r, 0
    else:
        wx, wp = 0, max(1, hp * r if hp > 0 else r // (hx if hx > 0 else 1))
    a2 = np.float32(wx) if wx > 0 else np.float32(1.0 / wp)
    a3 = np.float32(cfg.wf_first_xpoint // r + 0.5 * a2)
    itab = []
    for _ in range(cfg.wf_xpixels):
        itab.append(min(max(int(a3), 0), n1 - 1))
        a3 = np.float32(a3 + a2)
    return np.array(itab), hx, hp


_TRUTH = {}


def truth_of(name, golden=None, stupid=None, **cfg_kw):
    """run_case through the float64 build of the oracle (oracle/liblinrad_oracle64.so, tests/oracle_binding.py: every float of the restatement a
    double; the host bookkeeping that DEFINES the mixer -- mix1 phases, AFC tables, blanker thresholds -- stays in the reference's float32).
    Rings come back unrounded, `wf_pre` holds the waterfall lines before the truncation to short.  Cached per case."""
    key = (name, stupid, tuple(sorted(cfg_kw.items())))
    if key not in _TRUTH:
        from oracle_binding import open_truth
        t = run_case(open_truth, name, golden=golden, stupid=stupid, **cfg_kw)
        t["wf_pre"] = np.array(t["api"].wf_pre_lines, np.float64).reshape(-1, t["cfg"].wf_xpixels)
        t.pop("api").close()
        _TRUTH[key] = t
    return _TRUTH[key]


def truth_gate(rep, key, hip, ref, truth, tol=1e-5, factor=1.0):
    """THE FLOAT GATE (north_star: within 1e-5 RMS of the reference).  A ring passes on its relative RMS error against the reference; above
    the tolerance it passes only if the HIP result is at least as close to the float64 truth as the reference's own float32 result is
    (`truth`: the array, or a callable that makes it -- only called when needed).  No absolute floors."""
    e = relerr(hip, ref)
    rep[key] = e
    if e <= tol:
        return e
    assert truth is not None, f"{key}: relative RMS error {e:.3e} > {tol} and no float64 truth to measure both sides against"
    t = truth() if callable(truth) else truth
    eh, er = relerr(hip, t), relerr(ref, t)
    rep.setdefault("above_tol", {})[key] = {"hip_vs_ref": e, "hip_vs_truth": eh, "ref_vs_truth": er, "ratio": eh / max(er, 1e-300)}
    assert eh <= factor * er, f"{key}: rel {e:.3e} > {tol}; against the float64 truth HIP {eh:.3e}, the reference {er:.3e} (ratio {eh / max(er, 1e-300):.3f} > {factor})"
    return e


def waterfall_gate(rep, hip_lines, ref_lines, wf_pre):
    """THE INTEGER GATE for the quantised waterfall lines (short y = 1000 log10(sum * yfac), fft2.c:728-733).  Two float32 transforms of
    different rounding order cannot agree on every truncation, so each side is held to the float64 truth t = trunc(1000 log10(..)) instead:
      * every bin within 70 dB of the strongest one: |hip - t| <= 1 (a truncation boundary), or no more than the reference's own |ref - t|;
      * deeper bins (float32 noise of the transform itself exceeds a count there): at most one count beyond that;
      * HIP disagrees with the truth's integers no more often than the reference does (3 sigma of a count that small)."""
    h, r = hip_lines.astype(np.int64), ref_lines.astype(np.int64)
    t = np.clip(np.trunc(wf_pre), -32767, 32767).astype(np.int64)
    assert h.shape == r.shape == t.shape, (h.shape, r.shape, t.shape)
    dh, dr = np.abs(h - t), np.abs(r - t)
    depth = r.max() - r
    lim = np.maximum(1, dr) + (depth >= 7000)
    nh, nr = int(np.count_nonzero(dh)), int(np.count_nonzero(dr))
    rep["wf_vs_truth"] = {"bins": int(h.size), "hip_ne_truth": nh, "ref_ne_truth": nr, "hip_maxdev": int(dh.max()), "ref_maxdev": int(dr.max()),
                          "hip_beyond_ref": int(np.count_nonzero(dh > np.maximum(1, dr)))}
    assert np.all(dh <= lim), f"waterfall: {int(np.count_nonzero(dh > lim))} bins further from the float64 truth than a truncation boundary and the reference's own deviation allow (max {int(dh.max())})"
    assert nh <= nr + 3 * np.sqrt(nr) + 3, f"waterfall: {nh} bins differ from the truth's integers, the reference's own count is {nr}"


def golden_itrace(g):
    """columns of the reference trace matching run_case's itrace"""
    it = g["itrace"].reshape(-1, 16)
    return np.stack([it[:, 0], it[:, 1], it[:, 2], it[:, 3], it[:, 12], it[:, 13], it[:, 8], it[:, 9], it[:, 10],
                     it[:, 11], it[:, 15]], axis=1).astype(np.int64)


def compare_with_golden(out, g, tol=1e-5, check_blanker_exact=True, floor_slack=0,
                        mask_pending_timf2=False, truth=None, truth_factor=1.0, skip=()):
    """Assert parity of every ring, pointer trace and quantised line with the reference's output.

    Float rings: relative RMS error <= tol (north_star: 1e-5); a ring above it must be at least as close to the float64 truth
    as the reference's own result (truth_gate; `truth`: truth_of(case) or a callable that returns it).
    Pointers / call pattern: exact.  Blanker: identical cleared-sample set.  Quantised waterfall bins: against the float64 truth's
    integers (waterfall_gate) when `truth` is given, else exact.
    """
    rep = {}
    truth_cache = []

    def T():
        if not truth_cache:
            truth_cache.append(truth() if callable(truth) else truth)
        return truth_cache[0]

    def gate(key, a, b):
        truth_gate(rep, key, a, b, (lambda: T()[key][:a.size]) if truth is not None else None, tol, truth_factor)
    if "fft3" in out:
        assert np.array_equal(out["fft3_ptrs"], g["fft3_ptrs"][1:]), "fft3 pointers differ"
        gate("fft3", out["fft3"], g["fft3"])
        if "baseb_raw" in g:                                  # fft3_mix2's filter / decimate part, run by the compiled reference
            assert np.array_equal(out["baseb_ptrs"], g["baseb_ptrs"]), "baseband pointers differ"
            if "timf3_py" in g:
                assert out["timf3_py"] == int(g["timf3_py"][0]), "timf3_py differs"
            assert np.count_nonzero(g["baseb_raw"]) > 500
            gate("baseb_raw", out["baseb_raw"], g["baseb_raw"])
    if "timf2_blockpower" in out:
        assert np.array_equal(out["blockpower_ptrs"], g["blockpower_ptrs"]), "timf2 powersum pointers differ"
        e = relerr(out["timf2_blockpower"], g["timf2_blockpower"])
        rep["timf2_blockpower"] = e
        assert e <= tol, f"timf2_blockpower: {e:.3e}"
    if "afc_tables" in out:
        ref = np.stack([g["afc_fq_mid"], g["afc_fq_slope"], g["afc_fq_curv"], g["afc_fq_start"]])
        assert np.array_equal(out["afc_tables"], ref), "AFC frequency tables differ from the reference's"
    gi, oi = golden_itrace(g), out["itrace"]
    assert gi.shape == oi.shape, (gi.shape, oi.shape)
    ptr_cols = [0, 1, 2, 3, 6, 7, 8, 9, 10]
    assert np.array_equal(gi[:, ptr_cols], oi[:, ptr_cols]), "ring pointer trace differs from reference"
    dfloor = np.abs(gi[:, 4] - oi[:, 4]).max()
    rep["noise_floor_maxdiff"] = int(dfloor)
    assert dfloor <= floor_slack, f"timf2_noise_floor trace differs by {dfloor}"
    if floor_slack == 0:
        assert np.array_equal(gi[:, 5], oi[:, 5]), "stupid_bln_limit trace differs"
    stride = int(g["__stride"]) if "__stride" in g else 1
    strided = {"fft1_float", "fft1_sumsq", "timf2_float", "timf2_pwr_float", "fft2_float", "fft2_power_float"}

    def sub(key, arr):
        return arr[::stride] if (stride > 1 and key in strided) else arr

    for _, key in RINGS:
        if key in skip:                                        # (a ring the caller's configuration does not hold: cfg.fft2_float_sparse)
            continue
        a = out[key]
        b = g[key] if (stride > 1 and key in strided) else g[key][:a.size]
        if key == "timf2_float" and mask_pending_timf2:
            # sin^2 overlap-add: the reference parks the raw second half of the latest transform beyond timf2_pa
            # (timf2.c:1018-1025) until the next block adds to it; the HIP path never stores that scratch
            # (k_timf2).  Consumers only read up to timf2_pa, so the region is excluded here.
            cfg = out["cfg"]
            n1 = 1 << cfg.fft1_n
            pa = int(out["itrace"][-1, 0])
            idx = (pa + np.arange(4 * (n1 // 2))) % a.size
            keep = np.ones(a.size, bool)
            keep[idx] = False
            a = a * keep
            b = b * sub(key, keep)
        a = sub(key, a)
        e = relerr(a, b)
        if key == "timf2_pwr_float" and "timf2_pwr_float_noblank" in g:
            # power of the despiked weak signal: judge the error against the scale of the signal the transform
            # actually carried (pulses included), like every other ring
            den = np.linalg.norm(g["timf2_pwr_float_noblank"].astype(np.float64))
            num = np.linalg.norm(a.astype(np.float64) - b)
            e = 0.0 if num == 0 else float(num / max(den, 1e-300))
        out.setdefault("_cmp", {})[key] = (a, b)
        rep[key] = e
        if key == "timf3_float":
            gate(key, a, b)
        else:
            assert e <= tol, f"{key}: relative RMS error {e:.3e} > {tol}"
    a, b = sub("timf2_pwr_float", out["timf2_pwr_float"]) == 0, g["timf2_pwr_float"] == 0
    inter, union = np.sum(a & b), max(np.sum(a | b), 1)
    rep["cleared_jaccard"] = float(inter / union)
    rep["blanker_flips"] = int(np.sum(a != b))
    rep["cleared_samples"] = int(np.sum(b))
    if check_blanker_exact:
        assert np.array_equal(a, b), f"cleared-sample set differs (Jaccard {inter / union:.6f})"
    gw, ow = g["wf_lines"].reshape(-1, out["cfg"].wf_xpixels), out["wf_lines"]
    assert gw.shape == ow.shape, (gw.shape, ow.shape)
    if gw.size and "wf_lines" not in skip:                   # (skipped by a caller that holds the lines itself: the random tests, with their own depth mask)
        diff = np.abs(gw.astype(np.int32) - ow.astype(np.int32))
        rep["wf_mismatch_frac"] = float(np.mean(diff != 0))
        rep["wf_maxdiff"] = int(diff.max())
        rep["wf_bins"] = int(diff.size)
        rep["wf_mismatches"] = int(np.count_nonzero(diff))
        if truth is not None:
            waterfall_gate(rep, ow, gw, T()["wf_pre"])
        else:
            assert rep["wf_mismatches"] == 0, f"{rep['wf_mismatches']} waterfall bins differ (no float64 truth given to judge them)"
    gm, om = g["mixtrace"].reshape(-1, 8), out["mixtrace"]
    if om.size:
        assert np.array_equal(gm[:len(om), [0, 5, 6, 7]], om[:, [0, 5, 6, 7]]), "mix1 point / pointer trace differs"
        assert np.allclose(gm[:len(om), 1:5], om[:, 1:5], rtol=0, atol=2e-6), "mix1 phase bookkeeping differs"
    return rep
