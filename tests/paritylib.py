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


def run_case(open_fn, name, golden=None, stupid=None, batch=1, **cfg_kw):
    """Drive the case block by block in the reference call pattern (ref_harness.c main loop)."""
    d = case_params(name)
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
    itrace, wf_lines, mixtrace = [], [], []
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
        b += B
    out = {key: api.export(ring) for ring, key in RINGS}
    if d["fft3_n"]:
        out["fft3"] = api.export(abi.RING_FFT3)
        out["baseb_raw"] = api.export(abi.RING_BASEB_RAW)
        out["fft3_ptrs"] = np.array([api.p.fft3_pa, api.p.timf3_px, api.fft3_interleave_points])
        out["baseb_ptrs"] = np.array([api.p.baseb_pa, api.p.fft3_px])
    if d["blockpower_block"]:
        out["timf2_blockpower"] = api.export(abi.RING_TIMF2_BLOCKPOWER)
        out["blockpower_ptrs"] = np.array([api.p.timf2_blockpower_pa, api.p.timf2_pb])
    if afc is not None:
        out["afc_tables"] = np.stack([afc.mid, afc.slope, afc.curv, afc.start])
    out["itrace"] = np.array(itrace, np.int64)
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


def waterfall_boundary_report(out, g, diff):
    """SURVEY 8d gate for the quantised waterfall bins.  A line is (int)(1000 log10(sum * yfac)) of a waterfall averaging group's
    power sum (fft2.c:728-733); the sums of the last lines are rebuilt here from the golden's per-transform power ring
    (fft2_power_float, float32 adds in transform order like fft2.c:655-670), and for every bin of those lines where the HIP line
    differs from the reference's the distance of the pre-rounding value to the nearest integer boundary is reported, next to the
    float32 noise (in counts) a bin that far below the strongest one carries.  None where the mapping is not one data point or
    one maximum per pixel, or the golden stores strided rings."""
    cfg = out["cfg"]
    if ("__stride" in g and int(g["__stride"]) > 1) or cfg.wf_mode < 1 or "fft2_power_float" not in g or not cfg.second_fft_enable:
        return None
    n2, avg, nring = 1 << cfg.fft2_n, cfg.waterfall_avgnum, cfg.max_fft2n
    T = len(out["mixtrace"])                                   # fft2 transforms of the run
    nlines = diff.shape[0]
    power = g["fft2_power_float"].reshape(nring, n2)
    yfac = out["api"].get_table("wg_waterf_yfac", 1 << cfg.fft1_n)
    itab, hx, hp = waterfall_itab(cfg)
    gw = g["wf_lines"].reshape(nlines, -1)
    dists, noise, lines_checked, mism = [], [], 0, 0
    peak = gw.max()
    for line in range(nlines):
        t0 = line * avg
        if t0 < T - nring or t0 + avg > T:
            continue
        acc = power[t0 % nring].astype(np.float32).copy()
        for t in range(t0 + 1, t0 + avg):
            acc = (acc + power[t % nring]).astype(np.float32)
        first = cfg.wf_first_xpoint
        if cfg.wf_mode == 1:
            ps = acc[first:first + cfg.wf_xpixels]
        else:                                                      # maximum of hx data points per pixel (fft2.c:795-811)
            ps = np.array([acc[min(first + t * hx, n2):min(first + (t + 1) * hx, n2)].max(initial=0.0) for t in range(cfg.wf_xpixels)], np.float32)
        with np.errstate(divide="ignore"):
            v = 1000.0 * np.log10((ps * yfac[itab[:ps.size]]).astype(np.float32).astype(np.float64))
        y = np.clip(np.trunc(v), -32767, 32767)
        if not np.array_equal(y.astype(np.int64), gw[line, :ps.size].astype(np.int64)):
            continue                                               # the rebuilt sum is not this line's (ring position): skip it
        lines_checked += 1
        bad = np.nonzero(diff[line, :ps.size])[0]
        mism += bad.size
        for i in bad:
            fr = v[i] - np.floor(v[i])
            dists.append(float(min(fr, 1.0 - fr)))
            noise.append(float(868e-6 * 10.0 ** ((peak - gw[line, i]) / 2000.0)))
    dists, noise = np.array(dists), np.array(noise)
    return {"lines_checked": lines_checked, "checked_mismatches": int(mism),
            "within_1e-3": int(np.sum(dists <= 1e-3)), "max_distance": float(dists.max()) if dists.size else 0.0,
            # the noise figure is an RMS estimate: a bin flips when its error exceeds the distance, so distances of a few sigma occur
            "beyond_noise": int(np.sum(dists > np.maximum(1e-3, 4 * noise))),
            "max_distance_over_noise": float(np.max(dists / noise)) if dists.size else 0.0,
            "distances": [round(float(x), 5) for x in dists[:24]], "float32_noise_counts": [round(float(x), 5) for x in noise[:24]]}


def golden_itrace(g):
    """columns of the reference trace matching run_case's itrace"""
    it = g["itrace"].reshape(-1, 16)
    return np.stack([it[:, 0], it[:, 1], it[:, 2], it[:, 3], it[:, 12], it[:, 13], it[:, 8], it[:, 9], it[:, 10],
                     it[:, 11], it[:, 15]], axis=1).astype(np.int64)


def compare_with_golden(out, g, tol=1e-5, check_blanker_exact=True, wf_max_mismatch=0.03, floor_slack=0,
                        mask_pending_timf2=False):
    """Assert parity of every ring, pointer trace and quantised line with the reference's output.

    Float rings: relative RMS error <= tol (north_star: 1e-5).  Weak-band products (timf3) are also held to an
    absolute bound tied to the float32 resolution of the full-band signal they are cut from.
    Pointers / call pattern: exact.  Blanker: identical cleared-sample set.  Quantised waterfall bins: exact except
    where the pre-rounding value sits within float32 noise of an integer boundary (|diff| <= 1 there).
    """
    rep = {"escapes": []}          # rings accepted on the absolute float32 floor instead of the relative tolerance

    def gate(key, e, err, floor):
        """relative tolerance, or the absolute float32 floor of the wide-band spectrum the ring was cut from; the report says which"""
        rep[key] = e
        rep.setdefault("abs_err", {})[key] = float(err)
        rep.setdefault("abs_floor", {})[key] = float(floor)
        if e > tol and err <= floor:
            rep["escapes"].append(key)
        assert e <= tol or err <= floor, f"{key}: rel {e:.3e}, abs {err:.3e} > floor {floor:.3e}"
    if "fft3" in out:
        assert np.array_equal(out["fft3_ptrs"], g["fft3_ptrs"][1:]), "fft3 pointers differ"
        a, b = out["fft3"].astype(np.float64), g["fft3"].astype(np.float64)
        # band-limited product of mix1: same float32 floor argument as timf3
        n2 = 1 << out["cfg"].fft2_n
        wide = np.linalg.norm(g["fft2_float"].astype(np.float64)) / np.sqrt(out["cfg"].max_fft2n)
        floor = 16 * 6e-8 * wide * np.sqrt(a.size / (2.0 * n2))
        gate("fft3", relerr(a, b), np.linalg.norm(a - b), floor)
        if "baseb_raw" in g:                                  # fft3_mix2's filter / decimate part, run by the compiled reference
            assert np.array_equal(out["baseb_ptrs"], g["baseb_ptrs"]), "baseband pointers differ"
            a, b = out["baseb_raw"].astype(np.float64), g["baseb_raw"].astype(np.float64)
            assert np.count_nonzero(b) > 500
            gate("baseb_raw", relerr(a, b), np.linalg.norm(a - b), floor * np.sqrt(a.size / out["fft3"].size))
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
            # band-limited product: allow float32 noise of the wide-band spectrum it was cut from
            cfg = out["cfg"]
            n2 = 1 << cfg.fft2_n
            wide = np.linalg.norm(g["fft2_float"].astype(np.float64)) * np.sqrt(stride) / np.sqrt(cfg.max_fft2n)
            nm = n2 >> cfg.mix1_bandwidth_reduction_n
            floor = 4 * 6e-8 * wide * np.sqrt(nm / n2) * np.sqrt(out[key].size / nm / 2) * np.sqrt(nm)
            err = np.linalg.norm(a.astype(np.float64) - b.astype(np.float64))
            gate(key, e, err, floor)
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
    if gw.size:
        diff = np.abs(gw.astype(np.int32) - ow.astype(np.int32))
        rep["wf_mismatch_frac"] = float(np.mean(diff != 0))
        rep["wf_maxdiff"] = int(diff.max())
        # float32 FFT noise is set by the strongest bin: a bin D dB below it carries a relative power error of
        # about 2e-6*10^(D/20) (two float32 transforms of different rounding order, each ~1e-6 of the peak), i.e.
        # 868*that in 0.01 dB counts.  Exact +-1 within 54 dB of the peak.
        allowed = 1 + np.floor(868e-6 * 10.0 ** ((gw.max() - gw.astype(np.float64)) / 2000.0))
        rep["wf_bins"] = int(diff.size)
        rep["wf_mismatches"] = int(np.count_nonzero(diff))
        rep["wf_boundary"] = waterfall_boundary_report(out, g, diff)
        assert np.all(diff <= allowed), f"waterfall bins differ by up to {diff.max()} (beyond float32 noise)"
        assert np.mean(diff != 0) <= wf_max_mismatch, f"{np.mean(diff != 0):.4f} of waterfall bins differ"
        wb = rep["wf_boundary"]
        if wb is not None and wb["checked_mismatches"]:
            # SURVEY 8d gate: a quantised bin may differ only where its pre-rounding value sits on a rounding boundary -- within
            # 1e-3 counts, or within the float32 noise the bin's depth below the strongest bin allows (the bound `allowed` is built from)
            assert wb["beyond_noise"] == 0, wb
    gm, om = g["mixtrace"].reshape(-1, 8), out["mixtrace"]
    if om.size:
        assert np.array_equal(gm[:len(om), [0, 5, 6, 7]], om[:, [0, 5, 6, 7]]), "mix1 point / pointer trace differs"
        assert np.allclose(gm[:len(om), 1:5], om[:, 1:5], rtol=0, atol=2e-6), "mix1 phase bookkeeping differs"
    return rep
