"""Shared by tests/test_shim_exec_cpu.py (oracle/_ref/shim_harness: the Linrad-side glue over the oracle's ABI, build container) and
tests/test_gpu_shim.py (oracle/_ref/shim_harness_hip: the same patched reference objects and the same glue linked to liblinrad_hip.so,
GPU box): run the head-less driver of the PATCHED reference with fft1 version 21 selected and hold what Linrad sees on the host
afterwards -- pointer globals after every block, blanker scalars, fft1_sumsq / fft1_slowsum, waterfall lines, fft2_powersum_float,
timf3 (and fft3 / baseb_raw computed from it by the reference's own host code), the AFC tables, liminfo -- plus the device rings
fetched at the end, against the goldens of the UNPATCHED compiled reference."""
import os
import subprocess
import types


import numpy as np

from paritylib import compare_with_golden, golden_itrace, load_golden, relerr, truth_gate, truth_of
from refcases import case_params, clever_case, harness_args, lrh_config, sellim_case
from refdump import load_dump

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_CASES = ["n10_n12", "n10_mix1only", "n10_n12_afc", "n10_afc_mix1only", "n10_n12_fft3", "n9_n11_sin3",
                "n10_n12_dword",                                 # int32 samples with an I/Q sample skew (DWORD_INPUT, ui.sample_shift)
                "n9_n11_real", "n8_n10_real_dword_rev"]          # real samples: fft1 version 22 (fft1_reherm_dit_one's job, fft1_re.c:32-131)


def run_harness(harness, args, tmp_path, files, timeout=600):
    paths = {}
    for k, arr in files.items():
        paths[k] = str(tmp_path / f"{k}.bin")
        arr.tofile(paths[k])
    fo = str(tmp_path / "out.bin")
    r = subprocess.run([harness] + args(paths, fo), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "hip_open: 0" in r.stderr
    return load_dump(fo)


def check_golden_case(harness, tmp_path, name, extra=(), tol=1e-5, truth_factor=1.0):
    """one golden case through the patched call sites; `extra`: more harness arguments (shim_threads=1 ...)"""
    d, g = case_params(name), load_golden(name)
    dump = run_harness(harness, lambda p, fo: harness_args(d, p["in"], p["lim"], fo) + list(extra), tmp_path, {"in": g["iq"], "lim": g["liminfo"]})
    cfg = lrh_config(d, g["iq"], max_batch=1)
    out = {k: dump[k] for k in ("fft1_float", "fft1_sumsq", "fft1_slowsum", "timf2_float", "timf2_pwr_float", "fft2_float",
                                "fft2_power_float", "fft2_powersum_float", "timf3_float")}
    out["itrace"] = golden_itrace(dump)
    out["wf_lines"] = dump["wf_lines"].reshape(-1, cfg.wf_xpixels)
    out["mixtrace"] = dump["mixtrace"].reshape(-1, 8).astype(np.float64)[:int(dump["final"][10])]
    out["cfg"] = cfg
    out["api"] = types.SimpleNamespace(get_table=lambda t, n: g[t][:n])
    if d["fft3_n"]:      # the reference's own make_fft3_all / fft3_mix2 ran on the host from the timf3 blocks the glue brought back
        out.update(fft3=dump["fft3"], fft3_ptrs=dump["fft3_ptrs"][1:], baseb_raw=dump["baseb_raw"], baseb_ptrs=dump["baseb_ptrs"])
    if d["blockpower_block"]:
        out.update(timf2_blockpower=dump["timf2_blockpower"], blockpower_ptrs=dump["blockpower_ptrs"])
    if d["afc"]:
        out["afc_tables"] = np.stack([dump["afc_fq_mid"], dump["afc_fq_slope"], dump["afc_fq_curv"], dump["afc_fq_start"]])
    # do_mix1 parks the raw second half of its newest block beyond timf3_pa until the next block adds to it (mix1.c:188-194);
    # consumers read up to timf3_pa, and that is what comes back to the host: the parked half block is left out of the comparison
    pa, blk = int(dump["final"][9]), int(dump["mixtrace"].reshape(-1, 8)[0, 6]) if dump["mixtrace"].size >= 8 else 0
    g_full = g
    g = dict(g)
    idx = np.zeros(0, int)
    if blk > 0:
        idx = (pa + np.arange(blk)) % out["timf3_float"].size
        out["timf3_float"] = out["timf3_float"].copy()
        g["timf3_float"] = g["timf3_float"].copy()
        out["timf3_float"][idx] = 0
        g["timf3_float"][idx] = 0

    def truth():                 # the float64 build of the oracle on the case's own call pattern (paritylib.truth_of), masked like the two sides
        t = dict(truth_of(name, g_full))
        t["timf3_float"] = t["timf3_float"].copy()
        t["timf3_float"][idx] = 0
        return t
    # fft1_c accumulates the running averaging period in place at fft1_sumsq_pa (fft1.c:4126-4169); the graphs read completed
    # periods, and completed periods are what the glue brings back: an unfinished one at the end of the run is left out
    it = dump["itrace"].reshape(-1, 16)
    if it[-1, 10] != 0:
        n1 = 1 << d["n1"]
        out["fft1_sumsq"] = out["fft1_sumsq"].copy()
        g["fft1_sumsq"] = g["fft1_sumsq"].copy()
        out["fft1_sumsq"][it[-1, 9]:it[-1, 9] + n1] = 0
        g["fft1_sumsq"][it[-1, 9]:it[-1, 9] + n1] = 0
    assert np.array_equal(dump["final"], g["final"]), "final ring pointers differ"
    # (the HIP path never stores the raw half block the reference parks beyond timf2_pa, timf2.c:1018-1025; nothing reads it: masked)
    # shim_sparse=1: the glue opens the fft2 spectrum ring the way a patched xlinrad64 does (hipshim.c, hip_sparse_rings): only the band mix1
    # cuts out exists on the device -- everything Linrad sees on the host (sums, waterfall lines, timf3, pointers) is compared as before
    skip = ("fft2_float", "fft2_power_float") if "shim_sparse=1" in extra else ()
    rep = compare_with_golden(out, g, tol=tol, mask_pending_timf2=True, truth=truth, truth_factor=truth_factor, skip=skip)
    if "shim_net=1" in extra:
        # NET_RXOUT_FFT1 / TIMF2 / FFT2 on: the hooks in front of the senders' reads filled the host rings (timf2_float and fft2_float above came
        # through hip_net_timf2 / hip_net_fft2 a packet's worth at a time); block 0 of fft1_float as the dispatcher's memcpy would have found it
        # is the transform BEFORE fft1_c's filter correction (network.c:383-388)
        rep["fft1_net"] = relerr(dump["fft1_first_raw"], g["fft1_first_raw"])
        assert np.count_nonzero(g["fft1_first_raw"]) > 100 and rep["fft1_net"] <= tol, rep
    # scalars the GUI reads, as the glue keeps them (blank1.c:1550-1601): noise floor and limit are in itrace; the despiked power here
    t, gt = dump["trace"].reshape(-1, 16), g["trace"].reshape(-1, 16)
    for col in (0, 1, 2, 3, 5):          # noise floor, limit, stupid_blanker_rate, despiked_pwr[0], fft1_lowlevel_fraction
        assert np.allclose(t[:, col], gt[:, col], rtol=2e-5, atol=1e-6), (col, np.abs(t[:, col] - gt[:, col]).max())
    return rep


def check_sellim_case(harness, tmp_path, extra=()):
    """fft1_update_liminfo (sellim.c:738, patched) -> hip_fft1_update_liminfo: the table after every update, as the glue publishes it
    into Linrad's liminfo[], has the reference's routing pattern; make_timf2 routes with it"""
    name = "sellim_n10_n12"
    d, sl, iq = sellim_case(name)
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz")))
    dump = run_harness(harness, lambda p, fo: [a for a in harness_args(d, p["in"], "none", fo) if not a.startswith("liminfo=")] + ["sellim=1"] +
                       [f"{k}={v}" for k, v in sl.items()] + list(extra), tmp_path, {"in": iq})
    n1 = 1 << d["n1"]
    got, ref = dump["liminfo_trace"].reshape(-1, n1), g["liminfo_trace"].reshape(-1, n1)
    batch = max([int(a.split("=")[1]) for a in extra if a.startswith("shim_batch=")] + [1])
    batched = batch > 1
    if batched:
        # gpu_fft1_batch_size = 4 transforms per fft1_b call with fft_avg1num = 5: the limiter hook (wcw.c:1124) then runs in the middle of an
        # averaging period, where the library reads the newest FINISHED period provided the glue hands fft1_sumsq_counter over
        # (hip_sellim_par).  Reference = the oracle driven through its own ABI in the same call pattern with the counter in lrh_ptrs.
        ref = oracle_sellim_batched(name, g, batch)
    assert got.shape == ref.shape and (batched or np.array_equal(dump["liminfo_trace_blk"], g["liminfo_trace_blk"]))
    assert np.array_equal(np.sign(got), np.sign(ref)), int(np.sum(np.sign(got) != np.sign(ref)))
    pos = ref > 0
    assert np.max(np.abs(got[pos] - ref[pos]) / ref[pos]) <= 2e-6
    if batched:
        return
    it, gi = dump["itrace"].reshape(-1, 16), g["itrace"].reshape(-1, 16)
    for col in (0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 15):     # pointers, fft1_lowlevel_points, noise floor, limit, fft1_liminfo_cnt
        assert np.array_equal(it[:, col], gi[:, col]), col
    assert relerr(dump["fft1_slowsum"], g["fft1_slowsum"]) <= 1e-5
    assert np.array_equal(dump["timf2_pwr_float"] == 0, g["timf2_pwr_float"] == 0)


def oracle_sellim_batched(name, g, batch):
    """liminfo after every limiter update with `batch` transforms per fft1_b / fft1_c / make_timf2 call, in the harness's order"""
    import sellimlib
    from oracle_binding import open_oracle
    d, _, iq = sellim_case(name)
    cfg = lrh_config(d, iq, max_batch=batch)
    api = open_oracle(cfg)
    api.timf1_write(iq)
    api.set_mix1_selfreq(d["fq"])
    par = sellimlib.sellim_params(cfg, g)
    par.exact_stats = 0
    trace, cnt = [], 0
    for _ in range(d["nblk"] // batch):
        api.fft1_b(batch), api.fft1_c(batch), api.make_timf2(batch)
        api.first_noise_blanker()
        for _ in range(api.fft2_available()):
            api.make_fft2(1)
            api.fft2_mix1_fixed(1)
        if api.p.fft1_liminfo_cnt != cnt:
            api.fft1_update_liminfo(par)
            cnt = api.p.fft1_liminfo_cnt
            trace.append(api.get_liminfo())
    api.close()
    return np.array(trace)


def check_clever_case(harness, tmp_path, extra=()):
    """init_blanker's tables (buf.c:1771-2057, built by the reference itself in the harness) are handed over by hip_first_noise_blanker
    when hg.clever_bln_mode is set: resume pointer, fitted-pulse counter, thresholds equal the unpatched reference's call by call"""
    name = "clever_n10_n12"
    d, cl, iq, lim, des = clever_case(name)
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz")))
    dump = run_harness(harness, lambda p, fo: harness_args(d, p["in"], p["lim"], fo) + ["clever=1", f"desired={p['des']}", f"clever_factor={cl['clever_factor']}"] + list(extra),
                       tmp_path, {"in": iq, "lim": lim, "des": des})
    it, gi = dump["itrace"].reshape(-1, 16), g["itrace"].reshape(-1, 16)
    for col in (0, 1, 2, 3, 6, 8, 12, 13):                 # timf2_pa, timf2p_fit, timf2_pn2, timf2_px, blanker_points, fft2_na, noise floor, limit
        assert np.array_equal(it[:, col], gi[:, col]), col
    t, gt = dump["trace"].reshape(-1, 16), g["trace"].reshape(-1, 16)
    upd = it[:, 7] == 0                                     # calls that ended with a threshold update: the glue reads the scalars back there
    assert np.array_equal(t[upd, 7], gt[upd, 7]) and np.any(gt[upd, 7] > 0), "clever_blanker_rate"
    n1 = 1 << d["n1"]
    keep = np.ones(dump["timf2_float"].size, bool)
    keep[(int(dump["final"][3]) + np.arange(4 * (n1 // 2))) % keep.size] = False
    assert relerr(dump["timf2_float"] * keep, g["timf2_float"] * keep) <= 1e-5
    assert np.array_equal((dump["timf2_pwr_float"] == 0) & keep[::4], (g["timf2_pwr_float"] == 0) & keep[::4])


def check_refusal(harness, tmp_path):
    """hip_open answers non-zero (-> lirerr(1463), wcw.c hunk) instead of letting host code run on rings that stay empty"""
    name = "n10_n12"
    d, g = case_params(name), load_golden(name)
    paths = {}
    for k, arr in (("in", g["iq"]), ("lim", g["liminfo"])):
        paths[k] = str(tmp_path / f"{k}.bin")
        arr.tofile(paths[k])
    r = subprocess.run([harness] + harness_args(d, paths["in"], paths["lim"], str(tmp_path / "o.bin")) + ["shim_refuse=1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and '"hip_open": 10' in r.stdout, r.stdout + r.stderr


def check_free_running(harness, refh, tmp_path, name, workers):
    """timf2_routine on its own thread beside the dispatcher and its fft1_b workers, second_fft and narrowband threads, all running
    free (oracle/ref_harness.c run_reference_threads; events like lxsys.c:415-447).  The blanker is off: its thresholds follow the
    call pattern, which the scheduler decides here.  Reference = the unpatched compiled reference in its single-CPU order."""
    d, g = dict(case_params(name)), load_golden(name)
    d["stupid"] = 0
    fin, flim = str(tmp_path / "in.bin"), str(tmp_path / "lim.bin")
    g["iq"].tofile(fin)
    g["liminfo"].tofile(flim)
    dumps = []
    for exe, extra, tag in ((refh, [], "ref"), (harness, ["shim_threads=2", f"shim_workers={workers}"], "hip")):
        fo = str(tmp_path / f"{tag}.bin")
        r = subprocess.run([exe] + harness_args(d, fin, flim, fo) + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        dumps.append(load_dump(fo))
    ref, hip = dumps
    # The dispatcher's rings and make_timf2's output do not depend on the grouping of the calls.  first_noise_blanker does -- a call that
    # finds fewer than blanker_min_points new samples returns at once (blank1.c:712), so the free-running run may end one fft2 transform
    # short of the single-CPU order -- and with it how far fft2 / mix1 got: those rings are compared where both runs hold the same transform.
    ptrs = [0, 1, 2, 3]                  # fft1_pa fft1_nb fft1_nx timf2_pa
    assert np.array_equal(ref["final"][ptrs], hip["final"][ptrs]), (ref["final"], hip["final"])
    nref, nhip = int(ref["final"][10]), int(hip["final"][10])
    assert 0 <= nref - nhip <= 1 and nhip > 4, (nref, nhip)
    n1, n2, M = 1 << d["n1"], 1 << d["n2"], d["max_fft2n"]
    keep2 = np.ones(ref["timf2_float"].size, bool)                 # the raw half block parked beyond timf2_pa (timf2.c:1018-1025)
    keep2[(int(ref["final"][3]) + np.arange(4 * (n1 // 2))) % keep2.size] = False
    same_slot = np.array([(nref - 1 - ((nref - 1 - s) % M)) == (nhip - 1 - ((nhip - 1 - s) % M)) for s in range(M)])
    f2r, f2h = ref["fft2_float"].reshape(M, -1)[same_slot], hip["fft2_float"].reshape(M, -1)[same_slot]
    blk3 = int(ref["mixtrace"].reshape(-1, 8)[0, 6])
    keep3 = np.ones(ref["timf3_float"].size, bool)                 # from hip's timf3_pa to the end of the half block parked beyond ref's (mix1.c:188-194)
    keep3[(int(hip["final"][9]) + np.arange((nref - nhip + 1) * blk3)) % keep3.size] = False
    errs = {"fft1_float": relerr(hip["fft1_float"], ref["fft1_float"]), "fft1_slowsum": relerr(hip["fft1_slowsum"], ref["fft1_slowsum"]),
            "timf2_float": relerr(hip["timf2_float"] * keep2, ref["timf2_float"] * keep2),
            "timf2_pwr": relerr(hip["timf2_pwr_float"] * keep2[::4], ref["timf2_pwr_float"] * keep2[::4]),
            "fft2_float": relerr(f2h, f2r), "timf3": relerr(hip["timf3_float"] * keep3, ref["timf3_float"] * keep3)}
    if nref == nhip:
        errs["fft2_powersum"] = relerr(hip["fft2_powersum_float"], ref["fft2_powersum_float"])
    print(name, (nref, nhip), errs)
    assert np.count_nonzero(ref["timf3_float"] * keep3) > 100 and same_slot.sum() >= M - 1
    assert max(v for k, v in errs.items() if k != "timf3") <= 1e-5, errs
    # timf3 of n9_n11_sin3 is a weak band under a strong carrier (1.2e-5 between the two float32 sides): above 1e-5 both against the float64 truth
    t3 = truth_of(name, stupid=0)["timf3_float"]              # (blanker off, like both runs)
    rep = {}
    if nref == nhip:
        truth_gate(rep, "timf3", hip["timf3_float"] * keep3, ref["timf3_float"] * keep3, t3 * keep3)
    else:
        assert errs["timf3"] <= 2e-5, errs         # (the two runs stopped a transform apart: the truth's ring holds the longer run's blocks)


def check_spur_case(harness, tmp_path, name="spur_n10_n12", tol=1e-5):
    """genparm[MAX_NO_OF_SPURS] != 0 with the AFC on is served: the reference's own acquisition calls -- store_new_spur / spur_phase_lock
    (spursub.c:619, 1247), hooked -- take the carrier on from the device-resident fft2 spectra (lrh_spur_acquire), eliminate_spurs runs
    inside lrh_make_fft2, and the loop state the glue brings back after every transform is the unpatched reference's (golden spur_trace);
    the rings behind the subtraction too"""
    import spurlib
    from refcases import spur_case
    d, sp, iq, lim = spur_case(name)
    g = spurlib.load(name)
    dump = run_harness(harness, lambda p, fo: harness_args(d, p["in"], p["lim"], fo) + ["spur=1"] + [f"{k}={v}" for k, v in sp.items()],
                       tmp_path, {"in": iq, "lim": lim})
    assert int(dump["spur_locked"][0]) == int(g["spur_locked"][0]) > 0, (dump["spur_locked"], g["spur_locked"])
    cfg = lrh_config(d, iq)
    out = dict(trace=dump["spur_trace"].reshape(-1, 12)[:, :10].astype(np.float64), cfg=cfg, fft2=dump["fft2_float"], ps2=dump["fft2_powersum_float"])
    if not d["second_fft"]:                                          # the spur lives in the fft1 transforms (fft1_c's AFC branch, hip_fft1_c)
        out = dict(trace=out["trace"], cfg=cfg, fft1=dump["fft1_float"], sumsq=dump["fft1_sumsq"])
    keep3 = np.ones(dump["timf3_float"].size, bool)                 # the half block parked beyond timf3_pa (mix1.c:188-194) stays on the device
    keep3[(int(dump["final"][9]) + np.arange(int(dump["mixtrace"].reshape(-1, 8)[0, 6]))) % keep3.size] = False
    out["timf3"] = dump["timf3_float"] * keep3
    g = dict(g)
    g["timf3_float"] = g["timf3_float"] * keep3
    # the search spectrum for new spurs as the glue left it in Linrad's spursearch_spectrum: summed and cleaned on the device, fetched when
    # make_fft2's counter says one is finished (hip_spur_after_fft2)
    info = dump["spursearch_info"]
    a_, b_ = int(info[2]), int(info[3])
    out["ss"] = (dump["spursearch_spectrum"][a_:b_ + 1], float(dump["spursearch_thresholds"][min(int(info[0]), 64) - 1]), int(info[0]), int(info[1]))
    out["ss_range"] = (a_, b_)
    rep = spurlib.compare(out, g, tol)
    assert "search_spectrum_err" in rep
    assert np.array_equal(dump["final"], g["final"]) if "final" in g else True
    return rep


def check_spur_clicks_case(harness, tmp_path, name="spur_n10_n12_clicks", tol=1e-5):
    """Linrad's whole init_spur_elimination (spursub.c:181-343) over the glue: the operator's click -> peak search in the search spectrum the
    device summed and cleaned -> store_new_spur / spur_phase_lock (hooked: lrh_spur_acquire, which also takes the carrier out of the
    transforms it locked on, initial_remove_spur) -> the ordering pass on the loop state the glue brought back: the weaker of two spurs closer
    than four bins is dropped -- the LAST one by counting no_of_spurs down and nothing else (the glue cuts the device's list back before the
    next transform, hip_spur_resync), another one through remove_spur (hooked) -- and swap_spurs (hooked) sorts by frequency.  Every spur's
    loop state after every transform and the rings behind the subtraction against the unpatched reference's."""
    from refcases import spur_case

    def wrap(x):
        return (x + np.pi) % (2 * np.pi) - np.pi
    d, sp, iq, lim = spur_case(name)
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz")))
    dump = run_harness(harness, lambda p, fo: harness_args(d, p["in"], p["lim"], fo) + ["spur=1"] + [f"{k}={v}" for k, v in sp.items()],
                       tmp_path, {"in": iq, "lim": lim})
    assert np.array_equal(dump["spur_clicks"], g["spur_clicks"]), (dump["spur_clicks"].reshape(4, 8), g["spur_clicks"].reshape(4, 8))
    got, ref = dump["spur_trace_all"].reshape(-1, 40).astype(np.float64), g["spur_trace_all"].reshape(-1, 40).astype(np.float64)
    assert got.shape == ref.shape and np.array_equal(got[:, :2], ref[:, :2]), "number of spurs after every transform"
    rep = {"transforms": int(ref.shape[0]), "spurs_held": sorted(set(int(x) for x in ref[:, 1])), "clicks": g["spur_clicks"].reshape(4, 8)[:3, :3].astype(int).tolist()}
    err = dict(freq=0.0, phase=0.0, d1=0.0, d2=0.0, ampl=0.0, noise=0.0)
    for s_ in range(4):
        on = ref[:, 1] > s_
        if not on.any():
            continue
        a, b = got[on][:, 2 + 9 * s_:11 + 9 * s_], ref[on][:, 2 + 9 * s_:11 + 9 * s_]
        assert np.array_equal(a[:, :2], b[:, :2]), f"spur {s_}: spur_location / spur_flag trace differs"
        err["freq"] = max(err["freq"], float(np.max(np.abs(a[:, 2] - b[:, 2]))))
        err["phase"] = max(err["phase"], float(np.max(np.abs(wrap(a[:, 3] - b[:, 3])))))
        err["d1"] = max(err["d1"], float(np.max(np.abs(wrap(a[:, 4] - b[:, 4])))))
        err["d2"] = max(err["d2"], float(np.max(np.abs(a[:, 5] - b[:, 5]))))
        err["ampl"] = max(err["ampl"], float(np.max(np.abs(a[:, 6] - b[:, 6]) / np.abs(b[:, 6]))))
        err["noise"] = max(err["noise"], float(np.max(np.abs(a[:, 7] - b[:, 7]) / np.abs(b[:, 7]))))
    rep.update({k + "_err": v for k, v in err.items()})
    assert err["freq"] < 1e-3 and err["phase"] < 20 * tol * 1e2 and err["d1"] < 1e-3 and err["ampl"] < 10 * tol and err["noise"] < 1e-3, rep
    keep3 = np.ones(dump["timf3_float"].size, bool)                 # the half block parked beyond timf3_pa (mix1.c:188-194) stays on the device
    keep3[(int(dump["final"][9]) + np.arange(int(dump["mixtrace"].reshape(-1, 8)[0, 6]))) % keep3.size] = False
    rep["fft2"], rep["ps2"] = relerr(dump["fft2_float"], g["fft2_float"]), relerr(dump["fft2_powersum_float"], g["fft2_powersum_float"])
    rep["timf3"] = relerr(dump["timf3_float"] * keep3, g["timf3_float"] * keep3)
    assert rep["fft2"] < tol and rep["ps2"] < tol and rep["timf3"] < tol, rep
    info = dump["spursearch_info"]
    a_, b_ = int(info[2]), int(info[3])
    assert np.array_equal(info, g["spursearch_info"])
    sp_h, sp_r = dump["spursearch_spectrum"][a_:b_ + 1], g["spursearch_spectrum"][a_:b_ + 1]
    assert np.array_equal(sp_h < 0, sp_r < 0), "wiped peaks differ"
    rep["search_spectrum_err"] = float(np.max(np.abs(sp_h - sp_r)) / float(np.max(sp_r)))
    assert rep["search_spectrum_err"] < tol, rep
    assert np.array_equal(dump["final"], g["final"])
    return rep


def _run_2ch(harness, tmp_path, name, chain, extra):
    from refcases import twochan_case
    d, frames, lim = twochan_case(name, chain=chain)
    args = ["channels=2", f"ch2_c1={d['ch2_c1']!r}", f"ch2_c2={d['ch2_c2']!r}"] + list(extra)
    if chain:
        args += ["chain2=1"] + [f"pol_c{i + 1}={v!r}" for i, v in enumerate(d["pol"])]
    dump = run_harness(harness, lambda p, fo: harness_args(d, p["in"], p["lim"], fo) + args, tmp_path, {"in": frames, "lim": lim})
    return d, dump


def check_twochan_case(harness, tmp_path, name="twochan_n10", tol=1e-5):
    """ui.rx_rf_channels = 2 is served: the glue opens one context per channel (both on the one GPU) and makes the exchanges that Linrad's
    single arrays make implicitly -- fft1_sumsq / fft1_slowsum as sums of the two contexts', the coupled blanker's power and noise sums, the
    all-gather of the fft2 bins for the cross products -- through host memory.  Against the goldens of the compiled TWO-CHANNEL reference:
    (a) fft1_b / fft1_c / make_timf2: both channels' spectra and time functions in Linrad's interleaved layout, the power sums;
    (b) the same with the two-channel first_noise_blanker after every block: thresholds, pointers, cleared samples."""
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz")))
    d, dump = _run_2ch(harness, tmp_path, name, False, ["corr=1"])      # genparm[FFT1_CORRELATION_SPECTRUM] = 1: the cross spectrum rides along
    n1 = 1 << d["n1"]
    it, gi = dump["itrace"].reshape(-1, 16), g["itrace"].reshape(-1, 16)
    assert np.array_equal(it[:, [0, 9, 10, 11, 15]], gi[:, [0, 9, 10, 11, 15]]), "pointer trace"
    rep = {k: relerr(dump[k], g[k]) for k in ("fft1_float", "fft1_sumsq", "fft1_slowsum")}
    pa = int(gi[-1, 0])                                     # finished samples only (the reference parks a raw half block beyond timf2_pa)
    rep["timf2_float"] = relerr(dump["timf2_float"][:pa], g["timf2_float"][:pa])
    rep["timf2_pwr"] = relerr(dump["timf2_pwr_float"][:pa // 8], g["timf2_pwr_float"][:pa // 8])
    assert pa > 0 and np.abs(g["timf2_float"][:pa].reshape(-1, 2, 2, 2)[:, 1]).max() > 0
    assert max(rep.values()) <= tol, rep
    # the correlation spectrum (fft1.c:4146-4150, 4584-4603): the glue gathers both channels' transforms and brings the three arrays back
    assert int(dump["slowcorr_tot_avgnum"][0]) == int(g["slowcorr_tot_avgnum"][0]) > 0
    rep.update({k: relerr(dump[k], g[k]) for k in ("fft1_corrsum", "fft1_slowcorr", "fft1_slowcorr_tot")})
    assert rep["fft1_corrsum"] <= tol and rep["fft1_slowcorr"] <= 4 * tol and rep["fft1_slowcorr_tot"] <= tol and np.count_nonzero(g["fft1_slowcorr"]) > 100, rep
    # (b) coupled blanker
    d, dump = _run_2ch(harness, tmp_path, name, False, ["blanker2=1"])
    it, gi = dump["itrace"].reshape(-1, 16), g["bln_itrace"].reshape(-1, 16)
    assert np.array_equal(it[:, [0, 1, 2, 6, 7]], gi[:, [0, 1, 2, 6, 7]]), "blanker pointer trace"
    assert np.abs(it[:, 12] - gi[:, 12]).max() <= 1 and np.abs(it[:, 13] - gi[:, 13]).max() <= 5, "noise floor / limit trace"
    fit = int(gi[-1, 1])
    got, ref = dump["timf2_float"].reshape(-1, 2, 2, 2)[:fit], g["bln_timf2_float"].reshape(-1, 2, 2, 2)[:fit]
    cleared_g, cleared_r = (got[:, 0] == 0).all(axis=(1, 2)), g["bln_timf2_pwr_float"][:fit] == 0
    rep["cleared"] = int(cleared_r.sum())
    if np.array_equal(it[:, 12], gi[:, 12]):
        assert np.array_equal(cleared_g, cleared_r) and rep["cleared"] > 50
        rep["bln_timf2"] = relerr(got, ref)
        assert rep["bln_timf2"] <= tol, rep
    else:
        assert (cleared_g & cleared_r).sum() / max((cleared_g | cleared_r).sum(), 1) > 0.97
    return rep


def check_twochan_chain(harness, tmp_path, name="twochan_n10", tol=1e-5, extra=()):
    """... and on through make_fft2 (each channel's transform, TWOCHAN_POWER cross products and their sums from both, the
    polarisation-independent waterfall line, fft2.c:1622-1640, 1700-1815) and fft2_mix1_fixed; fft3 and fft3_mix2's polarisation
    transform then run as the reference's own host code on the interleaved timf3 the glue brought back"""
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"{name}_chain.npz")))
    d, dump = _run_2ch(harness, tmp_path, name, True, list(extra))
    assert np.array_equal(dump["final"], g["final"]), (dump["final"], g["final"])
    n2 = 1 << d["n2"]
    rep = {k: relerr(dump[k], g[k]) for k in ("fft2_float", "fft2_xypower", "fft2_xysum")}
    assert rep["fft2_float"] <= tol and rep["fft2_xypower"] <= 2 * tol and rep["fft2_xysum"] <= 2 * tol, rep
    gw, ow = g["wf_lines"].astype(np.int32), dump["wf_lines"].astype(np.int32)
    assert gw.shape == ow.shape and np.abs(gw - ow).max() <= 2 and np.mean(gw != ow) < 0.02
    # timf3 (interleaved {ch0, ch1}): the weak band under the float32 floor of the wide spectrum, as in tests/test_twochan.py
    blk = int(dump["mixtrace"].reshape(-1, 8)[0, 6])
    keep = np.ones(dump["timf3_float"].size, bool)
    keep[(int(dump["final"][9]) + np.arange(blk)) % keep.size] = False
    a, b = (dump["timf3_float"] * keep).astype(np.float64), (g["timf3_float"] * keep).astype(np.float64)
    wide = np.linalg.norm(g["fft2_float"].astype(np.float64)) / np.sqrt(g["fft2_float"].size / (4 * n2))
    nm = n2 >> d["mixred"]
    floor = 4 * 6e-8 * wide * np.sqrt(nm / n2) * np.sqrt(a.size / nm / 2) * np.sqrt(nm)
    rep["timf3"] = relerr(a, b)
    assert np.abs(b).max() > 0 and (rep["timf3"] <= tol or np.linalg.norm(a - b) <= floor), rep
    for k in ("fft3", "baseb_raw", "baseb_raw_orthog"):     # the reference's own host code on what the glue brought back
        rep[k] = relerr(dump[k], g[k])
        assert rep[k] <= 20 * tol, rep
    assert np.array_equal(dump["fft3_ptrs"], g["fft3_ptrs"]) and np.array_equal(dump["baseb_ptrs"], g["baseb_ptrs"])
    # NET_RXOUT_FFT1 / TIMF2 / FFT2 with two channels: the hooks in front of the senders' reads fill Linrad's interleaved host rings, a packet's
    # worth at a time: the same rings as fetched whole; the FFT1 payload (before fft1_c's correction) = the golden's corrected spectrum / the table
    if extra:
        return rep
    d2, dn = _run_2ch(harness, tmp_path, name, True, ["shim_net=1"])
    assert np.array_equal(dn["timf2_float"], dump["timf2_float"]) and np.array_equal(dn["fft2_float"], dump["fft2_float"])
    assert np.count_nonzero(dn["timf2_float"]) > 1000 and relerr(dn["fft2_float"], g["fft2_float"]) <= tol
    # (the FFT1 payload -- lrh_export_fft1_net, pinned per channel by the one-channel goldens' fft1_first_raw -- only has to land interleaved:
    # both channels present in block 0's slot, different from each other; the fft1 ring of this run has lapped, so there is nothing to divide by)
    raw = dn["fft1_first_raw"].reshape(-1, 2, 2)
    assert np.count_nonzero(raw[:, 0]) > raw.shape[0] and np.count_nonzero(raw[:, 1]) > raw.shape[0] and not np.array_equal(raw[:, 0], raw[:, 1])
    return rep
