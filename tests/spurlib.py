"""Spur-subtraction parity machinery shared by the oracle (CPU) and HIP (GPU) tests: drive a SPUR case in the harness's order, hand the
reference's acquired spur over at the transform where the reference locked it (control plane), and compare the tracking loop's
state after every transform and the rings behind the subtraction with the reference golden."""
import os

import numpy as np

from linrad_amd import abi
from linrad_amd.abi import LrhSpur
from refcases import lrh_config, spur_case

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))


def run(open_fn, name, g, batch=1, acquire=False, case=None):
    """acquire: the spur is found by the API's own store_new_spur / spur_phase_lock (spur_acquire) on the resident spectra instead of
    being handed over with the reference's acquisition result.  case: (d, sp, iq, lim) of another input (the random tests; g then carries the
    line-shape table and the hand-over point only)"""
    d, sp, iq, lim = case if case is not None else spur_case(name)
    if "iq" in g:
        assert np.array_equal(iq, g["iq"])
    else:                                                    # (large input: the golden holds two checksums of it)
        assert int(iq.astype(np.int64).sum()) == int(g["iq_sum"][0]) and int((iq.astype(np.int64) * (np.arange(iq.size) % 251)).sum()) == int(g["iq_sum"][1])
    cfg = lrh_config(d, iq)
    api = open_fn(cfg)
    api.timf1_write(iq)
    api.set_liminfo(lim)
    api.set_mix1_selfreq(d["fq"])
    st = g["spur_init_state"]
    speknum, start = int(st[10]), int(g["spur_locked"][0])
    api.spur_config(4, speknum, g["spur_spectra"])
    first, last = (int(g["spursearch_info"][2]), int(g["spursearch_info"][3])) if "spursearch_info" in g else (0, (1 << cfg.fft2_n) - 1)
    api.spur_search_config(first, last)                      # the search for further spurs runs beside the tracking (fft2.c:673-699)
    trace, nfft2, handed, unremoved = [], 0, False, None
    if not d["second_fft"]:
        return run_fft1(api, cfg, d, sp, g, acquire, first, last)
    for b in range(d["nblk"]):
        api.fft1_b(1), api.fft1_c(1), api.make_timf2(1)
        api.first_noise_blanker()
        k = api.fft2_available()
        while k > 0:
            kb = min(k, batch) if handed else 1
            if not handed and nfft2 + 1 > start:
                raise AssertionError("hand-over point missed")
            if not handed:
                kb = 1
            api.make_fft2(kb)
            nfft2 += kb
            k -= kb
            was_handed = handed
            # acquisition behind the transform and ahead of mix1, where second_fft / spur_removal run it (wcw.c:288-303, 204-247): the lock takes
            # the carrier out of the spur_speknum transforms it was closed on as well (initial_remove_spur, spursub.c:309, 346)
            if not handed and nfft2 == start and acquire:
                assert api.spur_acquire(sp["spur_pnt"]), "no lock"
                handed = True
                acq = api.spur_get()[0]
            elif not handed and nfft2 == start:                   # the reference's acquisition result, handed over by the control plane
                q = LrhSpur(int(st[0]), int(st[1]), *[float(x) for x in st[2:9]])
                maxn = cfg.max_fft2n
                api.spur_set([q], g["spur_init_table"][:maxn * 14], g["spur_init_signal"][:2 * maxn], g["spur_init_ind"][:maxn])
                handed = True
                # (a hand-over does not rewrite the ring: the rows initial_remove_spur touched and what mix1 makes of the newest of them differ)
                unremoved = dict(rows=[(api.p.fft2_na - 1 - m) % maxn for m in range(speknum)], timf3=(api.p.timf3_pa,))
            api.fft2_mix1_fixed(kb)
            if handed and not was_handed and not acquire:
                unremoved["timf3"] += (api.p.timf3_pa,)
            if was_handed:
                s = api.spur_get()[0]
                trace.append([s.spur_location, s.spur_flag, s.spur_freq, s.spur_d0pha, s.spur_d1pha, s.spur_d2pha, s.spur_ampl, s.spur_noise, s.spur_avgd2, nfft2 - 1])
    ss = api.spur_search_get()
    return dict(api=api, cfg=cfg, d=d, ss=ss, ss_range=(first, last), acq=(acq if acquire else None), unremoved=unremoved, trace=np.array(trace, np.float64), fft2=api.export(abi.RING_FFT2_FLOAT), timf3=api.export(abi.RING_TIMF3_FLOAT),
                ps2=api.export(abi.RING_FFT2_POWERSUM))


def run_fft1(api, cfg, d, sp, g, acquire, first, last):
    """second fft off: the spur lives in the fft1 transforms; fft1_c takes it out of each new one (fft1.c:4196-4244) and keeps the search
    spectrum from their powers; acquisition behind fft1_c, the history ending with the newest transform"""
    st = g["spur_init_state"]
    start = int(g["spur_locked"][0])
    trace, handed, acq, unremoved = [], False, None, None
    for b in range(d["nblk"]):
        api.fft1_b(1)
        api.fft1_c(1)
        if handed:
            s = api.spur_get()[0]
            trace.append([s.spur_location, s.spur_flag, s.spur_freq, s.spur_d0pha, s.spur_d1pha, s.spur_d2pha, s.spur_ampl, s.spur_noise, s.spur_avgd2, b])
        elif b + 1 == start:
            if acquire:
                api.p.fft2_na = api.p.fft1_nb                      # ffts_na: the slot behind the newest transform (see oracle/ref_harness.c)
                assert api.spur_acquire(sp["spur_pnt"]), "no lock"
                acq = api.spur_get()[0]
            else:
                q = LrhSpur(int(st[0]), int(st[1]), *[float(x) for x in st[2:9]])
                maxn = cfg.max_fft1n
                api.spur_set([q], g["spur_init_table"][:maxn * 14], g["spur_init_signal"][:2 * maxn], g["spur_init_ind"][:maxn])
                # (a hand-over does not rewrite the ring: what mix1 makes of the newest transform, from which initial_remove_spur took the carrier, differs)
                unremoved = dict(rows=[(api.p.fft1_nb - 1 - m) % maxn for m in range(int(st[10]))], timf3=(api.p.timf3_pa,))
            handed = True
        api.fft1_mix1_fixed(1)
        if unremoved is not None and len(unremoved["timf3"]) == 1:
            unremoved["timf3"] += (api.p.timf3_pa,)
    ss = api.spur_search_get()
    return dict(api=api, cfg=cfg, d=d, ss=ss, ss_range=(first, last), acq=acq, unremoved=unremoved, trace=np.array(trace, np.float64), fft1=api.export(abi.RING_FFT1_FLOAT),
                sumsq=api.export(abi.RING_FFT1_SUMSQ), timf3=api.export(abi.RING_TIMF3_FLOAT))


def compare_fft1(out, g, tol):
    """loop state after every transform, the fft1 ring behind the subtraction, the power sums formed from it, timf3, the search spectrum"""
    ref, got = g["spur_trace"].reshape(-1, 12), out["trace"]
    assert got.shape[0] == ref.shape[0] and got.shape[0] > 10
    assert np.array_equal(got[:, :2], ref[:, :2]), "spur_location / spur_flag trace differs"

    def wrap(x):
        return (x + np.pi) % (2 * np.pi) - np.pi

    def rel(a, b):
        a, b = a.astype(np.float64), b.astype(np.float64)
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    h1, r1, h3, r3 = out["fft1"], g["fft1_float"], out["timf3"], g["timf3_float"]
    if out.get("unremoved"):                                    # handed over, not acquired: see run()
        n1_, u = 2 << out["cfg"].fft1_n, out["unremoved"]
        h1, r1, h3, r3 = h1.copy(), r1.copy(), h3.copy(), r3.copy()
        for row in u["rows"]:
            h1[row * n1_:(row + 1) * n1_] = 0; r1[row * n1_:(row + 1) * n1_] = 0
        a, b = u["timf3"]
        idx = (a + np.arange(2 * ((b - a) % h3.size))) % h3.size
        h3[idx] = 0; r3[idx] = 0
    rep = {"transforms": int(ref.shape[0]), "freq_err_bins": float(np.max(np.abs(got[:, 2] - ref[:, 2]))),
           "phase_err_rad": float(np.max(np.abs(wrap(got[:, 3] - ref[:, 3])))), "ampl_rel": float(np.max(np.abs(got[:, 6] - ref[:, 6]) / np.abs(ref[:, 6]))),
           "fft1": rel(h1, r1), "timf3": rel(h3, r3)}
    it = g["itrace"].reshape(-1, 16)
    keep = np.ones(out["sumsq"].size, bool)
    if it[-1, 10] != 0:                                          # an unfinished averaging period at the end: fft1_c accumulates it in place
        keep[it[-1, 9]:it[-1, 9] + (1 << out["cfg"].fft1_n)] = False
    rep["sumsq"] = rel(out["sumsq"] * keep, g["fft1_sumsq"] * keep)
    loc = int(ref[-1, 0])
    n1 = 1 << out["cfg"].fft1_n
    f_h, f_r = h1.reshape(-1, n1, 2)[:, loc:loc + 7].astype(np.float64), r1.reshape(-1, n1, 2)[:, loc:loc + 7].astype(np.float64)
    rep["residual_vs_carrier"] = float(np.linalg.norm(f_r) / (np.sqrt(f_r.shape[0]) * abs(ref[-1, 6])))
    rep["residual_err_vs_carrier"] = float(np.linalg.norm(f_h - f_r) / (np.sqrt(f_r.shape[0]) * abs(ref[-1, 6])))
    assert rep["freq_err_bins"] < 1e-3 and rep["phase_err_rad"] < 20 * tol * 1e2 and rep["ampl_rel"] < 10 * tol, rep
    assert rep["fft1"] < tol and rep["sumsq"] < tol and rep["timf3"] < tol and rep["residual_err_vs_carrier"] < tol, rep
    assert rep["residual_vs_carrier"] < 0.1, rep                # the carrier is gone from the ring
    sp, thr, done, cnt = out["ss"]
    a, b = out["ss_range"]
    ref_sp, info = g["spursearch_spectrum"][a:b + 1], g["spursearch_info"]
    assert done == int(info[0]) and cnt == int(info[1]) and done >= 1, (done, cnt, info)
    assert abs(thr - g["spursearch_thresholds"][min(done, 64) - 1]) <= 10 * tol * thr
    assert np.array_equal(sp < 0, ref_sp < 0), "wiped peaks differ"
    rep["search_spectrum_err"] = float(np.max(np.abs(sp - ref_sp)) / float(np.max(ref_sp)))
    assert rep["search_spectrum_err"] < tol and np.count_nonzero((sp == 0) != (ref_sp == 0)) <= 2, rep
    return rep


def compare(out, g, tol, batch=1):
    if "fft1" in out:
        return compare_fft1(out, g, tol)
    ref = g["spur_trace"].reshape(-1, 12)
    got = out["trace"]
    if batch > 1:                                               # state is visible after every call only
        ref = ref[np.isin(ref[:, 9], got[:, 9])]
    assert got.shape[0] == ref.shape[0] and got.shape[0] > 10
    assert np.array_equal(got[:, :2], ref[:, :2]), "spur_location / spur_flag trace differs"
    rep = {"transforms": int(ref.shape[0]), "locations": sorted(set(int(x) for x in ref[:, 0]))}

    def wrap(x):
        return (x + np.pi) % (2 * np.pi) - np.pi
    rep["freq_err_bins"] = float(np.max(np.abs(got[:, 2] - ref[:, 2])))
    rep["phase_err_rad"] = float(np.max(np.abs(wrap(got[:, 3] - ref[:, 3]))))
    rep["d1_err"] = float(np.max(np.abs(wrap(got[:, 4] - ref[:, 4]))))
    rep["d2_err"] = float(np.max(np.abs(got[:, 5] - ref[:, 5])))
    rep["ampl_rel"] = float(np.max(np.abs(got[:, 6] - ref[:, 6]) / np.abs(ref[:, 6])))
    rep["noise_rel"] = float(np.max(np.abs(got[:, 7] - ref[:, 7]) / np.abs(ref[:, 7])))

    def rel(a, b):
        a, b = a.astype(np.float64), b.astype(np.float64)
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    h2, r2, h3, r3 = out["fft2"], g["fft2_float"], out["timf3"], g["timf3_float"]
    if out.get("unremoved"):                                    # handed over, not acquired: see run()
        n2_, u = 2 << out["cfg"].fft2_n, out["unremoved"]
        h2, r2, h3, r3 = h2.copy(), r2.copy(), h3.copy(), r3.copy()
        for row in u["rows"]:
            h2[row * n2_:(row + 1) * n2_] = 0; r2[row * n2_:(row + 1) * n2_] = 0
        a, b = u["timf3"]
        idx = (a + np.arange(2 * ((b - a) % h3.size))) % h3.size   # (+ the half the next call overlaps onto it)
        h3[idx] = 0; r3[idx] = 0
    rep["fft2"], rep["ps2"], rep["timf3"] = rel(h2, r2), rel(out["ps2"], g["fft2_powersum_float"]), rel(h3, r3)
    # the bins of the spur itself in the newest transforms: what is left after the subtraction, against the reference's residual
    cfg = out["cfg"]
    n2 = 1 << cfg.fft2_n
    loc = int(ref[-1, 0])
    f_h, f_r = h2.reshape(cfg.max_fft2n, n2, 2), r2.reshape(cfg.max_fft2n, n2, 2)
    res_h, res_r = f_h[:, loc:loc + 7].astype(np.float64), f_r[:, loc:loc + 7].astype(np.float64)
    rep["residual_vs_carrier"] = float(np.linalg.norm(res_r) / (np.sqrt(cfg.max_fft2n) * abs(ref[-1, 6])))
    rep["residual_err_vs_carrier"] = float(np.linalg.norm(res_h - res_r) / (np.sqrt(cfg.max_fft2n) * abs(ref[-1, 6])))
    assert rep["freq_err_bins"] < 1e-3 and rep["phase_err_rad"] < 20 * tol * 1e2 and rep["ampl_rel"] < 10 * tol, rep
    assert rep["fft2"] < tol and rep["ps2"] < tol and rep["timf3"] < tol, rep
    assert rep["residual_err_vs_carrier"] < tol, rep
    if "spursearch_spectrum" in g and batch == 1:
        # the search spectrum spursearch_spectrum_cleanup left last (spursub.c:40-175): the same peaks wiped (the -1e-8 marks), the same bins
        # clamped to zero, what remains to float32 rounding of the three-spur_speknum sums; threshold, count of finished spectra, running counter
        sp, thr, done, cnt = out["ss"]
        a, b = out["ss_range"]
        ref_sp = g["spursearch_spectrum"][a:b + 1]
        info = g["spursearch_info"]
        assert done == int(info[0]) and cnt == int(info[1]) and done >= 1, (done, cnt, info)
        assert abs(thr - g["spursearch_thresholds"][min(done, 64) - 1]) <= 10 * tol * thr
        assert np.array_equal(sp < 0, ref_sp < 0) and np.any(ref_sp < 0), "wiped peaks differ"
        scale = float(np.max(ref_sp))
        rep["search_spectrum_err"] = float(np.max(np.abs(sp - ref_sp)) / scale)
        rep["search_zero_mismatch"] = int(np.count_nonzero((sp == 0) != (ref_sp == 0)))
        assert rep["search_spectrum_err"] < tol and rep["search_zero_mismatch"] <= 2, rep
    return rep


def compare_acquisition(out, g):
    """the loop state right after the lock against the reference's own (spur_init_state of the golden)"""
    st, a = g["spur_init_state"], out["acq"]

    def wrap(x):
        return (x + np.pi) % (2 * np.pi) - np.pi
    rep = {"location": (a.spur_location, int(st[0])), "freq_err_bins": abs(a.spur_freq - st[2]), "phase_err": abs(wrap(a.spur_d0pha - st[3])),
           "d1_err": abs(wrap(a.spur_d1pha - st[4])), "d2_err": abs(a.spur_d2pha - st[5]), "ampl_rel": abs(a.spur_ampl - st[6]) / abs(st[6]),
           "noise_rel": abs(a.spur_noise - st[7]) / abs(st[7])}
    assert a.spur_location == int(st[0]) and a.spur_flag == 0
    assert rep["freq_err_bins"] < 1e-4 and rep["phase_err"] < 1e-3 and rep["d1_err"] < 1e-4 and rep["d2_err"] < 1e-5 and rep["ampl_rel"] < 1e-4 and rep["noise_rel"] < 1e-2, rep
    return rep
