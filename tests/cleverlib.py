"""Linear ("clever") blanker parity machinery shared by the oracle (CPU) and HIP (GPU) tests: install the tables the compiled
reference's init_blanker built (stored in the golden), drive the case block by block in the harness's order and compare the
blanker's scalars after every call and the rings behind with the golden."""
import os

import numpy as np

from linrad_amd import abi
from refcases import clever_case, lrh_config

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))


def install_tables(api, g, noise_floor):
    bi, bf = g["bln_ints"], g["bln_fparams"]
    api.set_blanker_tables(bln=g["bln"].reshape(-1, 4)[:, :3], refpulse=g["blanker_refpulse"], phasefunc=g["blanker_phasefunc"],
                           pulindex=g["blanker_pulindex"], largest_blnfit=int(bi[2]), clever_bln_factor=float(bf[1]),
                           clever_bln_limit=int(np.float32(noise_floor) * np.float32(bf[1])), liminfo_amplitude_factor=float(bf[0]))


def run(open_fn, name, g, case=None):
    """case: (d, cl, iq, lim, des) of another input on the golden's tables (the random tests); None: the golden's own case"""
    d, cl, iq, lim, des = case if case is not None else clever_case(name)
    assert case is not None or np.array_equal(iq, g["iq"])
    bi = g["bln_ints"]
    d = dict(d, pulsewidth=int(bi[1]), blnfit_range=int(bi[3]))
    cfg = lrh_config(d, iq)
    api = open_fn(cfg)
    par, cnt, amp = None, 0, []
    if cl["sellim"]:                                         # the limiter builds the routing table as the run goes (harness sellim=1)
        import sellimlib
        par = sellimlib.sellim_params(cfg, g)
    else:
        api.set_liminfo(lim)
    install_tables(api, g, d["noise_floor"])
    api.timf1_write(iq)
    api.set_mix1_selfreq(d["fq"])
    rows = []
    for b in range(d["nblk"]):
        api.fft1_b(1), api.fft1_c(1), api.make_timf2(1)
        api.first_noise_blanker()
        for _ in range(api.fft2_available()):
            api.make_fft2(1)
            api.fft2_mix1_fixed(1)
        if par is not None and api.p.fft1_liminfo_cnt != cnt:      # wcw.c:1124-1128
            api.fft1_update_liminfo(par)
            cnt = api.p.fft1_liminfo_cnt
            amp.append(api.liminfo_amplitude_factor())
        st = api.blanker_state()
        rows.append([api.p.timf2_pa, api.p.timf2p_fit, api.p.timf2_pn2, st.timf2_cleared_points, api.p.timf2_blanker_points,
                     st.timf2_noise_floor, st.stupid_bln_limit, st.clever_bln_limit, st.timf2_fitted_pulses, st.last_call_fitted,
                     st.last_call_rejected])
    return dict(api=api, d=d, rows=np.array(rows, np.int64), rate=api.blanker_state().clever_blanker_rate, amp=np.array(amp, np.float32),
                timf2=api.export(abi.RING_TIMF2_FLOAT), pwr=api.export(abi.RING_TIMF2_PWR), timf3=api.export(abi.RING_TIMF3_FLOAT))


def compare(out, g, tol):
    it, tr = g["itrace"].reshape(-1, 16), g["trace"].reshape(-1, 16)
    rows = out["rows"]
    ref = np.stack([it[:, 0], it[:, 1], it[:, 2], it[:, 5], it[:, 6], it[:, 12], it[:, 13], tr[:, 6].astype(np.int64), tr[:, 8].astype(np.int64)], 1)
    names = ["timf2_pa", "timf2p_fit", "timf2_pn2", "cleared_points", "blanker_points", "noise_floor", "stupid_limit", "clever_limit", "fitted_pulses"]
    rep = {"calls": int(rows.shape[0]), "fitted_total": int(rows[:, 9].sum()), "rejected_total": int(rows[:, 10].sum())}
    for j, nm in enumerate(names):
        bad = np.nonzero(rows[:, j] != ref[:, j])[0]
        rep[nm + "_first_diff"] = None if bad.size == 0 else (int(bad[0]), int(rows[bad[0], j]), int(ref[bad[0], j]))
    assert all(rep[nm + "_first_diff"] is None for nm in names), rep

    def rel(a, b):
        a, b = a.astype(np.float64), b.astype(np.float64)
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    n1 = out["api"].N1
    keep = np.ones(out["timf2"].size, bool)                   # sin^2 overlap: pending half beyond timf2_pa (paritylib)
    keep[(out["api"].p.timf2_pa + np.arange(4 * (n1 // 2))) % keep.size] = False
    rep["timf2"] = rel(out["timf2"] * keep, g["timf2_float"] * keep)
    rep["pwr"] = rel(out["pwr"] * keep[::4], g["timf2_pwr_float"] * keep[::4])
    rep["timf3"] = rel(out["timf3"], g["timf3_float"])
    rep["cleared_equal"] = bool(np.array_equal((out["pwr"] == 0) & keep[::4], (g["timf2_pwr_float"] == 0) & keep[::4]))
    if "amp_factor_trace" in g:                              # the factor the limiter hands the blanker after each of its updates
        ra = g["amp_factor_trace"][:len(out["amp"])]
        rep["amp_factors"] = sorted(set(np.round(ra, 4).tolist()))[:6]
        assert len(ra) == len(out["amp"]) and np.max(np.abs(out["amp"] - ra)) <= 1e-6 and np.any(ra > 1.0), rep
    assert rep["fitted_total"] > 0 and rep["rejected_total"] > 0, rep
    # timf3: relative tolerance, or the absolute float32 floor of the wide spectrum the band was cut from (paritylib.compare_with_golden)
    cfg = out["api"].cfg
    n2 = 1 << cfg.fft2_n
    nm = n2 >> cfg.mix1_bandwidth_reduction_n
    wide = np.linalg.norm(g["fft2_float"].astype(np.float64)) / np.sqrt(cfg.max_fft2n)
    floor = 4 * 6e-8 * wide * np.sqrt(nm / n2) * np.sqrt(out["timf3"].size / nm / 2) * np.sqrt(nm)
    err3 = float(np.linalg.norm(out["timf3"].astype(np.float64) - g["timf3_float"]))
    rep["timf3_abs"], rep["timf3_floor"], rep["timf3_escape"] = err3, float(floor), bool(rep["timf3"] > tol)
    assert rep["timf2"] <= tol and rep["pwr"] <= 10 * tol and (rep["timf3"] <= tol or err3 <= floor) and rep["cleared_equal"], rep
    return rep
