"""GPU: the selective limiter on the device-resident spectra (lrh_fft1_update_liminfo -> k_sellim + k_pack_liminfo) against the
compiled reference's liminfo after every update, and the rings its routing shapes."""
import numpy as np
import pytest

import sellimlib
from refcases import SELLIM

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(SELLIM))
def test_hip_selective_limiter_matches_reference(name):
    from linrad_amd.lib import open_hip
    g = sellimlib.load(name)
    rep = sellimlib.compare(sellimlib.run(open_hip, name, g), g, tol=1e-5, value_tol=1e-5)
    print(name, rep)
    assert rep["cleared_equal"]


def test_limiter_statistic_may_lag():
    """exact_stats = 0: nothing waits for the device; the routing is the same, the weak-bin count make_timf2 reports is that of the
    newest update whose readback has arrived"""
    from linrad_amd.lib import open_hip
    from linrad_amd.abi import default_sellim
    from refcases import lrh_config, sellim_case
    name = "sellim_n10_n12"
    g = sellimlib.load(name)
    d, _, iq = sellim_case(name)
    cfg = lrh_config(d, iq)
    outs = []
    for exact in (1, 0):
        rx = open_hip(cfg)
        rx.timf1_write(iq)
        rx.set_mix1_selfreq(d["fq"])
        par = sellimlib.sellim_params(cfg, g)
        par.exact_stats = exact
        lows, tabs, cnt = [], [], 0
        for b in range(60):
            rx.fft1_b(1), rx.fft1_c(1), rx.make_timf2(1)
            lows.append(rx.p.fft1_lowlevel_points)
            if rx.p.fft1_liminfo_cnt != cnt:
                rx.fft1_update_liminfo(par)
                cnt = rx.p.fft1_liminfo_cnt
                tabs.append(rx.get_liminfo())
        outs.append((np.array(lows), np.array(tabs), rx.export(3)))
    (l1, t1, r1), (l0, t0, r0) = outs
    assert np.array_equal(t1, t0) and np.array_equal(r1, r0)           # tables and the timf2 ring: identical
    # the count: never ahead of the exact run, and the same values in the same order (the newest finished update is installed whenever
    # make_timf2 or the next update looks; how late that is depends on how far the host runs ahead of the device)
    def distinct(a):
        return [int(v) for k, v in enumerate(a) if k == 0 or v != a[k - 1]]
    it = iter(distinct(l1))
    assert all(v in it for v in distinct(l0)), (distinct(l0), distinct(l1))
    assert all(l0[i] in l1[:i + 1] for i in range(len(l0)))


@pytest.mark.parametrize("pipeline", ["0", "1", "2"])
def test_hip_wideband_dsp_makes_the_limiter_calls_itself(pipeline, monkeypatch):
    """lrh_wideband_limiter: the limiter kernels at the end of every round inside lrh_wideband_dsp, in each of its schedules
    (LRH_PIPELINE 0 serial, 1 two streams, 2 one round late -- which the second limiter switches off), against the oracle doing the same
    and against the caller making the calls between single-round calls"""
    import numpy as np
    from linrad_amd.lib import open_hip
    from oracle_binding import open_oracle
    monkeypatch.setenv("LRH_PIPELINE", pipeline)
    name = "sellim2_n10_n12"
    g = sellimlib.load(name)
    for fft2_too in (True, False):
        h = sellimlib.run_dsp(open_hip, name, g, in_call=True, fft2_too=fft2_too)
        e = sellimlib.run_dsp(open_hip, name, g, in_call=False, fft2_too=fft2_too)
        o = sellimlib.run_dsp(open_oracle, name, g, in_call=True, fft2_too=fft2_too)
        assert h["p"] == o["p"] and np.array_equal(np.sign(h["lim"]), np.sign(o["lim"])) and abs(h["amp"] - o["amp"]) <= 1e-6
        assert np.array_equal(h["lim"], e["lim"]) and np.array_equal(h["timf2"], e["timf2"]) and np.array_equal(h["timf3"], e["timf3"])
        keep = np.ones(h["timf2"].size, bool)
        keep[(h["p"]["timf2_pa"] + np.arange(4 * 512)) % keep.size] = False
        err = np.linalg.norm((h["timf2"] - o["timf2"]) * keep) / np.linalg.norm(o["timf2"] * keep)
        assert err < 1e-5 and np.count_nonzero(h["lim"]) > 50, (err, fft2_too)
