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
    """exact_stats = 0: nothing waits for the device; the routing is the same, the weak-bin count make_timf2 reports is the
    previous update's"""
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
    avg1 = cfg.fft_avg1num
    assert np.array_equal(l0[3 * avg1:], l1[avg1:-2 * avg1])           # the count: two updates late
