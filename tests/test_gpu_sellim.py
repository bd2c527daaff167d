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


@pytest.mark.parametrize("fft1_n,fft2_n,par1", [(14, 16, 2), (15, 17, 2), (14, 16, 1), (15, 17, 1), (14, 16, 0), (15, 17, 0)])
def test_fullsize_limiters_match_oracle(fft1_n, fft2_n, par1):
    """fft1_size 16384 / fft2_size 65536 (BASELINE sizes; the limiter kernels' LDS layout, bit words and group loops at their real
    extent) and fft1_size 32768 / fft2_size 131072 (the reference's maximum: table and group minima in global memory, dense routing
    bits for the four-step make_timf2): both limiters inside lrh_wideband_dsp on the bench's synthetic signal, HIP against the oracle
    -- same table after the run, same amplitude factor, same pointers, timf2 to the north-star tolerance"""
    import numpy as np
    from linrad_amd import abi
    from linrad_amd.abi import default_sellim
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    from linrad_amd.workload import chain_config
    from oracle_binding import open_oracle
    n1, nblk, batch = 1 << fft1_n, 96, 16
    cfg = chain_config(fft1_n, fft2_n, batch=batch, rounds=nblk // batch)
    s = synth_defaults(n1, 0)
    iq = synth_iq(s, 0, cfg.timf1_bytes // 4)
    res = []
    for fn in (open_hip, open_oracle):
        rx = fn(cfg)
        rx.timf1_write(iq)
        rx.set_mix1_selfreq(0.31 * (1 << fft2_n) + 0.3)
        par = default_sellim(cfg, fft1_blocktime=(n1 // 2) / 160e6, blanker_ston_fft1=30.0, blanker_ston_fft2=30.0, fft2_blocktime=(1 << fft2_n) / 2 / 160e6, exact_stats=1,
                             sellim_par1=par1)      # hg.sellim_par1: the second limiter's three variants (sellim.c:169, 283, 535)
        rx.wideband_limiter(par, True)
        rx.wideband_dsp(nblk, batch)
        res.append(dict(lim=rx.get_liminfo(), amp=rx.liminfo_amplitude_factor(), p=rx.p.as_dict(), timf2=rx.export(abi.RING_TIMF2_FLOAT),
                        bs=rx.blanker_state()))
        rx.close()
    h, o = res
    ints = [k for k, v in h["p"].items() if isinstance(v, int)]
    assert {k: h["p"][k] for k in ints} == {k: o["p"][k] for k in ints}
    strong = int(np.count_nonzero(o["lim"]))
    mism = int(np.sum(np.sign(h["lim"]) != np.sign(o["lim"])))
    print("strong bins", strong, "pattern mismatches", mism, "amp", h["amp"], o["amp"], "attenuated", int(np.sum(o["lim"] > 0)))
    assert strong > 20 and mism == 0 and abs(h["amp"] - o["amp"]) <= 1e-6
    pos = o["lim"] > 0
    if pos.any():
        assert np.max(np.abs(h["lim"][pos] - o["lim"][pos]) / o["lim"][pos]) <= 1e-5
    keep = np.ones(h["timf2"].size, bool)
    keep[(h["p"]["timf2_pa"] + np.arange(4 * (n1 // 2))) % keep.size] = False
    flips = np.nonzero((((h["timf2"][0::4] == 0) & (h["timf2"][1::4] == 0)) != ((o["timf2"][0::4] == 0) & (o["timf2"][1::4] == 0))) & keep[0::4])[0]
    for i in flips[:8]:
        keep[4 * i:4 * i + 4] = False
    assert len(flips) <= 8
    err = np.linalg.norm((h["timf2"].astype(np.float64) - o["timf2"]) * keep) / np.linalg.norm(o["timf2"] * keep)
    print("timf2", err, "flips", len(flips))
    assert err < 1e-5


def test_wideband_limiter_is_refused_on_a_coupled_context():
    """dsp_coupled, the round-level order of two coupled channels, makes no limiter calls: installing one there must fail instead of being
    silently ignored (the reference's two-channel limiter works on the summed spectra, not built)"""
    from linrad_amd.abi import LrhError, default_sellim
    from linrad_amd.lib import open_hip
    from linrad_amd.workload import chain_config
    cfg = chain_config(10, 12, batch=4)
    cfg.blanker_channels, cfg.timf1_frame_channels, cfg.timf1_channel_index = 2, 2, 0
    rx = open_hip(cfg)
    with pytest.raises((LrhError, RuntimeError)):
        rx.wideband_limiter(default_sellim(cfg), False)
    rx.wideband_limiter(None)
    rx.close()
