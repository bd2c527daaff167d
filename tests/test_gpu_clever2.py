"""GPU: the linear blanker on two coupled channels (k_clever's two-channel fit: get_pulse_pol, transform_timf2_pol,
subtract_twochan_pulse, blank1.c:232-609) against the compiled two-channel reference's goldens and against the oracle."""
import numpy as np
import pytest

import clever2lib
from refcases import CLEVER2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("frames_mode,serial", [(False, 0), (True, 0), (False, 1)])
@pytest.mark.parametrize("name", list(CLEVER2))
def test_hip_two_channel_clever_blanker_matches_reference(name, frames_mode, serial, monkeypatch):
    """serial = 0: the region-parallel replay; serial = 1 (LRH_CLEVER_SERIAL): the span restored from the backup -- both channels'
    samples, summed and own power -- and replayed by one wave"""
    from linrad_amd.lib import open_hip
    if serial:
        monkeypatch.setenv("LRH_CLEVER_SERIAL", "1")
    g = clever2lib.load(name)
    res = clever2lib.run(open_hip, name, g, frames_mode=frames_mode)
    rep = clever2lib.compare(res, g, 1e-5)
    nser = [rx.blanker_state().clever_serial_calls for rx in res["rxs"]]
    print(name, "serial" if serial else "regions", rep, "one-wave replays", nser)
    assert nser[0] == nser[1] and (nser[0] >= 20 if serial else nser[0] <= 2)
    for rx in res["rxs"]:
        rx.close()


def test_hip_two_channel_clever_blanker_follows_the_oracle_call_by_call():
    from linrad_amd.lib import open_hip
    from oracle_binding import open_oracle
    name = "clever2_n10"
    g = clever2lib.load(name)
    h = clever2lib.run(open_hip, name, g, frames_mode=False)
    o = clever2lib.run(open_oracle, name, g, frames_mode=False)
    assert np.array_equal(h["rows"][:, :, [0, 1, 2, 3, 4, 8, 9, 10]], o["rows"][:, :, [0, 1, 2, 3, 4, 8, 9, 10]])     # pointers, cleared, fitted and rejected per call
    n1 = h["rxs"][0].N1
    keep = np.ones(h["out"][0]["timf2"].size, bool)             # sin^2 overlap: the pending half beyond timf2_pa
    keep[(h["out"][0]["p"]["timf2_pa"] + np.arange(4 * (n1 // 2))) % keep.size] = False
    for ch in (0, 1):
        a, b = h["out"][ch]["timf2"].astype(np.float64) * keep, o["out"][ch]["timf2"].astype(np.float64) * keep
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-5


def test_fullsize_two_channel_clever_blanker_matches_oracle():
    """fft1_size 16384 (BASELINE size), 16 blocks per blanker call: the bench's synthetic signal on two channels plus band-limited
    pulses of the calibrated response with one polarisation; a pair of HIP contexts against a pair of oracle contexts, exchanges by
    hand.  Same resume pointers and pulse counts call by call, both channels' timf2 to the north-star tolerance."""
    import cleverlib
    from linrad_amd import abi
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    from linrad_amd.workload import chain_config, strong_liminfo
    from oracle_binding import open_oracle
    from refcases import clever_desired
    g = cleverlib.load("clever_n10_n12")
    N1, nblk, batch = 16384, 48, 16
    H = 256
    rng = np.random.default_rng(78)
    des = clever_desired(14, 0.18).astype(np.float64)
    spec, k = np.fft.ifftshift(des), np.fft.fftfreq(N1)
    norm = np.abs(np.fft.ifft(spec)[0])
    cfgs, iqs = [], []
    z = None
    for ch in (0, 1):
        cfg = chain_config(14, 12, batch=batch, rounds=nblk // batch)
        cfg.blanker_pulsewidth, cfg.blnfit_range = int(g["bln_ints"][1]), int(g["bln_ints"][3])
        cfg.blanker_channels, cfg.timf1_channel_index = 2, ch
        while cfg.timf2pow_size < 2 * nblk * (N1 // 2):
            cfg.timf2pow_size *= 2
        s = synth_defaults(N1, ch)
        s.pulse_period = 0
        iq = synth_iq(s, 0, cfg.timf1_bytes // 4).astype(np.float64)
        n = iq.size // 2
        if z is None:
            z = np.zeros(n, complex)
            for pos in np.sort(rng.choice(np.arange(4 * N1, min(n, (nblk + 2) * N1 // 2) - 4 * N1, 64), 150, replace=False)):
                h = np.fft.ifft(spec * np.exp(-2j * np.pi * k * (rng.uniform(-0.5, 0.5) + H)))[:2 * H] / norm
                z[pos - H:pos + H] += np.exp(rng.uniform(np.log(2500.0), np.log(22000.0))) * np.exp(1j * rng.uniform(0, 6.28)) * h
        zz = z if ch == 0 else 0.8 * np.exp(0.6j) * z
        iq[0::2] += zz.real
        iq[1::2] += zz.imag
        cfgs.append(cfg)
        iqs.append(np.clip(np.round(iq), -32767, 32767).astype(np.int16))
    lim = strong_liminfo(synth_defaults(N1, 0), 14)
    res = []
    for fn in (open_hip, open_oracle):
        rxs = []
        for ch in (0, 1):
            rx = fn(cfgs[ch])
            rx.timf1_write(iqs[ch])
            rx.set_liminfo(lim)
            cleverlib.install_tables(rx, g, cfgs[ch].timf2_noise_floor)
            rxs.append(rx)
        rows = []
        for _ in range(nblk // batch):
            for rx in rxs:
                rx.fft1_b(batch), rx.fft1_c(batch), rx.make_timf2(batch)
            clever2lib.blanker_round(rxs)
            st = [rx.blanker_state() for rx in rxs]
            rows.append([(rx.p.timf2p_fit, rx.p.timf2_pn2, b.last_call_fitted, b.last_call_rejected, b.last_call_cleared) for rx, b in zip(rxs, st)])
        res.append(dict(rows=rows, timf2=[rx.export(abi.RING_TIMF2_FLOAT) for rx in rxs], pwr=[rx.export(abi.RING_TIMF2_PWR) for rx in rxs],
                        pa=rxs[0].p.timf2_pa, bs=[rx.blanker_state() for rx in rxs]))
        for rx in rxs:
            rx.close()
    h, o = res
    print("per call (fit pointer, pn2, fitted, rejected, cleared):", h["rows"], o["rows"])
    assert all(r[0] == r[1] for r in h["rows"])                   # the two contexts of a pair agree
    fitted = sum(r[0][2] for r in h["rows"])
    assert fitted > 50 and sum(r[0][3] for r in h["rows"]) > 0
    assert [[c[:4] for c in r] for r in h["rows"]] == [[c[:4] for c in r] for r in o["rows"]]
    keep = np.ones(h["timf2"][0].size, bool)
    keep[(h["pa"] + np.arange(4 * (N1 // 2))) % keep.size] = False
    limit = float(o["bs"][0].stupid_bln_limit)
    hs, os_ = h["pwr"][0] + h["pwr"][1], o["pwr"][0] + o["pwr"][1]
    flips = np.nonzero(((hs == 0) != (os_ == 0)) & keep[::4])[0]
    assert len(flips) <= 4, flips
    for i in flips:
        keep[4 * i:4 * i + 4] = False
    for ch in (0, 1):
        a, b = h["timf2"][ch].astype(np.float64) * keep, o["timf2"][ch].astype(np.float64) * keep
        err = np.linalg.norm(a - b) / np.linalg.norm(b)
        print("channel", ch, "timf2", err, "flips", len(flips))
        assert err < 1e-5
