"""GPU: the linear ("clever") blanker on the device (k_clever_prep / k_clever ahead of the stupid blanker inside
lrh_first_noise_blanker) against the compiled reference, which built the pulse tables itself (goldens tests/golden/clever_*.npz)."""
import numpy as np
import pytest

import cleverlib
from refcases import CLEVER

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("serial", [0, 1])
@pytest.mark.parametrize("name", list(CLEVER))
def test_hip_clever_blanker_matches_reference(name, serial, monkeypatch):
    """serial = 0: the region-parallel replay (one wave per quiet-gap-separated region, extents checked); serial = 1: the check is
    made to fail (LRH_CLEVER_SERIAL), so the span is restored from the backup and replayed by one wave in the reference's order"""
    from linrad_amd.lib import open_hip
    if serial:
        monkeypatch.setenv("LRH_CLEVER_SERIAL", "1")
    g = cleverlib.load(name)
    out = cleverlib.run(open_hip, name, g)
    rep = cleverlib.compare(out, g, 1e-5)
    nser = out["api"].blanker_state().clever_serial_calls
    print(name, "serial" if serial else "regions", rep, "one-wave replays", nser)
    assert nser >= 20 if serial else nser <= 2          # the blanker runs every third block or so (rate limit, blank1.c:712)
    out["api"].close()


def test_clever_tables_errors_and_off():
    from linrad_amd import abi
    from linrad_amd.abi import LrhError
    from linrad_amd.lib import open_hip
    from refcases import clever_case, lrh_config
    name = "clever_n10_n12"
    g = cleverlib.load(name)
    d, cl, iq, lim, des = clever_case(name)
    rx = open_hip(lrh_config(d, iq))                          # blnfit_range / pulsewidth of the uncalibrated default
    with pytest.raises((LrhError, RuntimeError), match=f"rc={abi.LRH_EINVAL}"):
        cleverlib.install_tables(rx, g, d["noise_floor"])
    rx.set_blanker_tables()                                  # off: accepted
    rx.close()


@pytest.mark.parametrize("deferred", [False, True])
def test_fullsize_clever_blanker_matches_oracle(monkeypatch, deferred):
    """fft1_size 16384 (BASELINE sizes), the bench's synthetic signal plus band-limited pulses of the calibrated response: the tables
    the reference built for the same fractional passband (golden clever_n10_n12; the response in samples does not depend on
    fft1_size), HIP vs oracle through lrh_wideband_dsp: same resume pointers, same fitted / rejected pulses, same rings.
    deferred: all rounds in one call on the one-round-late two-stream schedule -- the search of a round is issued a round late and its
    resume point comes back through a pinned slot when the next round's blanker call needs it."""
    if deferred:
        monkeypatch.setenv("LRH_PIPELINE", "2")              # forced: the automatic choice keeps rounds this small on the serial order
    from linrad_amd import abi
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    from linrad_amd.workload import chain_config, strong_liminfo
    from oracle_binding import open_oracle
    from refcases import clever_desired
    g = cleverlib.load("clever_n10_n12")
    N1, nblk, batch = 16384, 48, 16
    cfg = chain_config(14, 12, batch=batch, rounds=nblk // batch)
    cfg.blanker_pulsewidth, cfg.blnfit_range = int(g["bln_ints"][1]), int(g["bln_ints"][3])
    while cfg.timf2pow_size < 2 * nblk * (N1 // 2):          # rings that hold the whole run (block bookkeeping below)
        cfg.timf2pow_size *= 2
    while cfg.max_fft2n * 2048 < 2 * nblk * (N1 // 2):
        cfg.max_fft2n *= 2
    while cfg.timf3_size < 4 * nblk * (N1 // 2) // 64 * 2:
        cfg.timf3_size *= 2
    s = synth_defaults(N1, 0)
    s.pulse_period = 0
    iq = synth_iq(s, 0, cfg.timf1_bytes // 4).astype(np.float64)
    n = iq.size // 2
    rng = np.random.default_rng(77)
    des = clever_desired(14, 0.18).astype(np.float64)
    spec, k, H = np.fft.ifftshift(des), np.fft.fftfreq(N1), 256
    norm = np.abs(np.fft.ifft(spec)[0])
    z = np.zeros(n, complex)
    for pos in np.sort(rng.choice(np.arange(4 * N1, min(n, (nblk + 2) * N1 // 2) - 4 * N1, 64), 150, replace=False)):
        h = np.fft.ifft(spec * np.exp(-2j * np.pi * k * (rng.uniform(-0.5, 0.5) + H)))[:2 * H] / norm
        z[pos - H:pos + H] += np.exp(rng.uniform(np.log(2500.0), np.log(22000.0))) * np.exp(1j * rng.uniform(0, 6.28)) * h
    iq[0::2] += z.real
    iq[1::2] += z.imag
    iq = np.clip(np.round(iq), -32767, 32767).astype(np.int16)
    lim = strong_liminfo(s, 14)
    res = []
    for fn in (open_hip, open_oracle):
        rx = fn(cfg)
        rx.timf1_write(iq)
        rx.set_liminfo(lim)
        rx.set_mix1_selfreq(0.31 * 4096 + 0.3)
        cleverlib.install_tables(rx, g, cfg.timf2_noise_floor)
        if fn is open_hip and not deferred:
            rx.profile_enable(True)
        tot = [0, 0]
        if deferred and fn is open_hip:
            rx.wideband_dsp(nblk, batch)
        for _ in range(0 if deferred and fn is open_hip else nblk // batch):
            rx.wideband_dsp(batch, batch)
            st = rx.blanker_state()
            tot[0] += st.last_call_fitted
            tot[1] += st.last_call_rejected
        r = dict(p=rx.p.as_dict(), bs=rx.blanker_state(), tot=tot, timf2=rx.export(abi.RING_TIMF2_FLOAT), pwr=rx.export(abi.RING_TIMF2_PWR),
                 timf3=rx.export(abi.RING_TIMF3_FLOAT))
        if fn is open_hip:
            r["prof"] = {kk: rx.profile_get(kk) for kk in ("clever", "blanker")} if not deferred else None
        res.append(r)
        rx.close()
    h, o = res
    print("fitted / rejected", h["tot"], o["tot"], "stage ms (total, launches)", h["prof"], "slow-path calls", h["bs"].slow_path_calls, "one-wave replays", h["bs"].clever_serial_calls)
    ints = [kk for kk, v in h["p"].items() if isinstance(v, int)]
    assert {kk: h["p"][kk] for kk in ints} == {kk: o["p"][kk] for kk in ints}
    assert (deferred or h["tot"] == o["tot"]) and o["tot"][0] > 50
    assert h["bs"].last_call_fitted == o["bs"].last_call_fitted and h["bs"].last_call_rejected == o["bs"].last_call_rejected
    assert h["bs"].clever_bln_limit == o["bs"].clever_bln_limit and h["bs"].timf2_fitted_pulses == o["bs"].timf2_fitted_pulses

    def rel(a, b):
        return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64)))
    keep = np.ones(h["timf2"].size, bool)
    keep[(h["p"]["timf2_pa"] + np.arange(4 * (N1 // 2))) % keep.size] = False
    # a sample within float32 rounding of the stupid limit may be cleared on one side only (see fullsize_compare): few, borderline,
    # and the comparison then leaves out the samples / output blocks such a flip reaches -- at the same tolerance
    limit = float(o["bs"].stupid_bln_limit)
    flips = np.nonzero(((h["pwr"] == 0) != (o["pwr"] == 0)) & keep[::4])[0]
    assert len(flips) <= 4 and all(abs(max(h["pwr"][i], o["pwr"][i]) - limit) <= 1e-3 * limit for i in flips), flips
    for i in flips:
        keep[4 * i:4 * i + 4] = False
    N2, M2 = 4096, 2048
    Mm = (N2 >> cfg.mix1_bandwidth_reduction_n) // 2
    ntr = o["p"]["fft2_na"]
    assert o["p"]["timf2_px"] == 4 * ntr * M2 and o["p"]["timf3_pa"] == 2 * Mm * ntr, "rings wrapped: enlarge them for this test"
    hit = np.zeros(ntr + 2, bool)
    for i in flips:
        t1 = min(ntr - 1, i // M2)
        hit[max(0, t1 - 1):t1 + 1] = True
    ok3 = np.array([not (hit[t] or (t > 0 and hit[t - 1])) for t in range(ntr)])
    t3h, t3o = h["timf3"][:2 * Mm * ntr].reshape(ntr, -1), o["timf3"][:2 * Mm * ntr].reshape(ntr, -1)
    errs = dict(timf2=rel(h["timf2"] * keep, o["timf2"] * keep), pwr=rel(h["pwr"] * keep[::4], o["pwr"] * keep[::4]), timf3=rel(t3h[ok3], t3o[ok3]),
                flips=int(len(flips)), timf3_blocks_compared=int(ok3.sum()), timf3_blocks=int(ntr))
    print(errs)
    assert errs["timf2"] < 1e-5 and errs["timf3"] < 1e-5 and errs["pwr"] < 1e-4


def test_bench_shape_clever_blanker_matches_oracle():
    """The linear blanker at the call size bench.py --clever times: one call of 4096 fft1 blocks (33.5 M samples, 2000 calibrated pulses
    among them) after a 64-block call that leaves both sides the same noise floor and limits; the search rides the one-round-late
    schedule (it is issued when the round is flushed).  HIP against the oracle: resume pointers, fitted / rejected pulses, limits, and the
    timf2 ring to the north-star tolerance outside the few samples within float32 rounding of the stupid limit."""
    from linrad_amd import abi
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    from linrad_amd.workload import chain_config, strong_liminfo
    from oracle_binding import open_oracle
    from refcases import clever_desired
    g = cleverlib.load("clever_n10_n12")
    N1, batch, warm = 16384, 4096, 64
    cfg = chain_config(14, 16, batch=batch, fft3_n=12, mix2_n=8, rounds=2)
    cfg.blanker_pulsewidth, cfg.blnfit_range = int(g["bln_ints"][1]), int(g["bln_ints"][3])
    s = synth_defaults(N1, 0)
    s.pulse_period = 0
    iq = synth_iq(s, 0, cfg.timf1_bytes // 4).astype(np.float32)
    n = iq.size // 2
    rng = np.random.default_rng(78)
    des = clever_desired(14, 0.18).astype(np.float64)
    spec, k, H = np.fft.ifftshift(des), np.fft.fftfreq(N1), 256
    norm = np.abs(np.fft.ifft(spec)[0])
    hi = min(n, (warm + batch + 2) * N1 // 2) - 4 * N1
    for pos in np.sort(rng.choice(np.arange(4 * N1, hi, 64), 2000, replace=False)):
        h = np.fft.ifft(spec * np.exp(-2j * np.pi * k * (rng.uniform(-0.5, 0.5) + H)))[:2 * H] / norm
        z = np.exp(rng.uniform(np.log(2500.0), np.log(22000.0))) * np.exp(1j * rng.uniform(0, 6.28)) * h
        iq[2 * (pos - H):2 * (pos + H):2] += z.real.astype(np.float32)
        iq[2 * (pos - H) + 1:2 * (pos + H) + 1:2] += z.imag.astype(np.float32)
    iq = np.clip(np.round(iq), -32767, 32767).astype(np.int16)
    lim = strong_liminfo(s, 14)
    res = []
    for fn, sparse in ((open_hip, 1), (open_oracle, 0)):
        cfg.fft1_float_sparse = cfg.fft2_float_sparse = sparse
        rx = fn(cfg)
        rx.timf1_write(iq)
        rx.set_liminfo(lim)
        rx.set_mix1_selfreq(0.31 * 65536 + 0.3)
        rx.wideband_dsp(warm, warm)
        first = rx.blanker_state()
        cleverlib.install_tables(rx, g, first.timf2_noise_floor)
        rx.wideband_dsp(batch, batch)
        bs = rx.blanker_state()
        res.append(dict(p=rx.p.as_dict(), bs=bs, first=first, timf2=rx.export(abi.RING_TIMF2_FLOAT), pwr=rx.export(abi.RING_TIMF2_PWR)))
        rx.close()
    h, o = res
    print("fitted / rejected", (h["bs"].last_call_fitted, h["bs"].last_call_rejected), (o["bs"].last_call_fitted, o["bs"].last_call_rejected),
          "one-wave replays", h["bs"].clever_serial_calls, "slow-path calls", h["bs"].slow_path_calls)
    ints = [kk for kk, v in h["p"].items() if isinstance(v, int)]
    assert {kk: h["p"][kk] for kk in ints} == {kk: o["p"][kk] for kk in ints}
    assert h["first"].timf2_noise_floor == o["first"].timf2_noise_floor and h["first"].stupid_bln_limit == o["first"].stupid_bln_limit
    assert o["bs"].last_call_fitted > 500
    assert h["bs"].last_call_fitted == o["bs"].last_call_fitted and h["bs"].last_call_rejected == o["bs"].last_call_rejected
    assert h["bs"].timf2_fitted_pulses == o["bs"].timf2_fitted_pulses
    keep = np.ones(h["timf2"].size, bool)
    keep[(h["p"]["timf2_pa"] + np.arange(4 * (N1 // 2))) % keep.size] = False
    limit = float(o["first"].stupid_bln_limit)               # in force during the big call (the ring holds nothing older)
    flips = np.nonzero(((h["pwr"] == 0) != (o["pwr"] == 0)) & keep[::4])[0]
    margin = [abs(max(float(h["pwr"][i]), float(o["pwr"][i])) - limit) / limit for i in flips]
    print("flips", len(flips), "largest margin", max(margin, default=0.0))
    # a flip is a sample whose power sits within float32 rounding of the limit on the two sides -- or a guard sample: with the pulse
    # calibration the blanker also clears (int)(clr sqrt(pulmax / noise) / 100 + .5) samples before and after a run (blank1.c:1049-1083,
    # clr = (pulsewidth + 1) / 2 resp. pulsewidth + 1), a length that itself rounds at a boundary; the guard sample's own power may be
    # anywhere.  Every flip must be borderline itself or lie within a guard's reach of a sample both sides cleared.
    both = (h["pwr"] == 0) & (o["pwr"] == 0)
    reach = cfg.blanker_pulsewidth + 2
    bad = [(int(i), m) for i, m in zip(flips, margin) if m > 1e-4 and not both[max(0, int(i) - reach):int(i) + reach + 1].any()]
    for i, m in bad[:4]:                                      # context of an unexplained flip, for the log
        print("flip", i, m, "limit", limit, "\n hip", np.round(h["pwr"][i - 8:i + 9], 1).tolist(), "\n ora", np.round(o["pwr"][i - 8:i + 9], 1).tolist(),
              "\n hip weak", np.round(h["timf2"].reshape(-1, 4)[i - 3:i + 4, :2], 2).tolist(), "\n ora weak", np.round(o["timf2"].reshape(-1, 4)[i - 3:i + 4, :2], 2).tolist())
    # What is left: a pulse the linear blanker fitted and took out on one side and rejected on the other (its accept / reject tests compare
    # sums of residues with thresholds, blank1.c:161-232, 925-990: a borderline case flips with the float32 noise of the transform; the
    # pulse counts above can still agree when another borderline pulse flips the other way).  The rejected pulse's peak then stands above
    # the limit on that side alone and the stupid blanker clears it -- an isolated sample.  Seen: 1 in 33.5 M samples with k_fft1v, 0
    # with k_fft1w; more than a handful, or neighbours flipping along, would be a fault.
    flipset = set(int(i) for i in flips)
    assert len(bad) <= 4 and all((i - 1) not in flipset and (i + 1) not in flipset for i, _ in bad), bad[:12]
    assert len(flips) <= 2 * (o["pwr"].size // 400000 + 1)
    for i, _ in bad:                                          # ... and what the fit left behind differs around it: out of the ring comparison below
        keep[4 * (i - 128):4 * (i + 129)] = False
    for i in flips:
        keep[4 * i:4 * i + 4] = False
    err = float(np.linalg.norm((h["timf2"].astype(np.float64) - o["timf2"]) * keep) / np.linalg.norm(o["timf2"] * keep))
    print("timf2", err)
    assert err < 1e-5
