"""GPU: the chain at configurations nobody wrote down -- seeded random sizes, windows, averaging, blanker cadence, batch sizes, input formats and signals,
HIP against the oracle (float rings at 1e-5 relative RMS or held to the float64 build like everywhere else; pointer traces, averaging counters and --
where the noise floors agree -- the cleared-sample sets exact).  The goldens pin the restatement to the compiled reference at 21 hand-picked
configurations; this walks between them: 64 seeds in the suite (LRH_RANDOM_SEEDS for more; 200 were run in round 6).  First catch: the waterfall of a bin
with NO power (k_waterfall's log10(0), see there)."""
import numpy as np
import pytest

from paritylib import RINGS, relerr, run_case, truth_gate, waterfall_gate
from refcases import case_params, make_input, make_liminfo

pytestmark = pytest.mark.gpu


def _open_hip(cfg):
    from linrad_amd.lib import open_hip
    return open_hip(cfg)


def _open_oracle(cfg):
    from oracle_binding import open_oracle
    return open_oracle(cfg)


def _open_truth(cfg):
    from oracle_binding import open_truth
    return open_truth(cfg)


def random_case(seed):
    rng = np.random.default_rng(7000 + seed)
    d = case_params("n10_n12")                                  # every key with its default; overwritten below
    n1 = int(rng.integers(8, 13))
    second = rng.random() > 0.12
    n2 = int(np.clip(n1 + rng.integers(-2, 4), 8, 14)) if second else 10
    nx = n2 if second else n1
    d.update(n1=n1, n2=n2, second_fft=int(second), sinpow1=int(rng.choice([2, 2, 2, 0, 1, 3, 4])), sinpow2=int(rng.choice([2, 2, 2, 0, 3])),
             mixred=int(min(rng.integers(4, 7), nx - 3)), nblk=int(rng.integers(24, 72)), avg1num=int(rng.integers(2, 6)), avg2num=int(rng.integers(2, 5)),
             att_n=int(rng.integers(2, 7)), bln_interval=int(rng.integers(2, 5)), bln_avgnum=int(rng.choice([4, 8, 16])), wf_avgnum=int(rng.integers(1, 4)),
             wf_mode=int(rng.choice([1, 2, -2, 4])), seed=int(9000 + seed), timf2pow_log2=max(n1, n2) + 3, sumsq_blocks=8, lim_halfwidth=int(rng.integers(2, 5)),
             pulse_period=int(rng.choice([0, 997, 1999, 3001])), blockpower_block=0, gain=None, dword=int(rng.random() < 0.15), direction=int(rng.choice([1, 1, 1, -1])),
             sample_shift=0, real=0, fft3_n=0, mix2_n=0, afc=0, foldcorr_seed=0, golden_stride=1)
    N1, NX = 1 << n1, 1 << nx
    d["strong"] = [(float(rng.uniform(-0.4, 0.4) * N1), float(rng.choice([9000.0, 5000.0, 900.0]))) for _ in range(int(rng.integers(1, 4)))]
    d["weak"] = [(float(rng.uniform(-0.45, 0.45) * N1), float(rng.uniform(20, 70))) for _ in range(int(rng.integers(1, 3)))]
    d["fq"] = float(rng.uniform(0.15, 0.85) * NX)
    if second and NX > 1024:
        # the 1024 pixels of the waterfall around the strongest carrier: a window at the band's edge shows nothing but what leaks there from carriers 110 dB
        # up and outside it, and the 80 dB mask of the line gate below has nothing to hold on to (seed 558 of the third sweep: HIP 20-80 counts from the float64
        # value on pixels of 1e-4, the oracle 2)
        d["wf_centre_of"] = max(d["strong"], key=lambda c: c[1])[0]
    if d["sinpow1"] == 0:
        d["pulse_period"] = d["pulse_period"] or 997             # (no window: keep the blanker busy all the same)
    if seed >= 12:                                              # the extended sweep (LRH_RANDOM_SEEDS): the variants that have goldens of their own
        if second and rng.random() < 0.25 and n2 - d["mixred"] >= 6:
            d.update(fft3_n=int(min(8, n2 - d["mixred"] + 2)), fft3_sinpow=int(rng.choice([2, 2, 3])), mix2_n=6, max_fft3n=8, mix2=1, nblk=max(d["nblk"], 100))
            if d["fft3_n"] < 6 or d["mix2_n"] > d["fft3_n"]:
                d.update(fft3_n=0, mix2_n=0, mix2=0)
        if rng.random() < 0.12 and d["sinpow1"] in (0, 2) and d["direction"] > 0:
            d.update(real=1, dword=int(rng.random() < 0.3))
        if rng.random() < 0.08 and not d["real"]:
            d["sample_shift"] = int(rng.choice([-1, 1]))
    if seed >= 200:                                             # third sweep (LRH_RANDOM_SEEDS_EXT): the options the first two leave at their defaults
        r2 = np.random.default_rng(5500 + seed)
        if r2.random() < 0.35 and not d["real"]:
            d["foldcorr_seed"] = int(50 + seed)                  # I/Q mirror-image calibration (k_foldcorr behind the transform)
        if r2.random() < 0.35:
            d.update(afc=1, afc_bw=float(r2.choice([10.0, 20.0, 40.0])))   # mix1 follows a frequency that moves from transform to transform (fft?_mix1_afc)
        if second and r2.random() < 0.3:
            d["blockpower_block"] = 4 * int(r2.choice([64, 96, 128]))     # compute_timf2_powersum
        if r2.random() < 0.2:
            d["stupid"] = 0
        if r2.random() < 0.3:
            d["pulsewidth"] = int(r2.choice([1, 2, 3]))
    if "wf_centre_of" in d:                                     # (real input: carrier k of the half spectrum sits at bin |k| of N1, make_input)
        k0 = d.pop("wf_centre_of")
        c0 = abs(k0) * NX / N1 if d["real"] else NX // 2 + k0 * NX / N1
        d["wf_first"] = int(np.clip(c0 - 512, 0, NX - 1024))
    from refcases import level_gain
    d["gain"] = level_gain(n1, d["att_n"], d["sigma"] * (16384.0 / 49152.0 if d["dword"] else 1.0))
    return d, int(rng.choice([1, 1, 2, 3, 4]))


import os  # noqa: E402


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("LRH_RANDOM_SEEDS", "64")))) + list(range(200, 200 + int(os.environ.get("LRH_RANDOM_SEEDS_EXT", "24")))))
def test_random_configuration_matches_the_oracle(seed):
    d, batch = random_case(seed)
    iq, lim = make_input(d), make_liminfo(d)
    g = {"iq": iq, "liminfo": lim}
    if d["foldcorr_seed"]:
        from refcases import make_foldcorr
        g["foldcorr"] = make_foldcorr(d)
    if d["afc"]:                                                # what the harness's AFC_SUPPLY hands over: a frequency per transform, slow wander plus a step
        r3 = np.random.default_rng(5600 + seed)
        tt = np.arange(64 * d["nblk"] + 64)
        amp, per, at, step = r3.uniform(0.3, 2.0), r3.uniform(15, 60), int(r3.integers(10, 40)), r3.uniform(-3, 3)
        f = (d["fq"] + amp * np.sin(2 * np.pi * tt / per) + step * ((tt >= at) & (tt < at + 30))).astype(np.float32)
        g["afc_fq0"], g["afc_supplied"] = f[:1], f[1:]
    a = run_case(_open_hip, "random", golden=g, batch=batch, params=d)
    b = run_case(_open_oracle, "random", golden=g, batch=batch, params=d)
    truth = {}

    def t(key):
        if not truth:
            truth.update(run_case(_open_truth, "random", golden=g, batch=batch, params=d))
        return truth[key]
    info = {k: d[k] for k in ("n1", "n2", "second_fft", "sinpow1", "sinpow2", "mixred", "nblk", "dword", "direction")}
    cols = [0, 1, 2, 3, 6, 7, 8, 9, 10]
    assert np.array_equal(a["itrace"][:, cols], b["itrace"][:, cols]), info
    # the blanker's noise floor: the same, or -- where a start-up transient has the limit deep inside the noise and nearly every sample is cleared, so that ONE
    # sample within float32 rounding of the limit moves the mean of the few survivors (seed 11: 1 flip of 117 509 decisions, floor 6314 against 6303) --
    # within half a per cent; everything behind the blanker is then compared by the other seeds and the goldens, not here
    same_floor = np.array_equal(a["itrace"][:, 4], b["itrace"][:, 4])
    fa, fb = a["itrace"][:, 4].astype(float), b["itrace"][:, 4].astype(float)
    assert np.all(np.abs(fa - fb) <= np.maximum(1.0, 5e-3 * fb)), info
    if not same_floor and d["second_fft"]:
        assert np.count_nonzero((a["timf2_pwr_float"] == 0) != (b["timf2_pwr_float"] == 0)) <= 8, info
    assert np.array_equal(a["mixtrace"][:, [0, 5, 6, 7]], b["mixtrace"][:, [0, 5, 6, 7]]), info          # mix1_point, old_point, timf3_pa, nx
    rep = {}
    half = a["api"].fft1_interleave_points == a["api"].N1 // 2
    # a sample within float32 rounding of the blanker's limit (seed 462 of the third sweep: power 999.994 against a limit of 1000, kept by the oracle, cleared
    # by HIP): at most two; they are left out of the timf2 rings, and what the second fft makes of them is not compared
    flip_at = np.nonzero((a["timf2_pwr_float"] == 0) != (b["timf2_pwr_float"] == 0))[0] if d["second_fft"] else np.zeros(0, int)
    assert not same_floor or flip_at.size <= 2, (info, flip_at)
    # ... and one that the timf2 ring no longer holds still sits in the transforms and lines it went into: the blanker's own count of cleared points after every call tells
    dc = np.abs(a["cleared_trace"] - b["cleared_trace"])
    flipped_ever = bool(dc.any())
    assert not dc.size or dc.max() <= 3, (info, int(dc.max()))
    for _, key in RINGS:
        x, y = a[key], b[key]
        if not d["second_fft"] and key.startswith(("timf2", "fft2")):
            continue
        if (flip_at.size or flipped_ever) and key.startswith(("fft2", "timf3")):
            continue
        keep = np.ones(x.size, bool)
        if flip_at.size and key == "timf2_pwr_float":
            keep[flip_at] = False
        if flip_at.size and key == "timf2_float":
            for f_ in flip_at:
                keep[4 * f_:4 * f_ + 4] = False
        if key == "timf2_float" and half:                        # the raw half block the reference parks beyond timf2_pa (timf2.c:1018-1025)
            keep[(a["api"].p.timf2_pa + np.arange(4 * (a["api"].N1 // 2))) % x.size] = False
        if key.startswith(("timf2", "fft2", "timf3")) and not same_floor:
            continue                                             # a noise floor apart moves the limit: decisions differ from there on
        # (above 1e-5 both float32 sides are measured against the float64 build; 1.25: in these unplanned level plans the two float32 results sit 2-3e-5 from
        # the truth and 15 % apart from each other -- seeds 18 and 65 of the extended sweep; the goldens and the full-size tests hold 1.0 / 1.05)
        # (1.5 for a ring of a few hundred values, whose error ratio scatters accordingly: seed 650 of the third sweep, timf3 of 256 floats, 1.30)
        truth_gate(rep, key, x[keep], y[keep], (lambda k=key, m=keep: t(k)[m]), tol=1e-5, factor=1.25 if x.size >= 4096 else 1.5)
    flips = 0
    if same_floor and d["second_fft"]:
        flips = np.count_nonzero((a["timf2_pwr_float"] == 0) != (b["timf2_pwr_float"] == 0))
        assert flips <= 2, (info, flips)                         # a sample within float32 rounding of the limit
    if a["wf_lines"].size and same_floor and flips == 0 and not flipped_ever:
        # the integer gate of the goldens: each side against the float64 build's values before truncation.  Not with a flipped blanker decision (one cleared
        # impulse more or less is a flat 0.5 % in the weak bins of the lines it reaches: seeds 40 and 50 of the sweep) and not deeper than 80 dB below the
        # line's strongest bin, where the float32 transform noise is tens of counts on BOTH sides (seed 53: -117 dB, |hip - truth| and |oracle - truth| alike)
        t("wf_lines")
        pre = np.array(truth["api"].wf_pre_lines, np.float64).reshape(-1, a["cfg"].wf_xpixels)
        r_ = b["wf_lines"].astype(np.int64)
        # (80 dB below the strongest pixel of the RUN: the first line of a start-up with nothing cleared -- stupid = 0, seeds 270 and 333 of the third sweep -- is
        # 74 dB below the later ones altogether, two float32 noise floors at -100 and -115 dB over a float64 value of -210 dB)
        sel = (r_.max() - r_) < 8000
        # line by line: a decision that flipped on a sample the ring has since overwritten (seed 50: one sample 5e-6 below the limit, cleared by the oracle,
        # kept by HIP and by nobody's fault) still sits in the one or two lines its transforms went into -- at most two such lines, a per cent at most
        odd = []
        for ln in range(r_.shape[0]):
            if not sel[ln].any():
                continue
            try:
                waterfall_gate({}, a["wf_lines"][ln][sel[ln]], b["wf_lines"][ln][sel[ln]], pre[ln][sel[ln]])
            except AssertionError:
                odd.append(ln)
        dev = {ln: int(np.abs(a["wf_lines"][ln].astype(int) - b["wf_lines"][ln].astype(int))[sel[ln]].max()) if sel[ln].any() else 0 for ln in odd}
        # a line that misses the gate by counts (seed 200 of the third sweep: HIP 4 above the float64 value, the oracle 3 below, 0.07 dB between them) is not
        # a flip; the ones beyond 8 counts are, and there may be few of them
        assert sum(1 for v in dev.values() if v > 8) <= max(2, r_.shape[0] // 40), (info, dev)      # (seed 82 of the sweep: 3 of 150 lines)
        assert all(v <= 10 for v in dev.values()), (info, dev)
        rep["wf_lines_with_a_flip"] = len(odd)
    print(info, "batch", batch, {k: float("%.2e" % v) for k, v in rep.items() if isinstance(v, float)})


@pytest.mark.parametrize("name", ["n10_n12", "n9_n11_sin3", "n11_n9_nowin2"])
def test_silent_input_matches_the_oracle(name):
    """all-zero samples (a receiver with its antenna off): every ring zero, every pointer and counter as the oracle's, and the waterfall at the
    reference's floor -- 1000 log10(0) is -inf on the host, INT_MIN after the conversion, -32767 after the clamp (fft2.c:707-815)"""
    d = case_params(name)
    iq = np.zeros_like(make_input(d))
    g = {"iq": iq, "liminfo": make_liminfo(d)}
    a = run_case(_open_hip, name, golden=g, params=d)
    b = run_case(_open_oracle, name, golden=g, params=d)
    assert np.array_equal(a["itrace"], b["itrace"]) and np.array_equal(a["mixtrace"], b["mixtrace"])
    for _, key in RINGS:                                        # (fft1_slowsum carries the reference's 1e-8 floor, everything else is zero)
        assert np.array_equal(a[key], b[key]) and np.all(np.abs(a[key]) <= 1e-7), key
    assert a["wf_lines"].size and np.array_equal(a["wf_lines"], b["wf_lines"]) and np.all(a["wf_lines"] == -32767)


def random_dsp_case(seed):
    rng = np.random.default_rng(8100 + seed)
    fft1_n = int(rng.choice([12, 13, 14]))
    fft2_n = int(rng.integers(fft1_n - 2, 17))
    batch = int(rng.choice([16, 24, 32, 40, 64]))
    fft3 = bool(rng.random() < 0.4) and fft2_n >= 12
    calls = [int(batch * rng.integers(1, 4) + (rng.integers(1, batch) if rng.random() < 0.5 else 0)) for _ in range(int(rng.integers(1, 4)))]   # ragged last rounds too
    return dict(fft1_n=fft1_n, fft2_n=fft2_n, batch=batch, fft3_n=12 if fft3 else 0, mix2_n=8 if fft3 else 0, calls=calls, sparse=int(rng.random() < 0.5),
                dword=int(rng.random() < 0.2), fq=float(rng.uniform(0.2, 0.8)))


@pytest.mark.parametrize("seed", range(int(os.environ.get("LRH_RANDOM_DSP_SEEDS", "10"))))
def test_random_wideband_dsp_matches_the_oracle(seed):
    """lrh_wideband_dsp -- the batched entry the bench measures, with its fused kernels from 32 blocks per round up and its one-round-late schedule -- at
    random sizes, round lengths and call lengths (rounds that are not full, several calls on one context), against the oracle's lro_wideband_dsp:
    pointers and counters exact, rings at 1e-5 or held to the float64 build"""
    from linrad_amd import abi
    from linrad_amd.lib import synth_defaults, synth_iq
    from linrad_amd.workload import chain_config, strong_liminfo
    q = random_dsp_case(seed)
    total = sum(q["calls"])
    cfg = chain_config(q["fft1_n"], q["fft2_n"], batch=q["batch"], fft3_n=q["fft3_n"], mix2_n=q["mix2_n"], rounds=max(1, (max(q["calls"]) + q["batch"] - 1) // q["batch"]))
    N1, M1 = 1 << q["fft1_n"], (1 << q["fft1_n"]) // 2
    while cfg.timf1_bytes < 4 * (total * M1 + 2 * N1) * (2 if q["dword"] else 1):
        cfg.timf1_bytes *= 2
    cfg.timf1_dword_input = q["dword"]
    s = synth_defaults(N1, 0)
    iq16 = synth_iq(s, 0, cfg.timf1_bytes // (8 if q["dword"] else 4))
    iq = ((iq16.astype(np.int32) << 14) | 0x2000) if q["dword"] else iq16
    lim = strong_liminfo(s, q["fft1_n"])
    rings = [(abi.RING_FFT1_SUMSQ, "sumsq"), (abi.RING_FFT1_SLOWSUM, "slowsum"), (abi.RING_TIMF2_PWR, "pwr"), (abi.RING_TIMF2_FLOAT, "timf2"),
             (abi.RING_FFT2_POWERSUM, "ps2"), (abi.RING_TIMF3_FLOAT, "timf3"), (abi.RING_WG_WATERF, "wf")] + ([(abi.RING_BASEB_RAW, "baseb")] if q["fft3_n"] else [])
    res = {}
    for name, fn in (("hip", _open_hip), ("ora", _open_oracle), ("truth", _open_truth)):
        if name == "truth" and res.get("skip_truth"):
            continue
        cfg.fft1_float_sparse = cfg.fft2_float_sparse = q["sparse"] if name == "hip" else 0
        rx = fn(cfg)
        rx.timf1_write(iq)
        rx.set_liminfo(lim)
        rx.set_mix1_selfreq(q["fq"] * (1 << q["fft2_n"]))
        for n in q["calls"]:
            rx.wideband_dsp(n, q["batch"])
        res[name] = {k: rx.export(r) for r, k in rings}
        res[name]["p"] = rx.p.as_dict()
        res[name]["floor"] = rx.blanker_state().timf2_noise_floor
        rx.close()
    cfg.fft1_float_sparse = cfg.fft2_float_sparse = 0
    h, o, T = res["hip"], res["ora"], res["truth"]
    assert h["p"] == o["p"], (q, {k: (h["p"][k], o["p"][k]) for k in h["p"] if h["p"][k] != o["p"][k]})
    assert abs(h["floor"] - o["floor"]) <= max(1, 5e-3 * o["floor"]), q
    rep = {}
    flips = int(np.count_nonzero((h["pwr"] == 0) != (o["pwr"] == 0)))
    assert flips <= 4 + 4e-6 * h["pwr"].size, (q, flips)           # samples within float32 rounding of the limit: 1-2 per million decisions
    npa = o["p"]["timf2_pa"] // 4
    ring = h["timf2"].size // 4
    keep4 = np.ones(ring, bool)
    keep4[(npa + np.arange(M1)) % ring] = False                 # the raw half block the reference parks beyond timf2_pa
    for key in ("sumsq", "slowsum"):
        truth_gate(rep, key, h[key], o[key], lambda k=key: T[k], tol=1e-5, factor=1.25)
    same = flips == 0 and h["floor"] == o["floor"]
    if same:
        truth_gate(rep, "timf2", h["timf2"].reshape(-1, 4)[keep4], o["timf2"].reshape(-1, 4)[keep4], lambda: T["timf2"].reshape(-1, 4)[keep4], tol=1e-5, factor=1.25)
        nz = (h["pwr"] != 0) & (T["pwr"] != 0)
        truth_gate(rep, "pwr", h["pwr"][nz], o["pwr"][nz], lambda: T["pwr"][nz], tol=1e-5, factor=1.25)
        for key in ("ps2", "timf3") + (("baseb",) if q["fft3_n"] else ()):
            truth_gate(rep, key, h[key], o[key], lambda k=key: T[k], tol=1e-5, factor=1.25)
        # (a decision that flipped on a sample the ring has since overwritten still sits in the lines its transforms went into: see the stage-call test)
        dw = np.abs(h["wf"].astype(int) - o["wf"].astype(int))
        odd = int(np.count_nonzero((dw.reshape(-1, cfg.wf_xpixels) > 3).any(axis=1)))
        assert odd <= 3 and dw.max() <= 40, (q, odd, int(dw.max()))            # at most a few of the 64 lines carry such a flip
    print(q, "flips", flips, {k: float("%.2e" % v) for k, v in rep.items() if isinstance(v, float)})


def random_sellim_case(seed):
    """a selective-limiter case (refcases.SELLIM form) with random levels, switches and carriers on one of the small golden bases"""
    rng = np.random.default_rng(6300 + seed)
    base = str(rng.choice(["n10_n12", "n9_n11_shift"]))
    n1 = 1024 if base == "n10_n12" else 512
    maxlevel = int(rng.choice([1500, 4000, 12000]))
    nblk = int(rng.choice([96, 128, 160, 200]))
    on = int(rng.integers(10, nblk // 2))
    t = dict(base=base, nblk=nblk, maxlevel=maxlevel, lim_groups=int(rng.choice([16, 32, 64][:2 if n1 == 512 else 3])), max_fft1n=32, sumsq_blocks=16, stupid=0, blocktime=float(rng.choice([0.0008, 0.002, 0.005])),
             ston_fft1=float(rng.uniform(2.5, 6.0)), bw_fftxpts=int(rng.integers(8, 64)), seed=int(7000 + seed), sample_shift=0, blockpower_block=0,
             keyed=(float(rng.uniform(-0.45, 0.45) * n1), float(rng.uniform(0.3, 1.6) * maxlevel), on, int(rng.integers(on + 5, nblk))),
             par1=int(rng.integers(0, 3)), par2=int(rng.integers(0, 2)), par3=int(rng.integers(0, 2)), par4=int(rng.integers(0, 2)), par5=int(rng.integers(0, 3)),
             par6=int(rng.integers(0, 2)), par7=int(rng.integers(0, 2)), par8=int(rng.integers(0, 2)), sellim2=int(rng.random() < 0.6),
             ston_fft2=float(rng.uniform(8.0, 45.0)), wf_avgnum=int(rng.integers(1, 4)),
             strong=[(float(rng.uniform(-0.45, 0.45) * n1), float(rng.uniform(0.2, 1.5) * maxlevel)) for _ in range(int(rng.integers(0, 5)))],
             weak=[(float(rng.uniform(-0.45, 0.45) * n1), float(rng.uniform(30.0, 600.0))) for _ in range(int(rng.integers(0, 4)))])
    # (buf.c:816-820: 16 .. fft1_size / 16 groups.  stupid = 0: the blanker clears nothing -- a sample within rounding of its limit, one or two per million,
    # would otherwise end in timf3 as a difference of per cent, seeds 126 and 252; the chain tests above hold the blanker)
    return t, dict(batch=int(rng.choice([1, 2, 4, 8])), in_call=bool(rng.random() < 0.5), fq=float(rng.uniform(0.1, 0.9)))


@pytest.mark.parametrize("seed", range(int(os.environ.get("LRH_RANDOM_SELLIM_SEEDS", "32"))))
def test_random_selective_limiter_matches_the_oracle(seed):
    """both limiters (sellim.c:159-1130) at random levels, hg.sellim_par1..8, group sizes, carriers coming and going: the routing table after every
    round of blocks -- which bins are weak, strong, attenuated -- exact, the attenuations to 1e-5, the amplitude factor, the weak-bin counts and
    the rings the routing shapes.  The limiter calls inside lrh_wideband_dsp or made by the caller between rounds (wcw.c:1124-1133)"""
    import refcases
    from linrad_amd import abi
    from linrad_amd.abi import default_sellim
    from refcases import interleave, lrh_config, sellim_case
    t, how = random_sellim_case(seed)
    name = f"random_sellim_{seed}"
    refcases.SELLIM[name] = t
    try:
        d, sl, iq = sellim_case(name)
    finally:
        del refcases.SELLIM[name]
    batch = how["batch"]
    cfg = lrh_config(d, iq, max_batch=max(4, batch))
    N1, N2 = 1 << d["n1"], 1 << d["n2"]
    bt2 = float(np.float32(sl["blocktime"]) * np.float32(N2 - interleave(d["n2"], d["sinpow2"])) / np.float32(N1 - interleave(d["n1"], d["sinpow1"])))
    rings = [(abi.RING_TIMF2_FLOAT, "timf2"), (abi.RING_FFT1_SLOWSUM, "slowsum"), (abi.RING_TIMF3_FLOAT, "timf3")]
    nr = d["nblk"] // batch
    # the limiters come on after 24 blocks: in the start-up transient the second limiter's floors hang on bins 80 dB below the carriers (the lowest bin
    # of a region sets the limit of its floor, sellim.c:340-371), where two float32 transforms differ by per cent -- seed 133 flipped a bin that way, 0.6 %
    # from its limit; the goldens hold the start-up on cases that are away from such edges (tests/golden/make_golden_sellim.py)
    # End points inside the filter's skirts (4 bins either side; a calibrated Linrad sets them from the filter, fft1.c:4600-4618): the empty bins beyond hold
    # nothing but the transforms' rounding noise in the second fft, and the lowest bin of a region sets the region's floor (seed 191: 15-100 % apart there).
    erng = np.random.default_rng(6400 + seed)
    ends = [4 + int(erng.integers(0, 5)), 0, 0, N1 - 5 - int(erng.integers(0, 5))]
    ends[1], ends[2] = ends[0] + int(erng.integers(0, 9)), ends[3] - int(erng.integers(0, 9))     # first_point <= first_inband < last_inband <= last_point
    warm = 24 // batch
    res = []
    for fn in (_open_hip, _open_oracle):
        rx = fn(cfg)
        rx.timf1_write(iq)
        rx.set_mix1_selfreq(how["fq"] * N2)
        par = default_sellim(cfg, sellim_maxlevel=sl["maxlevel"], liminfo_group_points=max(1, N1 // sl["lim_groups"]), fft1_blocktime=sl["blocktime"],
                             blanker_ston_fft1=sl["ston_fft1"], baseband_bw_fftxpts=sl["bw_fftxpts"], blanker_ston_fft2=sl["ston_fft2"], fft2_blocktime=bt2,
                             exact_stats=1, fft1_first_point=ends[0], fft1_first_inband=ends[1], fft1_last_inband=ends[2], fft1_last_point=ends[3],
                             **{f"sellim_par{i}": sl[f"par{i}"] for i in range(1, 9)})
        trace, amps, lows = [], [], []
        c1 = c2 = 0
        for r in range(nr):
            if r == warm and how["in_call"]:
                rx.wideband_limiter(par, bool(sl["sellim2"]))
            rx.wideband_dsp(batch, batch)
            if r >= warm and not how["in_call"]:
                if rx.p.fft1_liminfo_cnt != c1:
                    rx.fft1_update_liminfo(par)
                    c1 = rx.p.fft1_liminfo_cnt
                if sl["sellim2"] and rx.p.fft2_liminfo_cnt != c2:
                    rx.fft2_update_liminfo(par)
                    c2 = rx.p.fft2_liminfo_cnt
            trace.append(rx.get_liminfo())
            amps.append(rx.liminfo_amplitude_factor())
            lows.append(rx.p.fft1_lowlevel_points)
        res.append(dict(trace=np.array(trace), amp=np.array(amps, np.float32), low=np.array(lows), p=rx.p.as_dict(), **{k: rx.export(r) for r, k in rings}))
        rx.close()

    def truth():
        """the float64 build routed by the oracle's tables round by round (its own limiter could decide otherwise on a bin at the threshold)"""
        if "t" not in truth.__dict__:
            rx = _open_truth(cfg)
            rx.timf1_write(iq)
            rx.set_mix1_selfreq(how["fq"] * N2)
            for r in range(nr):
                if r > warm:
                    rx.set_liminfo(res[1]["trace"][r - 1])
                    rx.set_liminfo_amplitude_factor(float(res[1]["amp"][r - 1]))
                rx.wideband_dsp(batch, batch)
            truth.t = {k: rx.export(r) for r, k in rings}
            rx.close()
        return truth.t
    h, o = res
    ctx = dict(seed=seed, case={k: v for k, v in t.items() if k not in ("strong", "weak")}, how=how, ends=ends)
    bad = np.nonzero((np.sign(h["trace"]) != np.sign(o["trace"])).any(axis=1))[0]
    if bad.size:
        # A bin at its threshold: the second limiter compares fft2 bins 60-70 dB below the carriers, where two float32 transforms are 0.5 % apart, with a
        # floor averaged over hundreds of such bins (scripts/sellim_diag.py: seed 232 sits 6.7e-6 from its limit in the oracle, seed 140 swaps two bins whose
        # powers differ by 0.5 %) -- about one run in a hundred.  One or two bins at the first round that differs: everything before it is held exactly,
        # the rest of the run (the hold-offs carry the decision on) is not compared.
        r0 = int(bad[0])
        bins = np.nonzero(np.sign(h["trace"][r0]) != np.sign(o["trace"][r0]))[0]
        assert bins.size <= 2 and r0 > warm, (ctx, "first round with another routing pattern", r0, bins[:8])
        assert np.array_equal(h["low"][:r0], o["low"][:r0]), ctx
        pos = o["trace"][:r0] > 0
        verr = float(np.max(np.abs(h["trace"][:r0][pos] - o["trace"][:r0][pos]) / o["trace"][:r0][pos])) if pos.any() else 0.0
        assert verr <= 1e-5 and float(np.max(np.abs(h["amp"][:r0] - o["amp"][:r0]))) <= 1e-6, (ctx, verr)
        print(ctx, "threshold-edge decision at round", r0, "bins", bins, "-- compared up to there")
        return
    assert np.array_equal(h["low"], o["low"]), ctx
    ints = [k for k, v in h["p"].items() if isinstance(v, int)]
    assert {k: h["p"][k] for k in ints} == {k: o["p"][k] for k in ints}, ctx
    pos = o["trace"] > 0
    verr = float(np.max(np.abs(h["trace"][pos] - o["trace"][pos]) / o["trace"][pos])) if pos.any() else 0.0
    aerr = float(np.max(np.abs(h["amp"] - o["amp"])))
    assert verr <= 1e-5 and aerr <= 1e-6, (ctx, verr, aerr)
    keep = np.ones(h["timf2"].size, bool)
    keep[(h["p"]["timf2_pa"] + np.arange(4 * (N1 // 2))) % keep.size] = False
    rep = {}
    truth_gate(rep, "timf2", h["timf2"] * keep, o["timf2"] * keep, lambda: truth()["timf2"] * keep, tol=1e-5, factor=1.25)
    truth_gate(rep, "timf3", h["timf3"], o["timf3"], lambda: truth()["timf3"], tol=1e-5, factor=1.25)
    truth_gate(rep, "slowsum", h["slowsum"], o["slowsum"], lambda: truth()["slowsum"], tol=1e-5, factor=1.25)
    print(ctx, "strong", int(np.count_nonzero(o["trace"][-1])), "attenuated", int(pos[-1].sum()), "table", verr, "amp", aerr, rep)


def random_clever_case(seed):
    """another pulse train (count, amplitudes, unresolved pairs, rectangular hits, noise) on the tables one of the linear-blanker goldens carries"""
    import refcases
    rng = np.random.default_rng(9100 + seed)
    base = str(rng.choice(["clever_n10_n12", "clever_n9_n11_only"]))
    t = dict(refcases.CLEVER[base])
    lo = float(rng.uniform(800.0, 4000.0))
    t.update(pulses=int(rng.integers(5, 120)), pulse_seed=int(300 + seed), amp=(lo, lo * float(rng.uniform(2.0, 12.0))), pairs=int(rng.integers(0, 10)),
             rects=int(rng.integers(0, 6)), seed=int(9200 + seed), nblk=int(rng.choice([64, 96, 128])))
    t["pairs"] = min(t["pairs"], t["pulses"])
    return base, t


@pytest.mark.parametrize("seed", range(int(os.environ.get("LRH_RANDOM_CLEVER_SEEDS", "16"))))
def test_random_linear_blanker_matches_the_oracle(seed):
    """the linear ("clever") blanker (blank1.c:36-1087: pulse search, fit against the reference pulse at the fitted fractional delay, subtraction, the
    rejected ones left to the stupid blanker) on random pulse trains: its scalars after every call -- pointers, cleared and fitted counts, noise floor,
    both limits -- exact, the rings behind at 1e-5"""
    import cleverlib
    import refcases
    base, t = random_clever_case(seed)
    g = cleverlib.load(base)
    name = f"random_clever_{seed}"
    refcases.CLEVER[name] = t
    try:
        case = refcases.clever_case(name)
    finally:
        del refcases.CLEVER[name]
    h = cleverlib.run(_open_hip, base, g, case=case)
    o = cleverlib.run(_open_oracle, base, g, case=case)
    names = ["timf2_pa", "timf2p_fit", "timf2_pn2", "cleared_points", "blanker_points", "noise_floor", "stupid_limit", "clever_limit", "fitted_pulses", "last_fitted", "last_rejected"]
    ctx = dict(seed=seed, base=base, case={k: v for k, v in t.items() if k in ("pulses", "amp", "pairs", "rects", "nblk")})
    n1 = h["api"].N1
    keep = np.ones(h["timf2"].size, bool)
    keep[(h["api"].p.timf2_pa + np.arange(4 * (n1 // 2))) % keep.size] = False
    flips = np.nonzero(((h["pwr"] == 0) != (o["pwr"] == 0)) & keep[::4])[0]
    # a sample within float32 rounding of the stupid blanker's limit (1-2 per million decisions) shows as a cleared-points count one or two apart --
    # also when the ring has moved on since (seed 100 of the long sweep) --, and the integer noise floor it is part of, or that two float32 means round to
    # either side of, a few counts apart (seeds 137, 143, 213, 278: 1-3 of 740 .. 48000): what follows such a run is not compared
    edge = False
    for j, nm in enumerate(names):
        bad = np.nonzero(h["rows"][:, j] != o["rows"][:, j])[0]
        dj = np.abs(h["rows"][:, j] - o["rows"][:, j])
        if bad.size and nm in ("cleared_points", "blanker_points") and dj.max() <= 2:
            edge = True
            continue
        if bad.size and nm in ("noise_floor", "stupid_limit", "clever_limit") and np.max(dj / np.maximum(1, o["rows"][:, j])) <= 5e-3:
            edge = True
            continue
        assert bad.size == 0, (ctx, nm, "first differs at call", int(bad[0]), int(h["rows"][bad[0], j]), int(o["rows"][bad[0], j]))
    if edge and not flips.size:
        print(ctx, "a decision or a floor at its rounding edge (traces above): rings not compared")
        h["api"].close(), o["api"].close()
        return
    if flips.size:                                              # the rings behind differ by that sample: its neighbourhood is left out, timf3 is not compared
        assert flips.size <= 2, (ctx, flips)
        for f in flips:
            keep[4 * f:4 * f + 4] = False
        e2 = relerr(h["timf2"] * keep, o["timf2"] * keep)
        print(ctx, "knife-edge clearing decision at", flips, "timf2 elsewhere", e2)
        assert e2 <= 1e-5, (ctx, e2)
        h["api"].close(), o["api"].close()
        return
    e2, ep, e3 = relerr(h["timf2"] * keep, o["timf2"] * keep), relerr(h["pwr"] * keep[::4], o["pwr"] * keep[::4]), relerr(h["timf3"], o["timf3"])
    print(ctx, "fitted", int(o["rows"][:, 9].sum()), "rejected", int(o["rows"][:, 10].sum()), "cleared", int(o["rows"][-1, 3]), "timf2", e2, "pwr", ep, "timf3", e3)
    assert np.array_equal((h["pwr"] == 0) & keep[::4], (o["pwr"] == 0) & keep[::4]), ctx
    assert e2 <= 1e-5 and ep <= 1e-5 and e3 <= 1e-5, (ctx, e2, ep, e3)
    h["api"].close(), o["api"].close()


def random_clever2_case(seed):
    """two coupled channels with the linear blanker on: a random pulse train (random_clever_case) that reaches channel 1 under a random sky phase and gain.
    Returns the names and entries to put into refcases.CLEVER / CLEVER2 for the run, and the golden whose tables (init_blanker's) go with the base case."""
    base, t = random_clever_case(1000 + seed)
    rng = np.random.default_rng(3700 + seed)
    t = dict(t, nblk=int(rng.choice([64, 96])))
    n1, n2 = f"random_clever2_base_{seed}", f"random_clever2_{seed}"
    t2 = dict(base=n1, nblk=t["nblk"], seed2=int(3800 + seed), sky_phase=float(rng.uniform(-3.1, 3.1)), gain=float(rng.uniform(0.5, 1.5)))
    return n1, t, n2, t2, {"clever_n10_n12": "clever2_n10", "clever_n9_n11_only": "clever2_n9_only"}[base]


@pytest.mark.parametrize("seed", range(int(os.environ.get("LRH_RANDOM_CLEVER2_SEEDS", "6"))))
def test_random_two_channel_linear_blanker_matches_the_oracle(seed):
    """the linear blanker on two coupled contexts (get_pulse_pol, transform_timf2_pol, subtract_twochan_pulse, blank1.c:232-609; the exchanges of
    tests/clever2lib.py made by hand) on random pulse trains, sky phase and channel gain, with the tables of the golden the pulse train's base case has:
    the scalars of both contexts after every call exact, their timf2 rings at 1e-5"""
    import clever2lib
    import refcases
    n1_, t1, n2_, t2, gname = random_clever2_case(seed)
    g = dict(clever2lib.load(gname))
    refcases.CLEVER[n1_], refcases.CLEVER2[n2_] = t1, t2
    try:
        g["frames"] = refcases.clever2_case(n2_)[2]
        h = clever2lib.run(_open_hip, n2_, g, frames_mode=True)
        o = clever2lib.run(_open_oracle, n2_, g, frames_mode=False)
    finally:
        del refcases.CLEVER[n1_], refcases.CLEVER2[n2_]
    ctx = dict(seed=seed, tables=gname, case={k: t2[k] for k in ("sky_phase", "gain", "nblk")}, pulses=t1["pulses"])
    rh, ro = h["rows"].astype(np.float64), o["rows"].astype(np.float64)
    assert rh.shape == ro.shape, (ctx, rh.shape, ro.shape)
    d_ = np.abs(rh - ro)
    edge = bool(d_.any())                                       # (a decision or an integer floor at its rounding edge: see the one-channel test above)
    lim = np.maximum(3, 5e-3 * np.abs(ro))
    lim[:, :, 3:5] = np.maximum(lim[:, :, 3:5], 8)              # (cleared / blanker points: a pulse goes with the neighbours it takes along -- seed 8, four samples)
    assert np.all(d_ <= lim), (ctx, "per-call scalars differ", np.argwhere(d_ > lim)[:4], rh[d_ > lim][:4], ro[d_ > lim][:4])
    if not edge:
        n1 = h["rxs"][0].N1
        keep = np.ones(h["out"][0]["pwr"].size, bool)            # per sample; the raw half block parked beyond timf2_pa is not the device's to store (paritylib)
        keep[(o["out"][0]["p"]["timf2_pa"] // 4 + np.arange(n1 // 2)) % keep.size] = False
        for ch in (0, 1):
            e = relerr(h["out"][ch]["timf2"].reshape(-1, 4)[keep], o["out"][ch]["timf2"].reshape(-1, 4)[keep])
            ep = relerr(h["out"][ch]["pwr"][keep], o["out"][ch]["pwr"][keep])
            assert e <= 1e-5 and ep <= 2e-5, (ctx, ch, e, ep)
            assert np.array_equal((h["out"][ch]["pwr"] == 0) & keep, (o["out"][ch]["pwr"] == 0) & keep), (ctx, ch)
    print(ctx, "fitted", int(ro[:, 0, 9].sum()), "rejected", int(ro[:, 0, 10].sum()), "rounding-edge decision: rings not compared" if edge else "scalars exact")
    for rx in h["rxs"] + o["rxs"]:
        rx.close()


def random_spur_case(seed):
    """a carrier of random frequency, drift and level on the n10_n12 base, acquired by the API's own store_new_spur / spur_phase_lock"""
    rng = np.random.default_rng(4400 + seed)
    speknum = int(rng.choice([8, 12, 16]))
    bin0 = float(rng.uniform(600.0, 3500.0))
    t = dict(base="n10_n12", nblk=int(rng.choice([240, 320, 400])), max_fft2n=64, blockpower_block=0, spur_pnt=int(bin0) - 3, spur_start=int(rng.integers(speknum + 4, 30)),
             spur_speknum=speknum, tone=(bin0, float(rng.uniform(-0.4, 0.4)), float(rng.uniform(400.0, 6000.0))), fq=bin0 + float(rng.uniform(-12.0, 12.0)), seed=int(4500 + seed),
             stupid=0)          # (the blanker clears nothing: a sample at its limit -- a flat error over a whole transform, seeds 81 .. 292 of the long sweep -- is the chain tests' business)
    return t, int(rng.choice([1, 2, 4]))


@pytest.mark.parametrize("seed", range(int(os.environ.get("LRH_RANDOM_SPUR_SEEDS", "12"))))
def test_random_spur_is_acquired_and_tracked_like_the_oracle(seed):
    """spur removal (spursub.c:181-343 store_new_spur / spur_phase_lock / initial_remove_spur, spur.c:36-494 eliminate_spurs) on a carrier of random
    frequency, drift and level: the same lock decision, the same window and flag after every transform, frequency / phase / amplitude of the loop,
    the fft2 ring behind the subtraction, its power sums and timf3"""
    import refcases
    import spurlib
    t, batch = random_spur_case(seed)
    name = f"random_spur_{seed}"
    refcases.SPUR[name] = t
    try:
        case = refcases.spur_case(name)
    finally:
        del refcases.SPUR[name]
    gg = spurlib.load("spur_n10_n12")
    st = np.zeros(16); st[10] = t["spur_speknum"]
    g = {"iq": case[2], "spur_init_state": st, "spur_locked": np.array([t["spur_start"]]), "spur_spectra": gg["spur_spectra"]}
    res = []
    for fn in (_open_hip, _open_oracle):
        try:
            res.append(spurlib.run(fn, name, g, batch=batch, acquire=True, case=case))
        except AssertionError as e:
            assert str(e) == "no lock"
            res.append(None)
    ctx = dict(seed=seed, tone=t["tone"], speknum=t["spur_speknum"], start=t["spur_start"], batch=batch)
    assert (res[0] is None) == (res[1] is None), (ctx, "lock decisions differ", res[0] is None, res[1] is None)
    if res[0] is None:
        print(ctx, "no lock on either side")
        return
    h, o = res
    assert h["trace"].shape == o["trace"].shape and h["trace"].shape[0] > 10, (ctx, h["trace"].shape, o["trace"].shape)
    assert np.array_equal(h["trace"][:, :2], o["trace"][:, :2]), (ctx, "spur_location / spur_flag trace differs")

    def wrap(x):
        return (x + np.pi) % (2 * np.pi) - np.pi
    locked = o["trace"][:, 1] == 0
    ferr = float(np.max(np.abs(h["trace"][:, 2] - o["trace"][:, 2])))
    perr = float(np.max(np.abs(wrap(h["trace"][locked, 3] - o["trace"][locked, 3])))) if locked.any() else 0.0
    aerr = float(np.max(np.abs(h["trace"][locked, 6] - o["trace"][locked, 6]) / np.abs(o["trace"][locked, 6]))) if locked.any() else 0.0
    e2, e3, ep = relerr(h["fft2"], o["fft2"]), relerr(h["timf3"], o["timf3"]), relerr(h["ps2"], o["ps2"])
    # timf3 is what the subtraction leaves of a passband the carrier fills: two float32 results of that cancellation are held to 1e-5 of what was
    # there BEFORE it (the same chain with no spur taken on), as spurlib.compare holds the residual to the carrier
    d, sp, iq, lim = case
    raw = _open_oracle(refcases.lrh_config(d, iq))
    raw.timf1_write(iq), raw.set_liminfo(lim), raw.set_mix1_selfreq(d["fq"])
    for b in range(d["nblk"]):
        raw.fft1_b(1), raw.fft1_c(1), raw.make_timf2(1)
        raw.first_noise_blanker()
        for _ in range(raw.fft2_available()):
            raw.make_fft2(1)
            raw.fft2_mix1_fixed(1)
    from linrad_amd import abi
    t3 = raw.export(abi.RING_TIMF3_FLOAT).astype(np.float64)
    raw.close()
    e3c = float(np.linalg.norm(h["timf3"].astype(np.float64) - o["timf3"]) / np.linalg.norm(t3))
    print(ctx, "transforms", int(h["trace"].shape[0]), "locked", int(locked.sum()), "freq", ferr, "phase", perr, "ampl", aerr, "fft2", e2, "timf3", e3, "vs unsubtracted", e3c, "ps2", ep,
          "carrier / residual", float(np.linalg.norm(t3) / np.linalg.norm(o["timf3"])))
    # The loop is an iterated estimator with convergence tests (spur.c:181-420): where HIP's wave-parallel sums and the oracle's serial ones take another number of
    # passes the two states end a milliradian apart (8 of 300 carriers in the long sweep: phase up to 4.6e-3 rad, amplitude 5e-5) -- both inside the loop's own
    # noise, the reference leaves 0.3 .. 40 % of the carrier itself.  So: the state to the goldens' tolerance (spurlib.compare: 0.02 rad, 1e-3 bins), what HIP
    # leaves of the carrier no more than what the oracle leaves, and everything away from the window at 1e-5.
    assert ferr <= 1e-3 and perr <= 2e-2 and aerr <= 2e-3, (ctx, ferr, perr, aerr)
    cfg_, loc = h["cfg"], int(o["trace"][-1, 0])
    n2_ = 1 << cfg_.fft2_n
    fh, fo = h["fft2"].reshape(cfg_.max_fft2n, n2_, 2).astype(np.float64), o["fft2"].reshape(cfg_.max_fft2n, n2_, 2).astype(np.float64)
    win = np.zeros(n2_, bool)
    win[max(0, loc - 2):loc + 10] = True
    res_h, res_o = float(np.linalg.norm(fh[:, win])), float(np.linalg.norm(fo[:, win]))
    e2_out = float(np.linalg.norm((fh - fo)[:, ~win]) / np.linalg.norm(fo[:, ~win]))
    assert res_h <= 1.1 * res_o + 1e-5 * np.linalg.norm(fo), (ctx, res_h, res_o)
    assert e2_out <= 1e-5 and ep <= 1e-4, (ctx, e2_out, ep)
    assert (e2 <= 1e-5 and e3c <= 1e-5 and ep <= 1e-5) or perr > 1e-5 or aerr > 1e-5, (ctx, e2, e3, e3c, ep, perr, aerr)     # identical states: identical rings
    h["api"].close(), o["api"].close()


def random_twochan_case(seed):
    """two RF channels (refcases.TWOCHAN form): the second one the first one's signals under a random sky phase with noise of its own, random
    phasing of channel 2, polarisation of the baseband pair and run length"""
    rng = np.random.default_rng(2200 + seed)
    gname = str(rng.choice(["twochan_n10", "twochan_n9_sin3"]))
    import refcases
    t = dict(refcases.TWOCHAN[gname])
    ang = float(rng.uniform(-3.1, 3.1))
    v = rng.normal(0, 1, 3); v /= np.linalg.norm(v)
    t.update(seed2=int(2300 + seed), sky_phase=float(rng.uniform(-3.1, 3.1)), ch2_c1=float(np.float32(np.cos(ang))), ch2_c2=float(np.float32(np.sin(ang))))
    t["chain"] = dict(t["chain"], nblk=int(rng.choice([48, 64, 80])), pol=tuple(float(np.float32(x)) for x in v))
    return gname, t, int(rng.choice([1, 2, 3]))


@pytest.mark.parametrize("seed", range(int(os.environ.get("LRH_RANDOM_TWOCHAN_SEEDS", "6"))))
def test_random_two_channel_chain_matches_the_oracle(seed):
    """two coupled contexts (one per RF channel) through the whole chain with the exchanges made by hand -- summed powers for the blanker, its
    statistics, the other channel's fft2 bins for the cross products, the polarisation sums of fft3_mix2 -- on random sky phase, channel-2 phasing,
    polarisation and run length: pointers and blanker state of both contexts exact, every ring at 1e-5"""
    import refcases
    import test_twochan as TC
    gname, t, batch = random_twochan_case(seed)
    name = f"random_twochan_{seed}"
    refcases.TWOCHAN[name] = t
    try:
        _, _, h, wf_h, n_h = TC._run_chain(_open_hip, name, True, batch, golden_name=gname)
        d, _, o, wf_o, n_o = TC._run_chain(_open_oracle, name, False, batch, golden_name=gname)
    finally:
        del refcases.TWOCHAN[name]
    ctx = dict(seed=seed, base=gname, sky_phase=t["sky_phase"], pol=t["chain"]["pol"], nblk=t["chain"]["nblk"], batch=batch)
    assert n_h == n_o and len(wf_h) == len(wf_o) and n_h >= 4, ctx
    rep = {}
    for ch in (0, 1):
        ph, po = ({k: v for k, v in q[ch]["p"].items() if k != "timf1p_px"} for q in (h, o))        # (the HIP contexts read their channel out of the interleaved frames: a ring of twice the bytes)
        assert ph == po, (ctx, ch, {k: (ph[k], po[k]) for k in ph if ph[k] != po[k]})
        flips = int(np.count_nonzero((h[ch]["pwr"] == 0) != (o[ch]["pwr"] == 0)))
        bh, bo = h[ch]["bs"], o[ch]["bs"]
        same = flips == 0 and bh.timf2_noise_floor == bo.timf2_noise_floor and bh.timf2_cleared_points == bo.timf2_cleared_points
        rep[f"flips{ch}"] = flips
        # (the base cases' pulses are all alike: when the count of neighbours a pulse takes with it, int(clr * sqrt(peak / noise) / 100 + 0.5), sits at a step, every
        # pulse goes the same way -- seed 172 of the long sweep, 8 samples in 7 runs, the floors equal)
        fl = np.nonzero((h[ch]["pwr"] == 0) != (o[ch]["pwr"] == 0))[0]
        runs = 0 if not fl.size else 1 + int(np.count_nonzero(np.diff(fl) > 16))
        assert runs <= 10 and flips <= 24 and abs(bh.timf2_noise_floor - bo.timf2_noise_floor) <= max(1, 5e-3 * bo.timf2_noise_floor), (ctx, ch, flips, runs, bh.timf2_noise_floor, bo.timf2_noise_floor)
        if not same:                                            # a sample within rounding of the blanker's limit: what follows it is not compared (see the tests above)
            continue
        for key in ("fft2", "xyp", "xys", "timf3", "fft3", "baseb"):
            a, b = h[ch][key].astype(np.float64), o[ch][key].astype(np.float64)
            e = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
            rep[f"{key}{ch}"] = float("%.2e" % e)
            # (timf3 and what follows are a weak band cut from the wide spectrum: held to the float32 floor of that spectrum like in test_twochan._check_chain)
            wide = np.linalg.norm(o[ch]["fft2"].astype(np.float64)) / np.sqrt(o[ch]["fft2"].size / (2 << d["n2"]))
            floor = 4 * 6e-8 * wide * np.sqrt(a.size) if key in ("timf3", "fft3", "baseb") else 0.0
            ok = e <= 1e-5 * (2 if key in ("xyp", "xys") else 1) or np.linalg.norm(a - b) <= floor
            if not ok and key in ("timf3", "fft3", "baseb"):
                # a decision that went the other way on a sample the timf2 and fft2 rings no longer hold (they passed above) still sits in one or two blocks of
                # these longer rings (seed 68 of the long sweep): the error then is all in a few blocks
                blk = ((a - b) ** 2).reshape(16, -1).sum(axis=1) if a.size % 16 == 0 else np.array([((a - b) ** 2).sum()])
                ok = np.sort(blk)[-3:].sum() >= 0.99 * blk.sum()
                rep["older_flip"] = rep.get("older_flip", 0) + int(ok)
            assert ok, (ctx, ch, key, e)
    if all(rep[f"flips{ch}"] == 0 for ch in (0, 1)) and not rep.get("older_flip"):
        dw = np.abs(np.array(wf_h, np.int32) - np.array(wf_o, np.int32))
        assert dw.max() <= 2 and np.mean(dw != 0) < 0.02, (ctx, int(dw.max()), float(np.mean(dw != 0)))
    print(ctx, rep)
