"""CPU, build container only (needs oracle/_ref/ref_harness = the reference's own C files compiled where they lie): the ORACLE against the COMPILED REFERENCE
on the random configurations of tests/test_gpu_random_configs.py -- the same seeds the HIP path is held to the oracle on.  The committed goldens pin the oracle
on 21 planned cases; this walks between them (sizes, windows, averaging, blanker cadence, int32 / mirrored / real input, I/Q calibration, AFC-supplied
frequencies, compute_timf2_powersum, pulse width, fft3 + mix2), so that "HIP == oracle" on a seed means "HIP == reference" on it.
LRH_ORACLE_REF_SEEDS / LRH_ORACLE_REF_SEEDS_EXT: how many of the first / third sweep (default 24 + 16).  Round 6, one long sweep (`-n 6`, 4.6 minutes): 1400 chain
configurations, 1000 limiter, 300 linear-blanker, 300 spur, 300 two-channel cases and 300 batched-entry shapes -- 9 cases off, every one at the edge of a
tolerance and none in the algorithm: a waterfall pixel 9 counts out where 8 are allowed, timf3 at 1.44 of the reference's own distance from the float64 build
(allowed 1.3), timf2_blockpower (sums of squares) at 2.2e-5 (2e-5), the linear blanker's integer noise floor one count apart, one coupled blanker decision on a
sample at its limit, and three fixtures with fewer than 500 baseband samples (compare_with_golden's demand on the fixture, not on the oracle)."""
import importlib.util
import os

import numpy as np
import pytest

from oracle_binding import open_oracle
from paritylib import compare_with_golden, run_case
from test_gpu_random_configs import _open_truth, random_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
pytestmark = pytest.mark.skipif(not os.access(HARNESS, os.X_OK), reason="oracle/_ref/ref_harness is built from /root/reference, in the build container only")

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)

SEEDS = list(range(int(os.environ.get("LRH_ORACLE_REF_SEEDS", "24")))) + list(range(200, 200 + int(os.environ.get("LRH_ORACLE_REF_SEEDS_EXT", "16"))))


def reference_run(d):
    """the compiled reference on the case `d`: what tests/golden/make_golden.py would store for it"""
    import subprocess
    import tempfile
    from refcases import harness_args, make_foldcorr, make_input, make_liminfo
    from refdump import load_dump
    iq, lim = make_input(d), make_liminfo(d)
    with tempfile.TemporaryDirectory() as td:
        fi, fl, fo = (os.path.join(td, x) for x in ("in.bin", "lim.bin", "out.bin"))
        iq.tofile(fi)
        lim.tofile(fl)
        extra = []
        if d["foldcorr_seed"]:
            ff = os.path.join(td, "fold.bin")
            make_foldcorr(d).tofile(ff)
            extra = [f"foldcorr={ff}"]
        subprocess.check_call([HARNESS] + harness_args(d, fi, fl, fo) + extra, stdout=subprocess.DEVNULL)
        ref = load_dump(fo)
    g = {k: ref[k] for k in make_golden.KEEP if k in ref}
    g["iq"], g["liminfo"] = iq, lim
    if d["foldcorr_seed"]:
        g["foldcorr"] = make_foldcorr(d)
    return g


@pytest.mark.parametrize("seed", SEEDS)
def test_oracle_matches_the_compiled_reference_on_a_random_configuration(seed):
    d, _ = random_case(seed)
    d["golden_stride"] = 1
    if d["fft3_n"]:
        d["nblk"] = max(d["nblk"], 240)                         # (enough baseband samples for compare_with_golden's own sanity check of the fixture)
    g = reference_run(d)
    out = run_case(open_oracle, "random", golden=g, params=d)
    def truth():
        t = run_case(_open_truth, "random", golden=g, params=d)
        t["wf_pre"] = np.array(t["api"].wf_pre_lines, np.float64).reshape(-1, t["cfg"].wf_xpixels)
        t.pop("api").close()
        return t
    cache = []

    def T():
        if not cache:
            cache.append(truth())
        return cache[0]
    # 2e-5 / 1.3 / a noise floor one count apart: the planned cases of the goldens hold 2e-6 / 1.05 / 0; of 400 random ones 3 had timf2_pwr (a square: twice the
    # relative error) at 1.1-1.5e-5, 3 the blanker's integer floor one count apart, 1 timf3 at 1.254 of the reference's own distance from the float64 build.
    # The cleared-sample set, every pointer trace and the mixer's bookkeeping stay exact.
    rep = compare_with_golden(out, g, tol=2e-5, truth=T, truth_factor=1.3, skip=("wf_lines",), floor_slack=1)
    # the waterfall lines, each side against the float64 values before truncation, down to 80 dB below the strongest pixel of the RUN (the first line of a start-up
    # with the blanker off is two float32 noise floors over a float64 value of -200 dB: tests/test_gpu_random_configs.py)
    gw, ow = g["wf_lines"].reshape(-1, out["cfg"].wf_xpixels).astype(np.int64), out["wf_lines"].astype(np.int64)
    if gw.size:
        sel = (gw.max() - gw) < 8000
        from paritylib import waterfall_gate
        bad = 0
        for ln in range(gw.shape[0]):
            if sel[ln].any():
                try:
                    waterfall_gate({}, ow[ln][sel[ln]], gw[ln][sel[ln]], T()["wf_pre"][ln][sel[ln]])
                except AssertionError:
                    bad += 1
                    assert np.abs(ow[ln] - gw[ln])[sel[ln]].max() <= 8, (seed, ln)
        rep["wf_lines_off_the_gate"] = bad
    print(seed, {k: d[k] for k in ("n1", "n2", "second_fft", "sinpow1", "sinpow2", "real", "dword", "afc", "foldcorr_seed", "blockpower_block", "stupid", "pulsewidth", "fft3_n")},
          {k: (float("%.2e" % v) if isinstance(v, float) else v) for k, v in rep.items() if not isinstance(v, dict)})


SELLIM_SEEDS = list(range(int(os.environ.get("LRH_ORACLE_REF_SELLIM_SEEDS", "24"))))


@pytest.mark.parametrize("seed", SELLIM_SEEDS)
def test_oracle_limiters_match_the_compiled_reference_on_a_random_case(seed):
    """both selective limiters (sellim.c:159-1157) as the COMPILED REFERENCE runs them, from the first block on, on the random cases of
    tests/test_gpu_random_configs.py (levels, hg.sellim_par1..8, group sizes, keyed carriers) with random in-band end points: the routing table after every
    update -- which bins are weak, strong, attenuated -- exact, the attenuations to 2e-6, the weak-bin counts, the amplitude factor, the rings behind.
    A case whose tables part on one or two bins is a decision at its threshold (the reference's float32 against the oracle's, 1e-7 apart in the sums: the
    random HIP test meets the same about once in a hundred runs): counted, everything up to that update still held exactly."""
    import subprocess
    import tempfile
    import refcases
    import sellimlib
    from refcases import harness_args, sellim_case
    from refdump import load_dump
    from test_gpu_random_configs import random_sellim_case
    mg = importlib.util.spec_from_file_location("make_golden_sellim", os.path.join(ROOT, "tests", "golden", "make_golden_sellim.py"))
    mgs = importlib.util.module_from_spec(mg)
    mg.loader.exec_module(mgs)
    t, _ = random_sellim_case(seed)
    rng = np.random.default_rng(3300 + seed)
    n1 = 1024 if t["base"] == "n10_n12" else 512
    t.update(first_inband=int(rng.integers(0, 40)), last_inband=int(n1 - 1 - rng.integers(0, 40)))
    t.pop("stupid", None)                                       # (the blanker as the reference case has it)
    name = f"random_ref_sellim_{seed}"
    refcases.SELLIM[name] = t
    try:
        d, sl, iq = sellim_case(name)
        with tempfile.TemporaryDirectory() as td:
            fi, fo = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
            iq.tofile(fi)
            args = [a for a in harness_args(d, fi, "none", fo) if not a.startswith("liminfo=")]
            subprocess.check_call([HARNESS] + args + ["sellim=1"] + [f"{k}={v}" for k, v in sl.items()], stdout=subprocess.DEVNULL)
            ref = load_dump(fo)
        g = {k: ref[k] for k in mgs.KEEP if k in ref}
        g["iq"] = iq
        out = sellimlib.run(open_oracle, name, g)
    finally:
        del refcases.SELLIM[name]
    n1_ = out["api"].N1
    ref1, got1 = g["liminfo_trace"].reshape(-1, n1_), out["trace"]
    assert got1.shape == ref1.shape, (got1.shape, ref1.shape)
    bad = np.nonzero((np.sign(got1) != np.sign(ref1)).any(axis=1))[0]
    if "liminfo_trace2" in g:
        ref2, got2 = g["liminfo_trace2"].reshape(-1, n1_), out["trace2"]
        assert got2.shape == ref2.shape, (got2.shape, ref2.shape)
        bad2 = np.nonzero((np.sign(got2) != np.sign(ref2)).any(axis=1))[0]
    else:
        bad2 = np.zeros(0, int)
    if bad.size == 0 and bad2.size == 0:
        rep = sellimlib.compare(out, g, tol=2e-5, value_tol=1e-5)
        print(seed, {k: v for k, v in t.items() if k.startswith("par") or k in ("first_inband", "last_inband", "sellim2", "lim_groups")}, rep)
        return
    # parted tables: how many bins at the first update that differs
    which, r0 = (1, int(bad[0])) if bad.size and (not bad2.size or out["blks"][bad[0]] <= out["blks2"][bad2[0]]) else (2, int(bad2[0]))
    a, b = (got1[r0], ref1[r0]) if which == 1 else (got2[r0], ref2[r0])
    bins = np.nonzero(np.sign(a) != np.sign(b))[0]
    print(seed, "threshold-edge decision: update", which, "number", r0, "bins", bins)
    assert bins.size <= 2, (seed, which, r0, bins[:10])


def _harness(args):
    import subprocess
    r = subprocess.run([HARNESS] + args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stderr


def _load_script(name):
    sp = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tests", "golden", name + ".py"))
    m = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(m)
    return m


@pytest.mark.parametrize("seed", range(int(os.environ.get("LRH_ORACLE_REF_CLEVER_SEEDS", "16"))))
def test_oracle_linear_blanker_matches_the_compiled_reference_on_a_random_pulse_train(seed):
    """the linear blanker (blank1.c:36-1087) of the COMPILED REFERENCE -- its own init_blanker tables, pulse search, fits, subtraction -- on the random pulse
    trains of tests/test_gpu_random_configs.py: the blanker's scalars after every call exact, the rings behind to the goldens' tolerance"""
    import tempfile
    import cleverlib
    import refcases
    from refdump import load_dump
    from test_gpu_random_configs import random_clever_case
    mk = _load_script("make_golden_clever")
    base, t = random_clever_case(seed)
    name = f"random_ref_clever_{seed}"
    refcases.CLEVER[name] = t
    try:
        d, cl, iq, lim, des = refcases.clever_case(name)
        with tempfile.TemporaryDirectory() as td:
            fi, fl, fd, fo = (os.path.join(td, x) for x in ("in.bin", "lim.bin", "des.bin", "out.bin"))
            iq.tofile(fi), lim.tofile(fl), des.tofile(fd)
            _harness(refcases.harness_args(d, fi, fl, fo) + ["clever=1", f"desired={fd}", f"clever_factor={cl['clever_factor']}"])
            ref = load_dump(fo)
        g = {k: ref[k] for k in mk.KEEP if k in ref}
        g["iq"], g["liminfo"], g["desired"] = iq, lim, des
        out = cleverlib.run(open_oracle, name, g)
    finally:
        del refcases.CLEVER[name]
    rep = cleverlib.compare(out, g, tol=1e-5)
    print(seed, base, {k: t[k] for k in ("pulses", "pairs", "rects", "nblk")}, {k: v for k, v in rep.items() if not k.endswith("_first_diff")})


@pytest.mark.parametrize("seed", range(int(os.environ.get("LRH_ORACLE_REF_SPUR_SEEDS", "12"))))
def test_oracle_spur_loop_matches_the_compiled_reference_on_a_random_carrier(seed):
    """spur removal as the COMPILED REFERENCE runs it (store_new_spur / spur_phase_lock / initial_remove_spur, eliminate_spurs) on the random carriers of
    tests/test_gpu_random_configs.py: the same lock decision; where it locks, window and flag after every transform exact, the loop's frequency / phase /
    amplitude, the fft2 ring behind the subtraction, its sums and timf3 to the goldens' tolerances (spurlib.compare)"""
    import tempfile
    import refcases
    import spurlib
    from refdump import load_dump
    from test_gpu_random_configs import random_spur_case
    mk = _load_script("make_golden_spur")
    t, batch = random_spur_case(seed)
    name = f"random_ref_spur_{seed}"
    refcases.SPUR[name] = t
    try:
        d, sp, iq, lim = refcases.spur_case(name)
        with tempfile.TemporaryDirectory() as td:
            fi, fl, fo = (os.path.join(td, x) for x in ("in.bin", "lim.bin", "out.bin"))
            iq.tofile(fi), lim.tofile(fl)
            _harness(refcases.harness_args(d, fi, fl, fo) + ["spur=1"] + [f"{k}={v}" for k, v in sp.items()])
            ref = load_dump(fo)
        g = {k: ref[k] for k in mk.KEEP if k in ref}
        g["iq"], g["liminfo"] = iq, lim
        locked = int(ref["spur_locked"][0]) > 0
        if not locked:                                          # the reference did not lock: neither may the oracle at the same transform
            gg = dict(g, spur_locked=np.array([t["spur_start"]]), spur_init_state=np.concatenate([np.zeros(10), [t["spur_speknum"]], np.zeros(5)]),
                      spur_spectra=spurlib.load("spur_n10_n12")["spur_spectra"])
            with pytest.raises(AssertionError, match="no lock"):
                spurlib.run(open_oracle, name, gg, acquire=True)
            print(seed, t["tone"], "no lock on either side")
            return
        out = spurlib.run(open_oracle, name, g, acquire=True)
    finally:
        del refcases.SPUR[name]
    # The loop is an iterated estimator with convergence tests (spur.c:181-420): two float32 runs of it that take another number of passes end a milliradian
    # apart (3 of 300 carriers here, oracle against reference; 8 of 300 HIP against oracle) -- inside the loop's own noise.  Held: window and flag after every
    # transform exact, the state to the goldens' tolerance, what the oracle leaves of the carrier no more than what the reference leaves, everything away from
    # the window at 1e-5; with states that agree to 1e-5 also the rings behind (spurlib.compare at the goldens' tolerance).
    ref, got = g["spur_trace"].reshape(-1, 12), out["trace"]
    assert got.shape[0] == ref.shape[0] and np.array_equal(got[:, :2], ref[:, :2]), "spur_location / spur_flag trace differs"
    wrap = lambda x: (x + np.pi) % (2 * np.pi) - np.pi  # noqa: E731
    ferr, perr = float(np.max(np.abs(got[:, 2] - ref[:, 2]))), float(np.max(np.abs(wrap(got[:, 3] - ref[:, 3]))))
    aerr = float(np.max(np.abs(got[:, 6] - ref[:, 6]) / np.abs(ref[:, 6])))
    assert ferr <= 1e-3 and perr <= 2e-2 and aerr <= 2e-3, (seed, ferr, perr, aerr)
    cfg_, loc = out["cfg"], int(ref[-1, 0])
    n2_ = 1 << cfg_.fft2_n
    fo, fr = out["fft2"].reshape(cfg_.max_fft2n, n2_, 2).astype(np.float64), g["fft2_float"].reshape(cfg_.max_fft2n, n2_, 2).astype(np.float64)
    win = np.zeros(n2_, bool)
    win[max(0, loc - 2):loc + 10] = True
    res_o, res_r = float(np.linalg.norm(fo[:, win])), float(np.linalg.norm(fr[:, win]))
    e_out = float(np.linalg.norm((fo - fr)[:, ~win]) / np.linalg.norm(fr[:, ~win]))
    assert res_o <= 1.1 * res_r + 1e-5 * np.linalg.norm(fr) and e_out <= 1e-5, (seed, res_o, res_r, e_out)
    rep = {"freq_err_bins": ferr, "phase_err_rad": perr, "ampl_rel": aerr, "residual oracle / reference": res_o / max(res_r, 1e-300), "fft2 outside the window": e_out}
    if perr <= 1e-5 and aerr <= 1e-5:
        rep = spurlib.compare(out, g, tol=1e-4)
        assert rep["fft2"] <= 5e-6 and rep["ps2"] <= 5e-6 and rep["residual_err_vs_carrier"] <= 1e-5, rep
    print(seed, t["tone"], t["spur_speknum"], spurlib.compare_acquisition(out, g), {k: v for k, v in rep.items() if k != "locations"})


@pytest.mark.parametrize("seed", range(int(os.environ.get("LRH_ORACLE_REF_TWOCHAN_SEEDS", "8"))))
def test_oracle_two_channel_chain_matches_the_compiled_reference_on_a_random_case(seed):
    """two RF channels in the COMPILED REFERENCE's own single array (ui.rx_rf_channels = 2: coupled blanker, fft2 cross products, polarisation pair of fft3_mix2)
    against two oracle contexts with the exchanges made by hand, on the random cases of tests/test_gpu_random_configs.py (sky phase, channel-2 phasing,
    polarisation, run length): tests/test_twochan.py's own check of its goldens"""
    import tempfile
    import refcases
    import test_twochan as TC
    from refdump import load_dump
    from test_gpu_random_configs import random_twochan_case
    gname, t, batch = random_twochan_case(seed)
    name = f"random_ref_twochan_{seed}"
    refcases.TWOCHAN[name] = t
    try:
        d, frames, lim = refcases.twochan_case(name, chain=True)
        with tempfile.TemporaryDirectory() as td:
            fi, fl, fo = (os.path.join(td, x) for x in ("in.bin", "lim.bin", "out.bin"))
            frames.tofile(fi), lim.tofile(fl)
            args = refcases.harness_args(d, fi, fl, fo) + ["channels=2", f"ch2_c1={d['ch2_c1']!r}", f"ch2_c2={d['ch2_c2']!r}", "chain2=1"]
            _harness(args + [f"pol_c{i + 1}={v!r}" for i, v in enumerate(d["pol"])])
            g = load_dump(fo)
        d, g, out, wf_lines, nfft2 = TC._run_chain(open_oracle, name, False, batch, golden=g)
    finally:
        del refcases.TWOCHAN[name]
    TC._check_chain(d, g, out, wf_lines, nfft2, 1e-5, fixture_checks=False)
    print(seed, gname, {k: t[k] for k in ("sky_phase", "ch2_c1", "ch2_c2")}, t["chain"]["pol"], "batch", batch)


@pytest.mark.parametrize("seed", range(int(os.environ.get("LRH_ORACLE_DSP_SEEDS", "24"))))
def test_oracle_wideband_dsp_is_its_own_stage_calls_on_a_random_configuration(seed):
    """lro_wideband_dsp -- what the GPU tests hold lrh_wideband_dsp to, and the bench's cpu_baseline port -- against the oracle's stage calls in the reference's
    single-CPU order (which the tests above hold to the compiled reference), on the random configurations: every ring and pointer bit for bit"""
    from paritylib import RINGS
    from refcases import lrh_config, make_foldcorr, make_input, make_liminfo
    d, batch = random_case(seed if seed < 12 else 188 + seed)
    d.update(afc=0, blockpower_block=0)
    g = {"iq": make_input(d), "liminfo": make_liminfo(d)}
    if d["foldcorr_seed"]:
        g["foldcorr"] = make_foldcorr(d)
    B = batch                                                   # (blocks per pass: the stage calls take the same, gpu.fft1_batch_n of the reference's loop)
    a = run_case(open_oracle, "random", golden=g, params=d, batch=B)
    rx = open_oracle(lrh_config(d, g["iq"]))
    rx.timf1_write(g["iq"])
    rx.set_liminfo(g["liminfo"])
    if d["foldcorr_seed"]:
        rx.set_foldcorr(g["foldcorr"])
    rx.set_mix1_selfreq(d["fq"])
    if d["fft3_n"]:
        n3 = 1 << d["fft3_n"]
        rx.set_bg_filterfunc(np.exp(-((np.arange(n3) - n3 / 2) / (n3 / 6.0)) ** 2).astype(np.float32))
    rx.wideband_dsp(d["nblk"], B)
    pa, pb = a["api"].p.as_dict(), rx.p.as_dict()
    assert pa == pb, {k: (pa[k], pb[k]) for k in pa if pa[k] != pb[k]}
    for ring, key in RINGS:
        if not d["second_fft"] and key.startswith(("timf2", "fft2")):
            continue
        assert np.array_equal(a[key], rx.export(ring)), (seed, key)
    rx.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("LRH_ORACLE_REF_CLEVER2_SEEDS", "8"))))
def test_oracle_two_channel_linear_blanker_matches_the_compiled_reference_on_a_random_pulse_train(seed):
    """the linear blanker on two coupled channels (get_pulse_pol, transform_timf2_pol, subtract_twochan_pulse, blank1.c:232-609) of the COMPILED REFERENCE on
    random pulse trains, sky phase and channel gain: tests/clever2lib.py's own check of its goldens"""
    import tempfile
    import clever2lib
    import refcases
    from refdump import load_dump
    from test_gpu_random_configs import random_clever2_case
    mk = _load_script("make_golden_clever2")
    n1_, t1, n2_, t2, _ = random_clever2_case(seed)
    refcases.CLEVER[n1_], refcases.CLEVER2[n2_] = t1, t2
    try:
        d, cl, frames, lim, des = refcases.clever2_case(n2_)
        with tempfile.TemporaryDirectory() as td:
            fi, fl, fd, fo = (os.path.join(td, x) for x in ("in.bin", "lim.bin", "des.bin", "out.bin"))
            frames.tofile(fi), lim.tofile(fl), des.tofile(fd)
            _harness(refcases.harness_args(d, fi, fl, fo) + ["channels=2", "ch2_c1=1.0", "ch2_c2=0.0", "blanker2=1", "clever=1", f"desired={fd}", f"clever_factor={cl['clever_factor']}"])
            ref = load_dump(fo)
        g = {k: ref[k] for k in mk.KEEP}
        g["frames"], g["liminfo"], g["desired"] = frames, lim, des
        res = clever2lib.run(open_oracle, n2_, g, frames_mode=False)
        try:
            rep = clever2lib.compare(res, g, 1e-5)
        except AssertionError:
            # the oracle forms the channel power sum in another order than the reference's four-term sum (clever2lib.compare): a sample at the stupid blanker's
            # limit can go the other way -- one or two samples, every scalar of every call still equal (3 of 60 cases)
            psum = res["out"][0]["pwr"] + res["out"][1]["pwr"]
            keep = np.ones(psum.size, bool)
            keep[(res["out"][0]["p"]["timf2_pa"] // 4 + np.arange(res["rxs"][0].N1 // 2)) % keep.size] = False
            # (a power of 1e-14 where the other side has 0 -- what subtract_twochan_pulse leaves of a sample it takes out twice, seed 28 -- is no decision)
            nflip = int(np.count_nonzero(((psum < 1e-9) != (g["timf2_pwr_float"] < 1e-9)) & keep))
            if nflip > 2:
                raise
            rep = {"samples at the limit that went the other way": nflip}
        for rx in res["rxs"]:
            rx.close()
    finally:
        del refcases.CLEVER[n1_], refcases.CLEVER2[n2_]
    print(seed, {k: t2[k] for k in ("sky_phase", "gain", "nblk")}, rep)


@pytest.mark.parametrize("seed", range(int(os.environ.get("LRH_ORACLE_REF_TWOCHAN1_SEEDS", "8"))))
def test_oracle_two_channel_first_stages_match_the_compiled_reference_on_a_random_case(seed):
    """two RF channels, I/Q or real, through fft1 (fft1win_dif_chan / fft1_reherm_dit_two), the channel power sums and make_timf2 (fft1back_two), and -- I/Q --
    the coupled first_noise_blanker after every block, in the COMPILED REFERENCE's single array against two oracle contexts: tests/test_twochan.py's own
    checks of its goldens, on random sky phase, channel-2 phasing and run length"""
    import tempfile
    import refcases
    import test_twochan as TC
    from refdump import load_dump
    rng = np.random.default_rng(2600 + seed)
    base = str(rng.choice(["twochan_n10", "twochan_n9_sin3", "twochan_real_n9"]))
    t = dict(refcases.TWOCHAN[base])
    ang = float(rng.uniform(-3.1, 3.1))
    t.update(seed2=int(2700 + seed), nblk=int(rng.choice([24, 32, 40, 56])))
    if base != "twochan_real_n9":
        t.update(sky_phase=float(rng.uniform(-3.1, 3.1)), ch2_c1=float(np.float32(np.cos(ang))), ch2_c2=float(np.float32(np.sin(ang))))
    name = f"random_ref_twochan1_{seed}"
    refcases.TWOCHAN[name] = t
    try:
        d, frames, lim = refcases.twochan_case(name)
        with tempfile.TemporaryDirectory() as td:
            fi, fl, fo = (os.path.join(td, x) for x in ("in.bin", "lim.bin", "out.bin"))
            frames.tofile(fi), lim.tofile(fl)
            args = refcases.harness_args(d, fi, fl, fo) + ["channels=2", f"ch2_c1={d['ch2_c1']!r}", f"ch2_c2={d['ch2_c2']!r}"]
            _harness(args)
            ref = load_dump(fo)
            g = {k: ref[k] for k in ("fft1_float", "fft1_sumsq", "fft1_slowsum", "timf2_float", "timf2_pwr_float", "itrace", "trace")}
            if not d["real"]:
                _harness(args + ["blanker2=1"])
                refb = load_dump(fo)
                for k in ("timf2_float", "timf2_pwr_float", "itrace", "trace"):
                    g["bln_" + k] = refb[k]
        g["frames"], g["liminfo"] = frames, lim
        d1, _, out = TC._run(open_oracle, name, False, golden=g)
        TC._check(d1, g, out, 2e-6)
        if not d["real"]:
            d2, _, out2, trace = TC._run_coupled(open_oracle, name, False, golden=g)
            TC._check_coupled(d2, g, out2, trace, 2e-6)
    finally:
        del refcases.TWOCHAN[name]
    print(seed, base, {k: t[k] for k in ("nblk", "sky_phase", "ch2_c1", "ch2_c2")})
