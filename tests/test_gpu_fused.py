"""GPU: k_fft1w -- forward transform, fft1_c's sums and the weak stream of make_timf2 as one kernel inside lrh_wideband_dsp
(fft1_size 16384, sin^2 window, int16 I/Q) with the sparse strong-stream pass behind it -- against the two-kernel path it replaces
(k_fft1 + k_timf2<.., SS>, LRH_FUSE_FFT1=0), and cfg.fft1_float_sparse against the full spectrum ring.  Parity with the oracle at
these sizes is tests/test_gpu_fullsize.py: test_fullsize_fused_kernel_matches_oracle (rounds of 32 blocks, full and sparse rings, launch
counters asserted) for this kernel, test_fullsize_chain_matches_oracle (rounds of 16 blocks) for the two-kernel path."""
import os

import numpy as np
import pytest

from linrad_amd import abi
from linrad_amd.workload import chain_config, strong_liminfo

pytestmark = pytest.mark.gpu
N1 = 16384
RINGS = [(abi.RING_FFT1_FLOAT, "fft1"), (abi.RING_FFT1_SUMSQ, "sumsq"), (abi.RING_FFT1_SLOWSUM, "slowsum"), (abi.RING_TIMF2_FLOAT, "timf2"),
         (abi.RING_TIMF2_PWR, "pwr"), (abi.RING_FFT2_FLOAT, "fft2"), (abi.RING_FFT2_POWERSUM, "ps2"), (abi.RING_TIMF3_FLOAT, "timf3")]


def _run(env, sparse=0, nblk=160, batch=32, calls=1, blanker=True, change_table_at=None, fft2_n=12, sparse2=0, fft3_n=0, sd=None, more_strong=0,
         fft1_n=14, fn=None):
    if sd is not None:                                      # LRH_SD is read at every launch of the strong pass: set for the whole run
        keep_sd = os.environ.get("LRH_SD")
        os.environ["LRH_SD"] = sd
        try:
            return _run(env, sparse, nblk, batch, calls, blanker, change_table_at, fft2_n, sparse2, fft3_n, None, more_strong, fft1_n, fn)
        finally:
            os.environ.pop("LRH_SD", None) if keep_sd is None else os.environ.__setitem__("LRH_SD", keep_sd)
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    N1 = 1 << fft1_n
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        cfg = chain_config(fft1_n, fft2_n, batch=batch, rounds=nblk // batch, fft3_n=fft3_n, mix2_n=8 if fft3_n else 0)
        cfg.fft1_float_sparse = sparse
        cfg.fft2_float_sparse = sparse2
        if not blanker:
            cfg.stupid_bln_mode = 0
        rx = (fn or open_hip)(cfg)                          # the environment is read here
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    s = synth_defaults(N1, 0)
    rx.timf1_write(synth_iq(s, 0, cfg.timf1_bytes // 4))
    lim = strong_liminfo(s, fft1_n)
    if more_strong:                                         # a stretch of the band routed strong as well (more bins than k_timf2_sd takes: LRH_SD_KMAX 128)
        lim[N1 // 3:N1 // 3 + more_strong] = 1.0
    rx.set_liminfo(lim)
    rx.set_mix1_selfreq(0.31 * (1 << fft2_n) + 0.3)
    per = nblk // calls
    for i in range(calls):
        if change_table_at is not None and i == change_table_at:       # the routing table changes between two calls: the first transform of
            lim2 = lim.copy(); lim2[N1 // 5:N1 // 5 + 10] = 1.0        # the next call overlaps a partner routed with the OLD table
            rx.set_liminfo(lim2)
        rx.wideband_dsp(per, batch)
    out = {k: rx.export(r) for r, k in RINGS}
    out["wf"] = rx.export(abi.RING_WG_WATERF)
    if fft3_n:
        out["baseb"] = rx.export(abi.RING_BASEB_RAW)
    out["cfg"] = cfg
    out["p"] = rx.p.as_dict()
    out["lim"] = rx.get_liminfo()
    rx.close()
    return out


def _tol(k):
    """two HIP paths against each other, float32 rounding: 2e-6; timf3 is a weak band cut out next to carriers 50 dB up, and where the
    paths form the strong stream differently (k_timf2_sd's direct sum / a transform) the CARRIERS' rounding is 2.1e-6 of that band"""
    return 3e-6 if k == "timf3" else 2e-6


def _rel(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.mark.parametrize("pipeline", ["0", "2"])
def test_fused_kernel_matches_the_two_kernel_path(pipeline):
    """same job through k_fft1w + strong pass and through k_fft1 + k_timf2<.., SS>: every ring to float32 rounding (the 32-point and
    the 16-point per thread forms of the transform factor their last-pass twiddles differently), pointers identical"""
    a = _run({"LRH_FUSE_FFT1": "1", "LRH_PIPELINE": pipeline}, blanker=False)
    b = _run({"LRH_FUSE_FFT1": "0", "LRH_PIPELINE": pipeline}, blanker=False)
    assert a["p"] == b["p"]
    rep = {k: _rel(a[k], b[k]) for _, k in RINGS}
    print(rep)
    assert np.count_nonzero(a["timf3"]) > 100 and np.count_nonzero(a["sumsq"]) > N1
    for k, e in rep.items():
        assert e < _tol(k), (k, e)


def test_sparse_spectrum_ring_changes_nothing_downstream():
    """cfg.fft1_float_sparse: only the strong bins reach the fft1 ring; every product of the chain is bit-identical to the full-ring run"""
    full = _run({"LRH_FUSE_FFT1": "1"}, sparse=0)
    sp = _run({"LRH_FUSE_FFT1": "1"}, sparse=1)
    assert full["p"] == sp["p"]
    for _, k in RINGS[1:]:
        assert np.array_equal(full[k], sp[k]), k
    strong = np.nonzero(full["lim"])[0]
    f, s = full["fft1"].reshape(-1, N1, 2), sp["fft1"].reshape(-1, N1, 2)
    assert strong.size > 10 and np.array_equal(f[:, strong], s[:, strong])
    weak = np.nonzero(full["lim"] == 0)[0]
    assert not np.any(s[:, weak])                          # never written: the ring's initial zeros


@pytest.mark.parametrize("sparse", [0, 1])
def test_fused_path_is_the_same_in_one_call_and_in_many(sparse):
    """the overlap partner of a call's first transform is recomputed from the samples of the previous call's last block: five calls of
    one round each, with the routing table changed in between, equal the two-kernel path call for call"""
    # (blanker off for the comparison across the two paths: a sample within float32 rounding of the limit may be cleared in one only)
    a = _run({"LRH_FUSE_FFT1": "1"}, sparse=sparse, calls=5, change_table_at=3, blanker=False)
    b = _run({"LRH_FUSE_FFT1": "0"}, sparse=0, calls=5, change_table_at=3, blanker=False)
    assert a["p"] == b["p"]
    for _, k in RINGS[3:]:
        assert _rel(a[k], b[k]) < (8e-6 if k == "pwr" else _tol(k)), k      # pwr: a squared quantity, despiked (blanker on)
    one = _run({"LRH_FUSE_FFT1": "1"}, sparse=sparse, calls=1)
    many = _run({"LRH_FUSE_FFT1": "1"}, sparse=sparse, calls=5)
    for _, k in RINGS[1:]:
        assert np.array_equal(one[k], many[k]), k


@pytest.mark.parametrize("fft2_n,fft3_n", [(12, 0), (16, 12)])
def test_sparse_fft2_ring_changes_nothing_downstream(fft2_n, fft3_n):
    """cfg.fft2_float_sparse: of every fft2 transform only the band fft2_mix1_fixed cuts out reaches the ring; power sums, waterfall
    lines, timf3 and the baseband are bit-identical to the full-ring run, and the stored band equals the full ring's"""
    full = _run({}, sparse=1, sparse2=0, fft2_n=fft2_n, fft3_n=fft3_n, nblk=256 if fft3_n else 160)
    sp = _run({}, sparse=1, sparse2=1, fft2_n=fft2_n, fft3_n=fft3_n, nblk=256 if fft3_n else 160)
    assert full["p"] == sp["p"]
    for k in ("ps2", "timf3", "wf", "pwr", "timf2") + (("baseb",) if fft3_n else ()):
        assert np.array_equal(full[k], sp[k]), k
    assert np.count_nonzero(sp["timf3"]) > 100 and np.any(sp["wf"])
    n2 = 1 << fft2_n
    centre, half = int(0.31 * n2 + 0.3 + 0.5), (n2 >> 6) // 2
    f, s = full["fft2"].reshape(-1, n2, 2), sp["fft2"].reshape(-1, n2, 2)
    assert np.array_equal(f[:, centre - half:centre + half], s[:, centre - half:centre + half])
    assert not np.any(s[:, :centre - half - 64]) and not np.any(s[:, centre + half + 64:])


def test_sparse_fft2_ring_refuses_a_band_it_did_not_keep():
    """the selected frequency moves between lrh_make_fft2 and lrh_fft2_mix1_fixed: an error, not silence"""
    from linrad_amd.abi import LrhError
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    cfg = chain_config(14, 12, batch=32)
    cfg.fft2_float_sparse = 1
    rx = open_hip(cfg)
    rx.timf1_write(synth_iq(synth_defaults(N1, 0), 0, cfg.timf1_bytes // 4))
    rx.set_liminfo(np.zeros(N1, np.float32))
    rx.set_mix1_selfreq(1000.3)
    rx.fft1_b(32), rx.fft1_c(32), rx.make_timf2(32), rx.first_noise_blanker()
    k = rx.fft2_available()
    assert k >= 1
    rx.make_fft2(1)
    rx.set_mix1_selfreq(3000.0)
    with pytest.raises((LrhError, RuntimeError), match="fft2_float_sparse"):
        rx.fft2_mix1_fixed(1)
    rx.set_mix1_selfreq(1010.0)                           # within the 64-bin margin: served
    rx.fft2_mix1_fixed(1)
    rx.close()


def _run15(env, sparse, nblk=48, batch=16, calls=1):
    """fft1_size 32768 / fft2_size 131072 through lrh_wideband_dsp (the four-step path)"""
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        cfg = chain_config(15, 17, batch=batch, rounds=nblk // batch)
        cfg.fft1_float_sparse = sparse
        cfg.stupid_bln_mode = 0
        rx = open_hip(cfg)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    s = synth_defaults(32768, 0)
    rx.timf1_write(synth_iq(s, 0, cfg.timf1_bytes // 4))
    rx.set_liminfo(strong_liminfo(s, 15))
    rx.set_mix1_selfreq(0.31 * (1 << 17) + 0.3)
    for _ in range(calls):
        rx.wideband_dsp(nblk // calls, batch)
    out = {k: rx.export(r) for r, k in RINGS}
    out["p"] = rx.p.as_dict()
    rx.close()
    return out


@pytest.mark.parametrize("calls,pipeline", [(1, "0"), (3, "0"), (1, "2"), (3, "2")])
def test_fused_row_column_kernel_at_32768_equals_the_separate_kernels(calls, pipeline):
    """k_fft1r_t2c (row step of fft1 + fft1_c's sums + column step of both timf2 streams) against k_fft1_rows, k_sumsq and k_timf2_cols:
    the same butterflies on the same data in the same order -- every ring bit for bit; the sums too (same additions in the same order).
    With cfg.fft1_float_sparse the spectrum stays off the ring and nothing downstream changes."""
    a = _run15({"LRH_FUSE_FFT1": "1", "LRH_PIPELINE": pipeline}, sparse=0, calls=calls)       # "2": the one-round-late two-stream schedule, forced
    b = _run15({"LRH_FUSE_FFT1": "0", "LRH_PIPELINE": "0"}, sparse=0, calls=calls)
    c = _run15({"LRH_FUSE_FFT1": "1", "LRH_PIPELINE": pipeline}, sparse=1, calls=calls)
    assert a["p"] == b["p"] == c["p"]
    assert np.count_nonzero(a["timf3"]) > 100 and np.count_nonzero(a["sumsq"]) > 32768
    for _, k in RINGS:
        assert np.array_equal(a[k], b[k]), k
        if k != "fft1":
            assert np.array_equal(a[k], c[k]), k
    assert np.count_nonzero(c["fft1"]) < np.count_nonzero(a["fft1"])


def test_rounds_of_a_few_blocks_between_fused_rounds():
    """without LRH_FUSE_FFT1 the library picks the kernel by the round's length (k_fft1w from 32 blocks, k_fft1 + k_timf2 below): a
    caller that mixes long and short rounds crosses from one to the other and back -- the overlap partner comes from the fft1 ring one
    way and is recomputed from the samples the other way -- and must end where the two-kernel path ends"""
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq

    def run(env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        os.environ.pop("LRH_FUSE_FFT1", None) if "LRH_FUSE_FFT1" not in env else None
        try:
            cfg = chain_config(14, 12, batch=32, rounds=5)
            cfg.stupid_bln_mode = 0
            rx = open_hip(cfg)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        s = synth_defaults(N1, 0)
        rx.timf1_write(synth_iq(s, 0, cfg.timf1_bytes // 4))
        rx.set_liminfo(strong_liminfo(s, 14))
        rx.set_mix1_selfreq(0.31 * 4096 + 0.3)
        for nblk, batch in ((64, 32), (16, 8), (3, 1), (64, 32)):
            rx.wideband_dsp(nblk, batch)
        out = {k: rx.export(r) for r, k in RINGS}
        out["p"] = rx.p.as_dict()
        rx.close()
        return out
    a, b = run({}), run({"LRH_FUSE_FFT1": "0"})
    assert a["p"] == b["p"]
    for _, k in RINGS:
        assert _rel(a[k], b[k]) < _tol(k), k
    assert not np.array_equal(a["timf2"], b["timf2"])      # (the fused kernel did run: its last-pass twiddles round differently)


def _run_n(env, fft1_n, fft2_n, dword, fn=None, nblk=96, batch=32, sparse=0, real=0):
    """fft1_size 2^fft1_n through lrh_wideband_dsp in rounds of `batch`; int16 or int32 samples (the int16 signal left-justified in 32 bits,
    like hardware that delivers 24-bit words does, fft1.c:4656-4663)"""
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        cfg = chain_config(fft1_n, fft2_n, batch=batch, rounds=nblk // batch)
        cfg.fft1_float_sparse = sparse
        cfg.stupid_bln_mode = 0
        cfg.timf1_dword_input = dword
        cfg.timf1_real_input = real
        if dword:
            cfg.timf1_bytes *= 2
        rx = (fn or open_hip)(cfg)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    n1 = 1 << fft1_n
    s = synth_defaults(n1, 0)
    iq = synth_iq(s, 0, cfg.timf1_bytes // (8 if dword else 4))
    if real:                                               # real samples: noise + cosines at bins of the half spectrum 0 .. fs/2 (the ring holds 2 reals where it held I and Q)
        n = iq.size
        t = np.arange(n, dtype=np.float64)
        rng = np.random.default_rng(11)
        x = rng.normal(0, 40.0, n)
        for k, a_ in ((0.061 * n1, 5000.0), (0.31 * n1 + 0.45, 900.0), (0.5 * n1 + 3.25, 2500.0), (0.93 * n1 + 0.5, 300.0)):
            x += a_ * np.cos(2 * np.pi * k * t / (2 * n1) + rng.uniform(0, 6.28))
        iq = np.clip(np.round(x), -32767, 32767).astype(np.int16)
    rx.timf1_write((iq.astype(np.int32) << 16) if dword else iq)
    rx.set_liminfo(strong_liminfo(s, fft1_n))
    rx.set_mix1_selfreq(0.31 * (1 << fft2_n) + 0.3)
    launches = None
    if fn is None:
        rx.profile_enable(2)
    rx.wideband_dsp(nblk, batch)
    if fn is None:
        launches = {k: rx.profile_get(k)[1] for k in ("fft1w", "fft1", "timf2", "timf2s")}
        rx.profile_enable(0)
    out = {k: rx.export(r) for r, k in RINGS}
    out["p"] = rx.p.as_dict()
    out["launches"] = launches
    rx.close()
    return out


@pytest.mark.parametrize("fft1_n,fft2_n,dword", [(14, 12, 1), (13, 15, 0), (13, 12, 1), (12, 14, 0), (12, 10, 1)])
def test_fused_kernel_at_other_sizes_and_int32_matches_the_two_kernel_path_and_the_oracle(fft1_n, fft2_n, dword):
    """k_fft1v serves fft1_size 4096 / 8192 / 16384 (Linrad sizes fft1 from the bandwidth, buf.c:139-335: 4096 and 8192 are the common sizes at
    2 - 10 Msps) and int32 samples (fft1.c:4656-4663): against k_fft1 + k_timf2<.., SS> ring by ring, against the oracle at 1e-5, and with the
    sparse spectrum ring nothing downstream changes"""
    from oracle_binding import open_oracle
    a = _run_n({"LRH_FUSE_FFT1": "1"}, fft1_n, fft2_n, dword)
    b = _run_n({"LRH_FUSE_FFT1": "0"}, fft1_n, fft2_n, dword)
    o = _run_n({}, fft1_n, fft2_n, dword, fn=open_oracle)
    assert a["launches"]["fft1w"] == 3 and a["launches"]["fft1"] == 0 and b["launches"]["fft1w"] == 0, (a["launches"], b["launches"])
    assert a["p"] == b["p"] == o["p"]
    n1 = 1 << fft1_n
    keep = np.ones(a["timf2"].size, bool)                  # the raw half block parked beyond timf2_pa (timf2.c:1018-1025): only the oracle stores it
    keep[(a["p"]["timf2_pa"] + np.arange(4 * (n1 // 2))) % keep.size] = False
    rep = {k: (_rel(a[k], b[k]), _rel(a[k] * (keep if k == "timf2" else 1), o[k] * (keep if k == "timf2" else 1))) for _, k in RINGS}
    print(fft1_n, dword, rep)
    assert np.count_nonzero(a["timf3"]) > 100 and np.count_nonzero(a["sumsq"]) > n1
    for k, (e2, eo) in rep.items():
        assert e2 < _tol(k) and eo < 1e-5, (k, e2, eo)
    sp = _run_n({"LRH_FUSE_FFT1": "1"}, fft1_n, fft2_n, dword, sparse=1)
    for _, k in RINGS[1:]:
        assert np.array_equal(a[k], sp[k]), k


@pytest.mark.parametrize("fft1_n,fft2_n,dword", [(14, 12, 0), (13, 12, 1), (12, 14, 0)])
def test_fused_kernel_with_real_samples_matches_the_two_kernel_path_and_the_oracle(fft1_n, fft2_n, dword):
    """real input (fft1 version 2, fft1_reherm_dit_one, fft1_re.c:32-131) inside the fused kernel: k_fft1v<.., REAL> = the N-point transform of the sample
    pairs + one more exchange for S[N-k] + k_realsplit's even / odd split, then the sums, the weak stream and the overlap as for I/Q.  Against
    k_fft1<REAL> + k_realsplit + k_timf2<.., SS> (LRH_FUSE_REAL=0) ring by ring, against the oracle at 1e-5, sparse ring bit-identical downstream"""
    from oracle_binding import open_oracle
    a = _run_n({"LRH_FUSE_FFT1": "1"}, fft1_n, fft2_n, dword, real=1)
    b = _run_n({"LRH_FUSE_FFT1": "1", "LRH_FUSE_REAL": "0"}, fft1_n, fft2_n, dword, real=1)
    o = _run_n({}, fft1_n, fft2_n, dword, fn=open_oracle, real=1)
    assert a["launches"]["fft1w"] == 3 and a["launches"]["fft1"] == 0 and b["launches"]["fft1w"] == 0 and b["launches"]["fft1"] > 0, (a["launches"], b["launches"])
    assert a["p"] == b["p"] == o["p"]
    n1 = 1 << fft1_n
    keep = np.ones(a["timf2"].size, bool)
    keep[(a["p"]["timf2_pa"] + np.arange(4 * (n1 // 2))) % keep.size] = False
    rep = {k: (_rel(a[k], b[k]), _rel(a[k] * (keep if k == "timf2" else 1), o[k] * (keep if k == "timf2" else 1))) for _, k in RINGS}
    print(fft1_n, dword, rep)
    assert np.count_nonzero(a["timf3"]) > 100 and np.count_nonzero(a["sumsq"]) > n1
    for k, (e2, eo) in rep.items():
        assert e2 < _tol(k) and eo < 1e-5, (k, e2, eo)
    sp = _run_n({"LRH_FUSE_FFT1": "1"}, fft1_n, fft2_n, dword, sparse=1, real=1)
    assert sp["launches"]["fft1w"] == 3
    for _, k in RINGS[1:]:
        assert np.array_equal(a[k], sp[k]), k


def _run_gap(fn, sparse, table):
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    cfg = chain_config(14, 12, batch=32, rounds=4)
    cfg.fft1_float_sparse = sparse
    cfg.stupid_bln_mode = 0
    rx = (fn or open_hip)(cfg)
    n1 = 1 << 14
    s = synth_defaults(n1, 0)
    rx.timf1_write(synth_iq(s, 0, cfg.timf1_bytes // 4))
    rx.set_liminfo(strong_liminfo(s, 14))
    rx.set_mix1_selfreq(0.31 * 4096 + 0.3)
    if fn is None:
        rx.profile_enable(2)
    rx.wideband_dsp(32, 32)
    if table:
        fc = np.zeros(2 * n1, np.float32)
        g = 1.0 / (150.0 * n1 * float(n1) ** -0.4)         # the level of the default table (make_filcorrstart, fft1.c:4653-4663)
        fc[0::2] = g * (1.0 + 0.25 * np.cos(np.arange(n1) * 0.001))
        fc[1::2] = 0.1 * g
        rx.set_filtercorr(fc)
    else:
        rx.p.timf1p_px = (rx.p.timf1p_px + 5 * (n1 // 2) * 4 + 4 * 1234) & (cfg.timf1_bytes - 1)   # the producer skipped ahead
    rx.wideband_dsp(64, 32)
    launches = {k: rx.profile_get(k)[1] for k in ("fft1w", "fft1")} if fn is None else None
    out = {k: rx.export(r) for r, k in RINGS}
    out["p"] = rx.p.as_dict()
    rx.close()
    return out, launches


@pytest.mark.parametrize("table", [0, 1])
def test_a_gap_in_the_walk_over_timf1_does_not_reach_the_fused_kernels_partner(table):
    """The fused kernel rebuilds the overlap partner of a call's first transform from the timf1 ring one block behind it.  After a jump of
    timf1p_px, or new filter tables, that is no longer the transform the reference overlaps with (it carries the previous transform's back
    half, timf2.c:1018-1025).  With the whole spectrum kept the library takes the two-kernel path for that one call and stays on the
    oracle; with the sparse ring the call starts over like the first of a stream and only its first transform's share differs."""
    from oracle_binding import open_oracle
    a, la = _run_gap(None, 0, table)
    o, _ = _run_gap(open_oracle, 0, table)
    assert la == {"fft1w": 2, "fft1": 1}, la               # first call fused, the call after the gap in two kernels, then fused again
    assert a["p"] == o["p"]
    n1 = 1 << 14
    keep = np.ones(a["timf2"].size, bool)
    keep[(a["p"]["timf2_pa"] + np.arange(4 * (n1 // 2))) % keep.size] = False
    for _, k in RINGS:
        m = keep if k == "timf2" else 1
        assert _rel(a[k] * m, o[k] * m) < 1e-5, k
    sp, ls = _run_gap(None, 1, table)
    assert ls == {"fft1w": 3, "fft1": 0}, ls
    first = np.zeros(a["timf2"].size, bool)
    first[32 * 4 * (n1 // 2) + np.arange(4 * (n1 // 2))] = True     # timf2 floats of transform 32, the first after the gap
    weak = (np.arange(a["timf2"].size) & 2) == 0                      # [weak re, weak im, strong re, strong im] per sample
    e_rest = _rel(sp["timf2"] * ~first, a["timf2"] * ~first)          # (that call ran as two kernels in `a`: rounding apart, not bit-equal)
    e_weak = _rel(sp["timf2"] * (first & weak), a["timf2"] * (first & weak))
    e_strong = _rel(sp["timf2"] * (first & ~weak), a["timf2"] * (first & ~weak))
    print(table, e_rest, e_weak, e_strong)
    assert e_rest < 2e-6 and e_weak > 0.1, (e_rest, e_weak, e_strong)


def test_one_block_calls_with_the_tail_on_the_side_stream_equal_the_serial_order():
    """One fft1 block per lrh_wideband_dsp call -- Linrad's own call pattern: the blanker (rate-limited to every fourth call), fft2, mix1, fft3 and
    mix2 of such calls are issued on the side stream, the sums of the one-workgroup k_timf2 launch go straight to the ring and its two streams run as
    two workgroups.  Bit for bit what the serial order gives (LRH_SIDE_TAIL=0), with the selective limiter in the call, stage calls and
    exports in between (they order the main stream behind the tail), and a few larger rounds mixed in."""
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    from linrad_amd.abi import default_sellim

    def run(env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            cfg = chain_config(14, 16, batch=4, rounds=32, fft3_n=12, mix2_n=8)
            rx = open_hip(cfg)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        s = synth_defaults(N1, 0)
        rx.timf1_write(synth_iq(s, 0, cfg.timf1_bytes // 4))
        rx.set_liminfo(strong_liminfo(s, 14))
        rx.set_mix1_selfreq(0.31 * 65536 + 0.3)
        rx.wideband_limiter(default_sellim(cfg, fft1_blocktime=8192 / 160e6, blanker_ston_fft1=30.0, exact_stats=1), False)   # exact_stats: the weak-bin count is read back at once (otherwise it arrives when it arrives)
        mid = None
        for i in range(96):
            rx.wideband_dsp(1, 1)
            if i == 40:
                mid = rx.export(abi.RING_TIMF3_FLOAT).copy()        # an entry point other than lrh_wideband_dsp: joins the side stream
            if i == 60:
                rx.wideband_dsp(8, 4)
        out = {k: rx.export(r) for r, k in RINGS}
        out["mid"] = mid
        out["p"] = rx.p.as_dict()
        out["bs"] = rx.blanker_state().timf2_noise_floor
        rx.close()
        return out
    a, b = run({"LRH_SIDE_TAIL": "1"}), run({"LRH_SIDE_TAIL": "0"})
    assert a["p"] == b["p"] and a["bs"] == b["bs"]
    assert np.count_nonzero(a["timf3"]) > 100 and np.count_nonzero(a["mid"]) > 10
    assert np.array_equal(a["mid"], b["mid"])
    for _, k in RINGS:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("fft1_n,sparse", [(14, 0), (14, 1), (13, 1), (12, 1)])
def test_strong_stream_by_direct_summation_equals_the_transform_kernel(fft1_n, sparse):
    """k_timf2_sd (the strong bins summed directly, no transform) against k_timf2<.., STRONG_ONLY> (LRH_SD=0) on the same calls -- five calls
    of one round each with the routing table changed in between, so that a call's first block overlaps a partner routed with the OLD table:
    the strong half of timf2 to float32 rounding, everything the strong pass does not touch bit for bit, what follows it to rounding."""
    kw = dict(sparse=sparse, calls=5, change_table_at=3, blanker=False, fft1_n=fft1_n, fft2_n=fft1_n - 2)
    a = _run({"LRH_FUSE_FFT1": "1"}, sd="1", **kw)
    b = _run({"LRH_FUSE_FFT1": "1"}, sd="0", **kw)
    assert a["p"] == b["p"]
    ta, tb = a["timf2"].reshape(-1, 4), b["timf2"].reshape(-1, 4)
    assert np.array_equal(ta[:, :2], tb[:, :2]) and np.array_equal(a["pwr"], b["pwr"]) and np.array_equal(a["sumsq"], b["sumsq"])
    assert np.count_nonzero(tb[:, 2:]) > 1000
    e = _rel(ta[:, 2:], tb[:, 2:])
    from oracle_binding import open_truth
    t = _run({}, fn=open_truth, **kw)                        # the float64 build of the oracle on the same calls
    npa = t["p"]["timf2_pa"] // 4
    tt = t["timf2"].reshape(-1, 4)
    ea, eb = _rel(ta[:npa, 2:], tt[:npa, 2:]), _rel(tb[:npa, 2:], tt[:npa, 2:])
    print("strong stream: direct sum against the transform %.3e; against the float64 truth: direct sum %.3e, transform %.3e" % (e, ea, eb))
    assert e < 1e-6 and ea < 5e-7 and ea < 1.5 * eb
    for k in ("fft2", "ps2", "timf3"):
        assert _rel(a[k], b[k]) < _tol(k), k


def test_more_strong_bins_than_the_direct_sum_takes_go_through_the_transform_kernel():
    """more than LRH_SD_KMAX bins routed strong: k_timf2_sd returns and the transform kernel behind it does the launch -- the same
    samples, bit for bit, as with the direct sum switched off"""
    a = _run({"LRH_FUSE_FFT1": "1"}, sd="1", sparse=1, more_strong=200, blanker=False)
    b = _run({"LRH_FUSE_FFT1": "1"}, sd="0", sparse=1, more_strong=200, blanker=False)
    assert a["p"] == b["p"] and np.count_nonzero(a["lim"]) > 200
    for _, k in RINGS:
        assert np.array_equal(a[k], b[k]), k
    c = _run({"LRH_FUSE_FFT1": "1"}, sd="1", sparse=1, more_strong=60, blanker=False)      # ... and 60 more bins (about 110 in all) still go the direct way
    d = _run({"LRH_FUSE_FFT1": "1"}, sd="0", sparse=1, more_strong=60, blanker=False)
    tc, td = c["timf2"].reshape(-1, 4), d["timf2"].reshape(-1, 4)
    assert np.array_equal(tc[:, :2], td[:, :2]) and not np.array_equal(tc[:, 2:], td[:, 2:]) and _rel(tc[:, 2:], td[:, 2:]) < 1e-6
