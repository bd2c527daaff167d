"""Coherent combine of more than two receivers (BASELINE configs[4], phased array: one channel per GPU, A = sum_c w_c X_c over
the mix2 band as one all-reduce per fft3_mix2 batch).  The reference stops at two channels, so there is nothing to
pin against: the checks are the properties the step must have -- linearity (the combined baseband equals the weighted sum of
the single-channel basebands) and the array gain of phase-aligned weights."""
import os
import socket
import sys

import numpy as np
import pytest

from linrad_amd import abi
from refcases import case_params, lrh_config, make_input, make_liminfo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NCH = 4
PHASE = [0.0, 0.9, -1.7, 2.4, 0.35, -2.6, 1.55, -0.8]       # sky phase per channel (up to the 8 of BASELINE configs[4])


def _channel_input(d, ch):
    """the case's signal turned by the channel's sky phase, with the channel's own noise"""
    iq = make_input(d).astype(np.float64)
    z = iq[0::2] + 1j * iq[1::2]
    rng0 = np.random.default_rng(d["seed"])
    n = z.size
    noise0 = rng0.normal(0, d["sigma"], n) + 1j * rng0.normal(0, d["sigma"], n)
    rng = np.random.default_rng(1000 + ch)
    zc = (z - noise0) * np.exp(1j * PHASE[ch]) + rng.normal(0, d["sigma"], n) + 1j * rng.normal(0, d["sigma"], n)
    out = np.empty(2 * n, np.int16)
    out[0::2], out[1::2] = np.clip(np.round(zc.real), -32767, 32767), np.clip(np.round(zc.imag), -32767, 32767)
    return out


def _open(open_fn, d, ch, weights=None):
    iq = _channel_input(d, ch)
    rx = open_fn(lrh_config(d, iq, stupid_bln_mode=0))
    rx.timf1_write(iq)
    rx.set_liminfo(make_liminfo(d))
    rx.set_mix1_selfreq(d["fq"])
    n3 = 1 << d["fft3_n"]
    rx.set_bg_filterfunc(np.exp(-((np.arange(n3) - n3 / 2) / (n3 / 6.0)) ** 2).astype(np.float32))
    if weights is not None:
        rx.set_combine_weights(*weights)
    return rx


def _drive(rxs, d, combine):
    for _ in range(d["nblk"]):
        for rx in rxs:
            rx.fft1_b(1), rx.fft1_c(1), rx.make_timf2(1), rx.first_noise_blanker()
        k = rxs[0].fft2_available()
        for _ in range(k):
            for rx in rxs:
                rx.make_fft2(1), rx.fft2_mix1_fixed(1)
            k3 = rxs[0].fft3_available()
            if not k3:
                continue
            for rx in rxs:
                rx.make_fft3_all(k3)
            if combine:
                cnt = [rx.mix2_pol_begin(k3) for rx in rxs]
                tot = sum(rx.exchange_read(rx.X_POL, cnt[0]) for rx in rxs)        # the all-reduce, by hand
                for rx in rxs:
                    rx.exchange_write(rx.X_POL, tot)
            for rx in rxs:
                rx.fft3_mix2(k3)


def _check(open_fn, tol, NCH=NCH):
    d = case_params("n10_n12_fft3")
    d["nblk"] = 72
    # the chain conjugates its input: a sky phase +p arrives in the baseband as -p, so e^{+jp} aligns it
    w = [np.exp(1j * p) / NCH for p in PHASE[:NCH]]                 # phase-aligning weights: the array's beam on the source
    w2 = [np.exp(1j * p + 2j * np.pi * c / NCH) / NCH for c, p in enumerate(PHASE[:NCH])]    # second beam: a null on the source
    single = [_open(open_fn, d, ch) for ch in range(NCH)]
    _drive(single, d, combine=False)
    base = [rx.export(abi.RING_BASEB_RAW).astype(np.float64).view(np.complex128) for rx in single]
    comb = [_open(open_fn, d, ch, (w[ch], w2[ch])) for ch in range(NCH)]
    _drive(comb, d, combine=True)
    got = [rx.export(abi.RING_BASEB_RAW).astype(np.float64).view(np.complex128) for rx in comb]
    want = sum(wc * b for wc, b in zip(w, base))
    assert np.count_nonzero(want) > 200
    for g in got:                                                   # every context carries the same sum A
        assert np.linalg.norm(g - want) <= tol * np.linalg.norm(want)
    # array gain: the aligned sum keeps the source and averages the noise down; a single channel scaled alike does not
    used = np.abs(want) > 0
    assert np.mean(np.abs(want[used]) ** 2) > 0.5 * np.mean(np.abs(base[0][used]) ** 2)
    null = sum(wc * b for wc, b in zip(w2, base))
    assert np.mean(np.abs(null[used]) ** 2) < 0.5 * np.mean(np.abs(want[used]) ** 2)


def _combined(open_fn, NCH):
    """baseb_raw of every context of an NCH-channel array after the coherent combine (two beams), the all-reduce by hand"""
    d = case_params("n10_n12_fft3")
    d["nblk"] = 72
    w = [np.exp(1j * p) / NCH for p in PHASE[:NCH]]
    w2 = [np.exp(1j * p + 2j * np.pi * c / NCH) / NCH for c, p in enumerate(PHASE[:NCH])]
    comb = [_open(open_fn, d, ch, (w[ch], w2[ch])) for ch in range(NCH)]
    _drive(comb, d, combine=True)
    out = [rx.export(abi.RING_BASEB_RAW).astype(np.float64).view(np.complex128) for rx in comb]
    for rx in comb:
        rx.close()
    return out


@pytest.mark.gpu
def test_hip_eight_channel_combine_matches_the_oracle():
    """BASELINE configs[4] held to the oracle, not to itself: eight HIP contexts (one per channel, the all-reduce of LRH_X_POL by
    hand) against eight ORACLE contexts driven the same way -- the combined beam within the north-star's 1e-5.  (The reference
    stops at two channels, so the oracle's combine is pinned by its two-channel case, tests/test_twochan.py, and by linearity.)"""
    from linrad_amd.lib import open_hip
    from oracle_binding import open_oracle
    ref = _combined(open_oracle, 8)
    got = _combined(open_hip, 8)
    assert np.count_nonzero(ref[0]) > 200
    err = max(np.linalg.norm(g - r) / np.linalg.norm(r) for g, r in zip(got, ref))
    print("eight-channel combine, HIP vs oracle: relative RMS error", err)
    assert err <= 1e-5, err


def _fullsize_array(open_fn, hiplib, nch, nblocks=256, batch=32):
    """the bench's configs[4] workload (fft1 16384 -> fft2 65536 -> mix1 1024 -> fft3 4096 -> mix2 256, SURVEY 8d signal with sky
    phase 0.7 c on channel c, beam weights of bench.py) through lrh_wideband_dsp per channel + the combine step"""
    from linrad_amd.workload import chain_config, strong_liminfo
    cfg = chain_config(14, 16, batch=batch, fft3_n=12, mix2_n=8, rounds=nblocks // batch)
    cfg.stupid_bln_mode = 0           # like the small case: a blanker decision within float32 rounding of the limit may differ between HIP and
    rxs = []                          # oracle (tests/test_gpu_fullsize.py deals with those); the combine is what is compared here
    for ch in range(nch):
        rx = open_fn(cfg)
        s = hiplib.synth_defaults(1 << cfg.fft1_n, ch)
        rx.timf1_write(hiplib.synth_iq(s, 0, cfg.timf1_bytes // 4))
        rx.set_liminfo(strong_liminfo(s, cfg.fft1_n))
        rx.set_mix1_selfreq(0.31 * (1 << cfg.fft2_n) + 0.3)
        th = 0.7 * ch
        rx.set_combine_weights(np.exp(-1j * th) / nch, np.exp(-1j * (th + np.pi * ch / nch)) / nch)
        rxs.append(rx)
    for _ in range(nblocks // batch):
        for rx in rxs:
            rx.wideband_dsp(batch, batch)
        k3 = rxs[0].fft3_available()
        while k3 > 0:
            k3b = min(k3, max(1, cfg.max_fft3n // 2))
            for rx in rxs:
                rx.make_fft3_all(k3b)
            cnt = [rx.mix2_pol_begin(k3b) for rx in rxs]
            tot = sum(rx.exchange_read(rx.X_POL, cnt[0]) for rx in rxs)
            for rx in rxs:
                rx.exchange_write(rx.X_POL, tot)
                rx.fft3_mix2(k3b)
            k3 -= k3b
    out = [rx.export(abi.RING_BASEB_RAW).astype(np.float64).view(np.complex128) for rx in rxs]
    fft2 = rxs[0].export(abi.RING_FFT2_FLOAT).astype(np.float64)
    for rx in rxs:
        rx.close()
    return out, fft2, cfg


@pytest.mark.gpu
def test_hip_eight_channel_fullsize_combine_matches_the_oracle():
    """the same at BASELINE's full sizes (fft1 16384, fft2 65536): eight HIP contexts against eight oracle contexts on the bench's
    own signal; every context ends with the same beam.  The baseband is a 256-bin cut out of a 65536-bin float32 spectrum that
    carries an 8000-LSB carrier, so besides the relative gate the absolute float32 floor of that spectrum applies (paritylib)."""
    from linrad_amd import lib as hiplib
    from oracle_binding import open_oracle
    got, _, cfg = _fullsize_array(hiplib.open_hip, hiplib, 8)
    ref, fft2, _ = _fullsize_array(open_oracle, hiplib, 8)
    assert np.count_nonzero(ref[0]) > 500
    for g in got[1:]:
        assert np.array_equal(g, got[0])
    err = np.linalg.norm(got[0] - ref[0])
    rel = err / np.linalg.norm(ref[0])
    n2 = 1 << cfg.fft2_n
    wide = np.linalg.norm(fft2) / np.sqrt(cfg.max_fft2n)
    floor = 4 * 6e-8 * wide * np.sqrt(256.0 / n2) * np.sqrt(ref[0].size / 256.0)
    print("full-size eight-channel combine, HIP vs oracle: relative", rel, "absolute", err, "float32 floor", floor)
    assert rel <= 1e-5 or err <= floor, (rel, err, floor)


@pytest.mark.gpu
def test_bench_combine_step_runs_over_rccl_on_one_gpu():
    """pre-flight of the multi-GPU run: bench.py's configs[4] mode (coherent combine: ExternalStream + in-place all-reduce on the
    library's device buffers, the per-bin power-sum all-reduce) with backend nccl (= RCCL) and a one-rank group on this GPU, so that
    the first 8-GPU run cannot die in plumbing; rank 0's line states the group size the collectives saw"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(LRH_BENCH_FORCE_DIST="1", LRH_BENCH_BACKEND="nccl", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--combine", "--steps", "2", "--warmup", "1", "--batch", "256", "--rounds", "2",
                        "--no-cpu", "--no-secondary"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["config"]["collective_world_size"] == 1 and out["config"]["backend"] == "nccl" and out["n_gpus"] == 1
    assert "coherent combine" in out["mode"] and out["value"] > 100


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [["--coupled", "--no-secondary"], []])
def test_bench_two_ranks_on_one_gpu_over_gloo(flags):
    """the N = 2 line of the scaling run, rehearsed on one GPU: two ranks under torch.distributed.run (backend gloo, both on device 0)
    through bench.py's coupled pair (configs[3]: the collectives are issued inside lrh_wideband_dsp, so BOTH ranks must make the
    profiled calls of the stage-timing pass too -- rank 0 alone waited for ever) and through the default --gpus 2 run (configs[4] with
    configs[3] as its secondary object); a hang ends in the watchdog's stack dump, not in the driver's timeout"""
    import json
    import socket
    import subprocess
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(LRH_BENCH_BACKEND="gloo", LRH_BENCH_SAME_DEVICE="1", LRH_BENCH_WATCHDOG="240", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64", "--rounds", "2", "--no-cpu", *flags],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stderr[-2500:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and out["config"]["collective_world_size"] == 2 and out["value"] > 0
    if flags:
        assert "coupled" in out["mode"] and out["stage_us"]
    else:
        assert "coherent combine" in out["mode"] and out["secondary"].get("value", 0) > 0, out["secondary"]


def test_oracle_four_channel_coherent_combine():
    from oracle_binding import open_oracle
    _check(open_oracle, 2e-6)


def test_oracle_eight_channel_coherent_combine():
    """BASELINE configs[4] has eight receivers"""
    from oracle_binding import open_oracle
    _check(open_oracle, 2e-6, 8)


@pytest.mark.gpu
def test_hip_four_channel_coherent_combine():
    from linrad_amd.lib import open_hip
    _check(open_hip, 2e-5)


@pytest.mark.gpu
def test_hip_eight_channel_coherent_combine():
    """eight contexts on one GPU, the all-reduce by hand: BASELINE configs[4]'s channel count through the HIP path"""
    from linrad_amd.lib import open_hip
    _check(open_hip, 2e-5, 8)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from linrad_amd.multichan import coupled_fft3_mix2
    from oracle_binding import open_oracle
    import test_multichan_combine as t
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = case_params("n10_n12_fft3")
    d["nblk"] = 48
    rx = t._open(open_oracle, d, rank, (np.exp(1j * PHASE[rank]) / world, 0j))
    for _ in range(d["nblk"]):
        rx.fft1_b(1), rx.fft1_c(1), rx.make_timf2(1), rx.first_noise_blanker()
        for _ in range(rx.fft2_available()):
            rx.make_fft2(1), rx.fft2_mix1_fixed(1)
            k3 = rx.fft3_available()
            if k3:
                rx.make_fft3_all(k3)
                coupled_fft3_mix2(rx, k3, dist)
    out = rx.export(abi.RING_BASEB_RAW)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


def test_three_rank_combine_over_gloo():
    """world_size 3: the all-reduce of multichan.coupled_fft3_mix2 is not tied to two ranks; every rank ends with the same sum"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.count_nonzero(res[0]) > 100
    assert np.array_equal(res[0], res[1]) and np.array_equal(res[0], res[2])


def _worker8(rank, world, port, q):
    """one rank of the bench's configs[4] mode: lrh_wideband_dsp-style driving (the oracle's wideband_dsp leaves the narrowband
    side to the caller once combine weights are set), then fft3 / all-reduce of the weighted mix2 bins / mix2"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from linrad_amd.multichan import coupled_fft3_mix2
    from oracle_binding import open_oracle
    import test_multichan_combine as t
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = case_params("n10_n12_fft3")
    d["nblk"] = 48
    rx = t._open(open_oracle, d, rank, (np.exp(1j * PHASE[rank]) / world, np.exp(1j * PHASE[rank] + 2j * np.pi * rank / world) / world))
    for _ in range(d["nblk"] // 4):
        rx.wideband_dsp(4, 1)
        k3 = rx.fft3_available()
        while k3 > 0:
            k3b = min(k3, max(1, rx.cfg.max_fft3n // 2))
            rx.make_fft3_all(k3b)
            coupled_fft3_mix2(rx, k3b, dist)
            k3 -= k3b
    out = rx.export(abi.RING_BASEB_RAW)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


def test_eight_rank_combine_over_gloo():
    """world_size 8 (BASELINE configs[4]): every rank ends with the same beam, and it is the weighted sum of the eight
    single-channel basebands"""
    import torch.multiprocessing as mp
    from oracle_binding import open_oracle
    world = 8
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=400) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.count_nonzero(res[0]) > 100
    for r in range(1, world):
        assert np.array_equal(res[0], res[r])
    d = case_params("n10_n12_fft3")
    d["nblk"] = 48
    single = [_open(open_oracle, d, ch) for ch in range(world)]
    for rx in single:
        rx.wideband_dsp(d["nblk"], 1)                          # no weights set: the narrowband side runs inside
    want = sum(np.exp(1j * PHASE[ch]) / world * rx.export(abi.RING_BASEB_RAW).astype(np.float64).view(np.complex128)
               for ch, rx in enumerate(single))
    got = res[0].astype(np.float64).view(np.complex128)
    assert np.linalg.norm(got - want) <= 2e-6 * np.linalg.norm(want)


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` by hand starts two rank processes itself (the flag was parsed and ignored); without
    GPUs they fail, and the launcher says so and exits non-zero instead of reporting a one-rank run"""
    import subprocess
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present: the run would succeed")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert r.returncode != 0
    import re
    # whichever rank fails first is reported; its peer is terminated by the launcher
    assert "started 2 ranks" in r.stderr and re.search(r"rank [01] \(pid \d+\) exited with", r.stderr), r.stderr[-800:]
    assert '"n_gpus"' not in r.stdout
