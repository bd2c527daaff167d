"""GPU, TWO OR MORE DEVICES (skipped on a one-GPU lease; lights up on a multi-GPU node): one process per GPU, backend nccl (= RCCL over
xGMI), one HIP context per device -- the sharding of DESIGN 6 on real hardware.

  * the coupled polarisation pair (BASELINE configs[3]): rank r carries RF channel r on device r, ONE lrh_wideband_dsp call per rank with
    the cross-channel collectives issued from inside the library (lrh_set_exchange -> linrad_amd.multichan.install_exchange: all-reduce of
    the blanker's power sums and noise statistics, blank1.c:1236-1300; all-gather of the fft2 bins for the cross products, fft2.c:1622-1640;
    all-reduce of the polarisation sums, mix2.c:377-380), against the goldens of the compiled two-channel reference (twochan_n10_chain);
  * the N-channel coherent combine (BASELINE configs[4]): every rank's beam equals the weighted sum of N single-channel oracle contexts.

On one GPU the same code paths run with a one-rank group (tests/test_gpu_coupled.py), with two ranks over gloo on one device
(tests/test_multichan_combine.py::test_bench_two_ranks_on_one_gpu_over_gloo) and with two / eight contexts on one device
(tests/test_twochan.py, tests/test_multichan_combine.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _devices():
    import torch
    return torch.cuda.device_count()            # (does not initialise the GPU in this process)


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _spawn(target, world, *args):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _rank_setup(rank, world, port, rehearse):
    """rehearse: the same ranks on ONE GPU (device 0 for all, backend gloo on the library's device buffers) -- everything of these tests
    but RCCL itself, so that they are not first executed on the day a multi-GPU node shows up"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    index = 0 if rehearse else rank
    torch.cuda.set_device(index)
    dist.init_process_group("gloo" if rehearse else "nccl", rank=rank, world_size=world)
    assert dist.get_world_size() == world
    return torch, dist, torch.device(f"cuda:{index}"), index      # (gloo moves device tensors through host memory by itself)


def _pair_worker(rank, world, port, q, name, rehearse):
    torch, dist, dev, index = _rank_setup(rank, world, port, rehearse)
    from linrad_amd import abi
    from linrad_amd.lib import open_hip
    from linrad_amd.multichan import install_exchange
    from refcases import lrh_config, twochan_case
    d, frames, lim = twochan_case(name, chain=True)
    g = np.load(os.path.join(ROOT, "tests", "golden", f"{name}_chain.npz"))
    iq = np.ascontiguousarray(frames.reshape(-1, 4)[:, 2 * rank:2 * rank + 2]).ravel()
    cfg = lrh_config(d, iq, blanker_channels=2, timf1_channel_index=rank, device=index)
    cfg.timf1_bytes *= 2
    cfg.timf1_frame_channels = 2                              # Linrad's interleaved frames {I0, Q0, I1, Q1}: every rank reads its own channel
    rx = open_hip(cfg)
    rx.timf1_write(frames)
    rx.set_liminfo(lim)
    rx.set_mix1_selfreq(d["fq"])
    rx.set_bg_filterfunc(g["bg_filterfunc"])
    rx.set_pol(*d["pol"])
    if rank == 1:
        rx.set_ch2_phasing(d["ch2_c1"], d["ch2_c2"])
    install_exchange(rx, dist, dev)
    rx.wideband_dsp(d["nblk"], 1)
    out = dict(fft3=rx.export(abi.RING_FFT3), baseb=rx.export(abi.RING_BASEB_RAW), fft2=rx.export(abi.RING_FFT2_FLOAT), xyp=rx.export(abi.RING_FFT2_XYPOWER),
               xys=rx.export(abi.RING_FFT2_XYSUM), timf3=rx.export(abi.RING_TIMF3_FLOAT), p=rx.p.as_dict(), wf=rx.export(abi.RING_WG_WATERF))
    rx.close()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


@pytest.mark.parametrize("name,rehearse", [("twochan_n10", False), ("twochan_n9_sin3", False), ("twochan_n10", True)])
def test_polarisation_pair_on_two_gpus_matches_the_two_channel_reference(name, rehearse):
    if _devices() < 2 and not rehearse:
        pytest.skip("needs two GPUs (one RF channel per GPU)")
    from refcases import twochan_case
    from test_twochan import _check_chain
    res = _spawn(_pair_worker, 2, name, rehearse)
    d, _, _ = twochan_case(name, chain=True)
    g = np.load(os.path.join(ROOT, "tests", "golden", f"{name}_chain.npz"))
    _check_chain(d, g, [res[0], res[1]], None, 0, 1e-5)
    assert np.array_equal(res[0]["wf"], res[1]["wf"]) and np.any(res[0]["wf"])     # the polarisation-independent waterfall: the same lines on both ranks


def _combine_worker(rank, world, port, q, rehearse):
    torch, dist, dev, index = _rank_setup(rank, world, port, rehearse)
    from linrad_amd import abi
    from linrad_amd.lib import open_hip
    from linrad_amd.multichan import coupled_fft3_mix2
    from refcases import case_params
    import test_multichan_combine as t
    d = case_params("n10_n12_fft3")
    d["nblk"] = 48
    w = (np.exp(1j * t.PHASE[rank]) / world, np.exp(1j * t.PHASE[rank] + 2j * np.pi * rank / world) / world)
    rx = t._open(lambda cfg: open_hip(_on_device(cfg, index)), d, rank, w)
    for _ in range(d["nblk"] // 4):
        rx.wideband_dsp(4, 1)
        k3 = rx.fft3_available()
        while k3 > 0:
            k3b = min(k3, max(1, rx.cfg.max_fft3n // 2))
            rx.make_fft3_all(k3b)
            coupled_fft3_mix2(rx, k3b, dist, dev)            # all-reduce in place on the library's device buffer, on its own stream
            k3 -= k3b
    out = rx.export(abi.RING_BASEB_RAW)
    rx.close()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


def _on_device(cfg, dev):
    cfg.device = dev
    return cfg


@pytest.mark.parametrize("rehearse", [False, True])
def test_coherent_combine_over_rccl_one_channel_per_gpu(rehearse):
    n = _devices()
    if n < 2 and not rehearse:
        pytest.skip("needs two or more GPUs (one RF channel per GPU)")
    world = 3 if rehearse else min(n, 8)
    import test_multichan_combine as t
    from linrad_amd import abi
    from oracle_binding import open_oracle
    from refcases import case_params
    res = _spawn(_combine_worker, world, rehearse)
    for r in range(1, world):
        assert np.array_equal(res[0], res[r])                 # every rank ends with the same beam
    d = case_params("n10_n12_fft3")
    d["nblk"] = 48
    single = [t._open(open_oracle, d, ch) for ch in range(world)]
    for rx in single:
        rx.wideband_dsp(d["nblk"], 1)
    want = sum(np.exp(1j * t.PHASE[ch]) / world * rx.export(abi.RING_BASEB_RAW).astype(np.float64).view(np.complex128) for ch, rx in enumerate(single))
    got = res[0].astype(np.float64).view(np.complex128)
    assert np.count_nonzero(want) > 100 and np.linalg.norm(got - want) <= 1e-5 * np.linalg.norm(want)


def test_bench_line_on_two_gpus_reports_the_group_it_ran_in():
    """bench.py --gpus 2 the way the driver starts it (torch.distributed.run, backend nccl): n_gpus, collective_world_size read back from
    the process group, the coupled pair as the secondary object"""
    if _devices() < 2:
        pytest.skip("needs two GPUs")
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", LRH_BENCH_WATCHDOG="400")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2500:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and out["config"]["collective_world_size"] == 2 and out["config"]["backend"] == "nccl" and out["value"] > 0
