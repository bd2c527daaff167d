#!/usr/bin/env python3
"""bench.py -- Msamples/s of complex IQ through fft1 -> timf2(+blank1) -> fft2 -> mix1 on MI355X, % of HBM roofline,
with the CPU oracle timed beside it (BASELINE.json metric; contract in the task description).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--fft2-n 12]

One process per GPU (torch.distributed.run for N > 1): every rank runs its own RF channel (weak scaling, no data-path
collective); the only exchange is the all-reduce of the per-bin channel power sums (fft1.c:4138 semantics).
A "step" = lrh_wideband_dsp over one batch of B fft1 blocks of synthetic int16 IQ already resident in the device ring.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from linrad_amd import abi  # noqa: E402
from linrad_amd.multichan import channel_of_rank, cross_channel_power_sum, newest_sumsq_block  # noqa: E402
from linrad_amd.workload import (ALG_BYTES, ALG_BYTES_CHAIN, HBM_PEAK_GBS, chain_config, strong_liminfo)  # noqa: E402


def setup_receiver(cfg, channel, open_fn, synth_mod):
    N1 = 1 << cfg.fft1_n
    rx = open_fn(cfg)
    s = synth_mod.synth_defaults(N1, channel)
    nring = cfg.timf1_bytes // 4
    rx.timf1_write(synth_mod.synth_iq(s, 0, nring))
    rx.set_liminfo(strong_liminfo(s, cfg.fft1_n))
    rx.set_mix1_selfreq(0.31 * (1 << cfg.fft2_n) + 0.3)
    return rx


def measured_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r01_traffic.json: FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this very command, FETCH doubled as the gfx950 guide prescribes); None when
    the run's workload is not the profiled one."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        w = t["workload"]
        if (w["fft1_n"], w["fft2_n"], w["batch"]) != (args.fft1_n, args.fft2_n, args.batch):
            return None
        return t["kernels"][kernel]["traffic_bytes_per_launch"]
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(args, fft1_n, fft2_n):
    """The oracle (C restatement of the reference path, -O2 -ffast-math, 1 thread) on a bounded sample of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import open_oracle
    from linrad_amd import lib as hiplib
    nblk = args.cpu_blocks
    cfg = chain_config(fft1_n, fft2_n, batch=min(32, nblk))
    rx = setup_receiver(cfg, 0, open_oracle, hiplib)
    M1 = (1 << fft1_n) // 2
    rx.wideband_dsp(min(32, nblk), cfg.max_batch)          # warm the caches / tables
    t0 = time.perf_counter()
    rx.wideband_dsp(nblk, cfg.max_batch)
    dt = time.perf_counter() - t0
    return {"value": round(nblk * M1 / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"{nblk} fft1 blocks ({nblk * M1} samples) of the same workload, 1 thread, {dt:.1f} s"}


def cpu_reference(args, fft1_n, fft2_n):
    """The COMPILED REFERENCE itself (oracle/_ref/ref_harness: fventuri/linrad's own C files, -O2 -ffast-math, its
    single-CPU call order wcw.c:1036-1118) on a bounded sample of the same workload, one thread.  None when the binary
    is not there (it is built in the build container, where the reference sources are, and travels with the snapshot)."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not os.access(exe, os.X_OK):
        return None
    from linrad_amd import lib as hiplib
    from linrad_amd.workload import level_gain
    N1, N2 = 1 << fft1_n, 1 << fft2_n
    M1 = N1 // 2
    nblk = max(256, args.cpu_blocks // 2)
    s = hiplib.synth_defaults(N1, 0)
    iq = hiplib.synth_iq(s, 0, nblk * M1 + 2 * N1)
    lim = strong_liminfo(s, fft1_n)
    per_block = max(2, 2 * (M1 // max(1, N2 // 2)) + 2)
    pow2 = lambda v: 1 << int(np.ceil(np.log2(v)))  # noqa: E731
    with tempfile.TemporaryDirectory() as td:
        fi, fl, fo = (os.path.join(td, x) for x in ("in.bin", "lim.bin", "out.bin"))
        np.asarray(iq, np.int16).tofile(fi)
        lim.tofile(fl)
        cmd = [exe, f"n1={fft1_n}", f"n2={fft2_n}", "mixred=6", "att_n=6", f"gain={level_gain(fft1_n, 6)}", "avg1num=5", "avg2num=4",
               f"nblk={nblk}", "max_fft1n=8", f"max_fft2n={pow2(per_block)}", "sumsq_blocks=16", "stupid=1", "bln_interval=152",
               "bln_avgnum=1220", f"fq={0.31 * N2 + 0.3}", "wf_avgnum=8", f"wf_pixels={min(N2, 1024)}", "timing=1",
               f"in={fi}", f"liminfo={fl}", f"out={fo}"]
        try:
            out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300, check=True).stdout
            r = json.loads(out.strip().splitlines()[-1])
        except Exception:  # noqa: BLE001
            return None
    dt = r["loop_seconds"]
    return {"value": round(nblk * M1 / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "reference",
            "sample": f"{nblk} fft1 blocks ({nblk * M1} samples) of the same workload through the compiled reference "
                      f"(fft1_b, fft1_c, make_timf2, first_noise_blanker, make_fft2, fft2_mix1_fixed), 1 thread, {dt:.1f} s"}


def cpu_worker(fft1_n, fft2_n, nblk, channel):
    """One single-thread oracle pipeline (child process of cpu_baseline_all_cores); prints its loop time."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import open_oracle
    from linrad_amd import lib as hiplib
    cfg = chain_config(fft1_n, fft2_n, batch=min(32, nblk))
    rx = setup_receiver(cfg, channel, open_oracle, hiplib)
    rx.wideband_dsp(min(32, nblk), cfg.max_batch)
    t0 = time.perf_counter()
    rx.wideband_dsp(nblk, cfg.max_batch)
    print(json.dumps({"seconds": time.perf_counter() - t0}))


def cpu_baseline_all_cores(args, fft1_n, fft2_n):
    """the same oracle on every host core.  The reference spreads one receiver over stage threads
    (wcw.c:604-648); as its throughput on C cores cannot exceed C independent single-thread pipelines, that upper bound
    is what is measured here: C child processes (at most 16), one channel each, started together.  None if a child fails."""
    import subprocess
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ncpu = min(ncpu, 16)      # the reference's own topology ends near a dozen threads (6 fft1 workers + stage threads)
    nblk = max(256, args.cpu_blocks // 8)
    cmd = [sys.executable, os.path.abspath(__file__), "--fft1-n", str(fft1_n), "--fft2-n", str(fft2_n), "--cpu-blocks", str(nblk)]
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd + ["--cpu-worker", str(ch)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for ch in range(ncpu)]
    times = []
    for p_ in procs:
        try:
            out, _ = p_.communicate(timeout=300)
            times.append(json.loads(out.strip().splitlines()[-1])["seconds"])
        except Exception:  # noqa: BLE001
            p_.kill()
    if len(times) != ncpu:
        return None
    M1 = (1 << fft1_n) // 2
    dt = max(times)
    return {"value": round(ncpu * nblk * M1 / dt / 1e6, 4), "unit": "Msamples/s", "cores": ncpu, "kind": "port",
            "sample": f"{ncpu} processes x {nblk} fft1 blocks, one channel each, slowest {dt:.1f} s "
                      f"(wall incl. start-up {time.perf_counter() - t0:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    # batch = fft1 blocks handed to every kernel launch.  Throughput grows with it (fuller waves of workgroups, launch and
    # host overheads amortised: 23.3 Gsamples/s at 1024, 26.9 at 2048, 29.2 at 4096, 29.3 at 8192, round 1) at the price of
    # batch*8192 samples of latency; 4096 blocks are 1.1 ms of signal at the rate the chain sustains.
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=4, help="batches of --batch fft1 blocks per step (pipelined on two streams)")
    ap.add_argument("--fft1-n", type=int, default=14)
    ap.add_argument("--fft2-n", type=int, default=12)
    ap.add_argument("--cpu-blocks", type=int, default=16384)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--stream-host", action="store_true",
                    help="PCIe-inclusive variant (never the headline value): every step first hands its samples over from "
                         "page-locked host memory with lrh_timf1_write_async, overlapped with the previous step's kernels")
    ap.add_argument("--coupled", action="store_true",
                    help="BASELINE configs[3]: ranks 0/1 are the two channels of a polarisation pair (ui.rx_rf_channels = 2 sharded): "
                         "coupled blanker (2 all-reduces per call), fft2 cross products (all-gather), fft3 + polarisation transform "
                         "in mix2 (all-reduce), stage calls driven from linrad_amd.multichan.run_coupled; not the headline metric")
    ap.add_argument("--real-input", action="store_true",
                    help="real samples (fft1 version 2): every fft1 block takes 2*M1 reals; value still counts M1 complex-rate samples per block")
    ap.add_argument("--cpu-worker", type=int, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker is not None:                        # child of cpu_baseline_all_cores: no GPU, no torch
        cpu_worker(args.fft1_n, args.fft2_n, args.cpu_blocks, args.cpu_worker)
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    force_dist = os.environ.get("LRH_BENCH_FORCE_DIST") == "1"      # exercise the RCCL path on a single GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        # lazy communicator: an RCCL communicator on the device costs this pipeline ~6 % (measured, round 1) even when
        # idle, so it is only created by the first collective, i.e. when there really is more than one rank
        dist.init_process_group("nccl")
    from linrad_amd import lib as hiplib

    cfg = chain_config(args.fft1_n, args.fft2_n, batch=args.batch, device=local_rank)
    if args.real_input:
        cfg.timf1_real_input = 1
    if args.coupled:
        if world > 2:
            raise SystemExit("--coupled: a polarisation pair has two channels (Linrad's maximum, SURVEY F4)")
        cfg.blanker_channels, cfg.timf1_channel_index = 2, rank
        cfg.fft3_n, cfg.fft3_sinpow, cfg.mix2_n, cfg.max_fft3n, cfg.baseband_size = 10, 2, 8, 64, 1 << 16
        cfg.timf3_size = max(cfg.timf3_size, 1 << 16)
    N1, N2, M1 = 1 << args.fft1_n, 1 << args.fft2_n, (1 << args.fft1_n) // 2
    rx = setup_receiver(cfg, channel_of_rank(rank), hiplib.open_hip, hiplib)
    samples_per_step = args.batch * args.rounds * M1
    use_dist = dist is not None
    xchg = torch.zeros(N1, dtype=torch.float32, device=f"cuda:{local_rank}") if use_dist else None

    # the context's own stream, wrapped so that torch / RCCL work can be ordered against it without host waits
    # (and a side stream for the collective: the legacy default stream would serialise with every blocking stream)
    lrh_stream = torch.cuda.ExternalStream(rx.stream_handle(), device=torch.device("cuda", local_rank)) if use_dist else None
    comm_stream = torch.cuda.Stream(device=torch.device("cuda", local_rank)) if use_dist else None

    host_ring = None
    if args.stream_host:
        # the producer's side of the boundary: Linrad's timf1 arena, page-locked once; one step's worth of
        # new samples is 4 bytes per complex sample
        s0 = hiplib.synth_defaults(N1, channel_of_rank(rank))
        host_ring = np.ascontiguousarray(hiplib.synth_iq(s0, 0, cfg.timf1_bytes // 4))
        rx.host_register(host_ring)
        if args.batch * args.rounds * M1 * 4 > cfg.timf1_bytes // 2:
            raise SystemExit("--stream-host: one step's samples must fit half the timf1 ring (use --rounds 1)")
    step_bytes = args.batch * args.rounds * M1 * 4
    wr = [0]

    if args.coupled:
        from linrad_amd.multichan import run_coupled
        if dist is None:                                    # a one-rank group: the collectives are there, nothing to fetch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=0, world_size=1)
            use_dist = True
        rx.set_pol(0.8, 0.36, -0.48)
        dev = torch.device("cuda", local_rank)

    def step():
        if args.coupled:
            run_coupled(rx, args.batch * args.rounds, args.batch, dist, device=dev, xy=True, pol=True)
            return
        if host_ring is not None:
            nb = min(step_bytes, cfg.timf1_bytes)
            off = wr[0] % cfg.timf1_bytes
            first = min(nb, cfg.timf1_bytes - off)
            flat = host_ring.view(np.uint8)
            rx.timf1_write_async(flat[off:off + first], off)
            if nb > first:
                rx.timf1_write_async(flat[:nb - first], 0)
            wr[0] += nb
        rx.wideband_dsp(args.batch * args.rounds, args.batch)
        if use_dist:
            # cross-channel power sum of the newest averaged spectrum (fft1.c:4138: sum over channels per bin)
            lrh_stream.wait_stream(comm_stream)             # the previous all-reduce is done with xchg
            rx.export_device_async(abi.RING_FFT1_SUMSQ, xchg.data_ptr(), newest_sumsq_block(rx), N1)
            comm_stream.wait_stream(lrh_stream)
            with torch.cuda.stream(comm_stream):
                cross_channel_power_sum(xchg, dist)

    def barrier():
        if host_ring is not None:
            rx.timf1_write_wait()
        rx.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    rx.timer_start()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0                       # host-side enqueue time (the device runs behind it)
    ev_ms = rx.timer_stop()
    barrier()
    host_ph = rx.profile_get("host:mix1_phases")
    host_dsp = rx.profile_get("host:wideband_dsp")
    host_dsp_cpu = rx.profile_get("host:wideband_dsp_cpu")
    host_wait = rx.profile_get("host:staging_wait")
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_samples = world * args.steps * samples_per_step
    value = total_samples / dt / 1e6

    # ---- per-kernel timing with HIP events on the context stream (rank 0), same steps
    roof = None
    stages = {}
    if rank == 0 and not args.coupled:
        rx.profile_enable(True)
        rx.wideband_dsp(args.batch, args.batch)                # first serial pass after the two-stream run: discarded
        rx.sync()
        rx.profile_enable(True)                                # resets the accumulators
        for _ in range(max(3, min(args.steps, 10))):
            rx.wideband_dsp(args.batch, args.batch)
        rx.sync()
        for k in ("fft1", "sumsq", "sumsq_join", "slowsum", "timf2", "blanker", "fft2", "powersum2", "waterfall", "mix1"):
            ms, n = rx.profile_get(k)
            if n:
                stages[k] = {"ms_total": round(ms, 4), "launches": n, "avg_us": round(1e3 * ms / n, 2)}
                if k in ALG_BYTES:                         # stand-alone stage rates (secondary metric): one batch per launch
                    per = ALG_BYTES[k] + (ALG_BYTES["sumsq"] if k == "timf2" else 0.0)
                    stages[k]["Msamples_per_s"] = round(args.batch * M1 / (ms / n * 1e-3) / 1e6, 1)
                    stages[k]["alg_GBps"] = round(stages[k]["Msamples_per_s"] * per / 1e3, 1)
        rx.profile_enable(False)
        dom = max((k for k in stages if k in ALG_BYTES), key=lambda k: stages[k]["ms_total"])
        nsteps_prof = stages["fft1"]["launches"]
        launches_per_step = stages[dom]["launches"] / nsteps_prof
        per_sample = ALG_BYTES[dom]
        if dom == "timf2" and "sumsq" not in stages:           # fft1_c's power sums are computed inside k_timf2: both stages' bytes
            per_sample += ALG_BYTES["sumsq"]
        alg_bytes_launch = per_sample * (args.batch * M1) / launches_per_step      # the profiling loop runs single batches
        avg_s = stages[dom]["ms_total"] / stages[dom]["launches"] / 1e3
        achieved = alg_bytes_launch / avg_s / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(dom, args),
                "alg_bytes_per_launch": int(alg_bytes_launch), "avg_launch_us": round(avg_s * 1e6, 2),
                "chain_alg_GBps": round(value * ALG_BYTES_CHAIN / 1e3, 1),
                "chain_frac": round(value * ALG_BYTES_CHAIN / 1e3 / HBM_PEAK_GBS / world, 4)}
    if rank == 0 and roof is not None:
        # context for the 8 TB/s nominal peak: what a plain device-to-device copy reaches on this box (read + write bytes)
        try:
            nbytes = 1 << 30
            src = torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{local_rank}")
            dst = torch.empty_like(src)
            dst.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            roof["device_copy_GBps"] = round(10 * 2 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            del src, dst
        except Exception:  # noqa: BLE001
            roof["device_copy_GBps"] = None
    cpu = cpu_all = cpu_port = None
    if rank == 0 and not args.no_cpu:
        cpu_port = cpu_baseline(args, args.fft1_n, args.fft2_n)
        cpu = cpu_reference(args, args.fft1_n, args.fft2_n) or cpu_port
        cpu_all = cpu_baseline_all_cores(args, args.fft1_n, args.fft2_n)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        bs = rx.blanker_state()
        out = {
            "metric": "Msamples/s complex IQ through fft1->timf2->fft2->mix1; % HBM roofline",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: 1 channel/GPU complex-int16 IQ, fft1_size={N1} sin^2 50% overlap, "
                                   f"timf2 + stupid blanker, fft2_size={N2} sin^2, mix1 size {N2 >> 6}; "
                                   f"{args.rounds} x {args.batch} fft1 blocks ({samples_per_step} samples) per step, device-resident ring",
                       "fft1_size": N1, "fft2_size": N2, "batch_blocks": args.batch, "rounds_per_step": args.rounds, "channels": world,
                       "parallelism": f"1 RF channel per GPU x{world}"},
            "mode": "two coupled channels (polarisation pair), stage calls + collectives from linrad_amd.multichan" if args.coupled else "lrh_wideband_dsp",
            "input": "page-locked host ring over PCIe, lrh_timf1_write_async per step" if args.stream_host else "device-resident ring",
            "event_ms_per_step": round(ev_ms / args.steps, 4), "host_enqueue_ms_per_step": round(1e3 * t_enq / args.steps, 4),
            "host_cpu": {"mix1_phase_ms_per_call": round(host_ph[0] / max(host_ph[1], 1), 4), "wideband_dsp_ms_per_call": round(host_dsp[0] / max(host_dsp[1], 1), 4),
                         "wideband_dsp_cpu_ms_per_call": round(host_dsp_cpu[0] / max(host_dsp_cpu[1], 1), 4),
                         "staging_wait_ms_per_call": round(host_wait[0] / max(host_wait[1], 1), 4)},
            "realtime_factor": {k: round(value / world * 1e6 / r, 1) for k, r in (("10Msps", 10e6), ("40Msps", 40e6), ("160Msps", 160e6))},
            "roofline": roof, "cpu_baseline": cpu, "cpu_baseline_port": cpu_port, "cpu_baseline_all_cores": cpu_all, "stages": stages,
            "blanker": {"noise_floor": bs.timf2_noise_floor, "limit": bs.stupid_bln_limit,
                        "cleared_rate_pct": round(bs.stupid_blanker_rate, 3), "slow_path_calls": bs.slow_path_calls},
        }
        print(json.dumps(out))


if __name__ == "__main__":
    main()
