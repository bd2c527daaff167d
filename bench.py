#!/usr/bin/env python3
"""bench.py -- Msamples/s of complex IQ through fft1 -> timf2(+blank1) -> fft2 -> mix1 (-> fft3 -> mix2) on MI355X, fraction of
the HBM roofline, with the reference's CPU path timed beside it (BASELINE.json metric; contract in the task description).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--fft2-n 16] [--fft3-n 12]

Default workload (N = 1): BASELINE configs[2] made concrete as SURVEY 8d C3 -- fft1_size 16384, fft2_size 65536, stupid
blanker, mix1 size fft2_size/64, fft3 + mix2 behind it, everything inside the timed region; the lighter configs[1]
(fft2_size 4096, no fft3) is measured in the same run and reported as `secondary`.

N > 1: one process per GPU.  Started by the driver (torch.distributed.run sets RANK / WORLD_SIZE) this process IS a rank; started
by hand as `python bench.py --gpus N` it first spawns the N ranks itself, before anything touches the GPU, and exits with
their status.  Every rank runs its own RF channel through the same chain (weak scaling, BASELINE configs[4]); the exchanges
are the all-reduce of the per-bin channel power sums (fft1.c:4138) and the coherent combine of the channels' baseband bins
(lrh_set_combine_weights + all-reduce between lrh_mix2_pol_begin and lrh_fft3_mix2), both inside the timed region.
`--coupled` (N = 2) runs BASELINE configs[3], the polarisation pair with the coupled blanker and the fft2 cross products.
A "step" = the chain over `rounds` x `batch` fft1 blocks of synthetic int16 IQ already resident in the device ring.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from linrad_amd import abi  # noqa: E402
from linrad_amd.multichan import channel_of_rank, cross_channel_power_sum, newest_sumsq_block  # noqa: E402
from linrad_amd.workload import (ALG_BYTES, HBM_PEAK_GBS, alg_bytes_chain, chain_config, strong_liminfo, workload_name)  # noqa: E402

METRIC = "Msamples/s complex IQ through fft1->timf2->fft2->mix1; % HBM roofline"
STAGES = ("fft1", "fft1w", "timf2s", "spur", "clever", "xypower", "pol", "xcopy", "sumsq", "sumsq_join", "slowsum", "timf2", "blanker", "fft2", "powersum2", "waterfall", "mix1", "fft3", "mix2", "pol", "sellim")


def setup_receiver(cfg, channel, open_fn, synth_mod):
    N1 = 1 << cfg.fft1_n
    rx = open_fn(cfg)
    s = synth_mod.synth_defaults(N1, channel)
    nring = cfg.timf1_bytes // 4
    rx.timf1_write(synth_mod.synth_iq(s, 0, nring))
    rx.set_liminfo(strong_liminfo(s, cfg.fft1_n))
    rx.set_mix1_selfreq(0.31 * (1 << cfg.fft2_n) + 0.3)
    return rx


def source_sha16():
    """hash of the kernel / host sources the library is built from (scripts/summarize_profile.py stores the same in the traffic file)"""
    import hashlib
    h = hashlib.sha256()
    for fn in ("lrh_kernels.hip", "lrh_fft.hip.h", "lrh_kernels.hip.h", "lrh_host.hip", "lrh_timf2_sd.hip"):
        h.update(open(os.path.join(ROOT, "linrad_amd", "csrc", fn), "rb").read())
    return h.hexdigest()[:16]


_TRAFFIC = {}


def measured_traffic(stage, wl):
    """HBM bytes per launch of a stage's kernels from the committed rocprofv3 PMC passes (profiles/r03_traffic.json: FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this very command, FETCH doubled as the gfx950 guide prescribes).  None when this run's
    workload has not been profiled -- or when the file was collected on other sources than the ones this library is built from
    (the byte counts of a kernel that has changed since say nothing about it)."""
    if "t" not in _TRAFFIC:
        _TRAFFIC["t"] = None
        try:
            for fn in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json"):
                f = os.path.join(ROOT, "profiles", fn)
                if not os.path.exists(f):
                    continue
                t = json.load(open(f))
                if t.get("source_sha16") == source_sha16():
                    _TRAFFIC["t"], _TRAFFIC["file"] = t, fn
                    _TRAFFIC["stale"] = False
                    break
                _TRAFFIC["stale"] = True
        except Exception:  # noqa: BLE001
            pass
    t = _TRAFFIC["t"]
    try:
        return t["workloads"][wl]["kernels"][stage] if t else None
    except Exception:  # noqa: BLE001
        return None


# --------------------------------------------------------------------------------------------------------- CPU baselines
def cpu_baseline(args, w):
    """The oracle (C restatement of the reference path, -O2 -ffast-math, 1 thread) on a bounded sample of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import open_oracle
    from linrad_amd import lib as hiplib
    nblk = args.cpu_blocks
    cfg = chain_config(w["fft1_n"], w["fft2_n"], batch=min(32, nblk), fft3_n=w["fft3_n"], mix2_n=w["mix2_n"])
    rx = setup_receiver(cfg, 0, open_oracle, hiplib)
    M1 = (1 << w["fft1_n"]) // 2
    rx.wideband_dsp(min(32, nblk), cfg.max_batch)          # warm the caches / tables
    t0 = time.perf_counter()
    rx.wideband_dsp(nblk, cfg.max_batch)
    dt = time.perf_counter() - t0
    return {"value": round(nblk * M1 / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"{nblk} fft1 blocks ({nblk * M1} samples) of the same workload, 1 thread, {dt:.1f} s"}


def ref_harness_cmd(w, nblk, fi, fl, fo, threads=0):
    from linrad_amd.workload import level_gain
    N2 = 1 << w["fft2_n"]
    M1 = (1 << w["fft1_n"]) // 2
    per_block = max(2, 2 * (M1 // max(1, N2 // 2)) + 2)
    pow2 = lambda v: 1 << int(np.ceil(np.log2(v)))  # noqa: E731
    cmd = [os.path.join(ROOT, "oracle", "_ref", "ref_harness"), f"n1={w['fft1_n']}", f"n2={w['fft2_n']}", "mixred=6", "att_n=6",
           f"gain={level_gain(w['fft1_n'], 6)}", "avg1num=5", "avg2num=4", f"nblk={nblk}", "max_fft1n=8" if not threads else "max_fft1n=64",
           f"max_fft2n={pow2(per_block)}", "sumsq_blocks=16", "stupid=1", "bln_interval=152", "bln_avgnum=1220",
           f"fq={0.31 * N2 + 0.3}", "wf_avgnum=8", f"wf_pixels={min(N2, 1024)}", "timing=1", f"in={fi}", f"liminfo={fl}", f"out={fo}"]
    if w["fft3_n"]:
        cmd += [f"fft3_n={w['fft3_n']}", f"mix2_n={w['mix2_n']}", "max_fft3n=8", "mix2=1"]
    if threads:
        cmd += [f"threads={threads}"]
    return cmd


def cpu_reference(args, w, threads=0):
    """The COMPILED REFERENCE itself (oracle/_ref/ref_harness: fventuri/linrad's own C files, -O2 -ffast-math) on a bounded
    sample of the same workload.  threads = 0: its single-CPU call order (wcw.c:1036-1118), one thread.  threads = T: the
    reference's own thread topology (wcw.c:604-648: up to 6 fft1_b workers, the timf2 thread with fft1_c / make_timf2 / the
    blanker, the second-fft thread, the narrowband thread), stage functions unchanged, hand-offs by condition events like
    lxsys.c:415-447.  None when the binary is not there (it is built in the build container, where the reference sources are,
    and travels with the snapshot)."""
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not os.access(exe, os.X_OK):
        return None
    from linrad_amd import lib as hiplib
    N1 = 1 << w["fft1_n"]
    M1 = N1 // 2
    nblk = max(256, args.cpu_blocks // 2)
    s = hiplib.synth_defaults(N1, 0)
    iq = hiplib.synth_iq(s, 0, nblk * M1 + 2 * N1)
    lim = strong_liminfo(s, w["fft1_n"])
    with tempfile.TemporaryDirectory() as td:
        fi, fl, fo = (os.path.join(td, x) for x in ("in.bin", "lim.bin", "out.bin"))
        np.asarray(iq, np.int16).tofile(fi)
        lim.tofile(fl)
        try:
            out = subprocess.run(ref_harness_cmd(w, nblk, fi, fl, fo, threads), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                 timeout=300, check=True).stdout
            r = json.loads(out.strip().splitlines()[-1])
        except Exception:  # noqa: BLE001
            return None
    if threads and "threads" not in r:                     # a harness without the thread topology: nothing to report
        return None
    dt = r["loop_seconds"]
    stages = "fft1_b, fft1_c, make_timf2, first_noise_blanker, make_fft2, fft2_mix1_fixed" + (", make_fft3_all, fft3_mix2" if w["fft3_n"] else "")
    cores = int(r.get("threads", 1))
    how = "1 thread" if not threads else f"{cores} threads in the reference's stage topology ({r.get('topology', '')})"
    return {"value": round(nblk * M1 / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "reference",
            "sample": f"{nblk} fft1 blocks ({nblk * M1} samples) of the same workload through the compiled reference ({stages}), {how}, {dt:.1f} s"}


def glue_rate(args, w, batches=(1, 2, 4, 8, 16), workers=3):
    """THE DROP-IN'S RATE: oracle/_ref/shim_harness_hip = the reference objects as integration/linrad_hip.patch leaves them + integration/hipshim.c
    as shipped + liblinrad_hip.so, driven in Linrad's thread topology (dispatcher + fft1_b workers, THREAD_TIMF2, THREAD_SECOND_FFT, the
    narrowband thread; wcw.c:401-441, 250-304, 476-500) with the samples entering through the finish_rx_read hook (rxin.c:1423) one dispatch
    at a time and every host-visible product read back by the glue (fft1_sumsq / fft1_slowsum per averaging period, waterfall lines and
    fft2_powersum_float per line, blanker scalars, every timf3 block; fft3 / mix2 then run as the reference's own host code).  One run per
    gpu.fft1_batch_n = log2(batch) (buf.c:248-257).  PCIe-inclusive by construction; never the headline `value`.
    None when the binary is not there (built in the build container, travels with the snapshot)."""
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "shim_harness_hip")
    if not os.access(exe, os.X_OK):
        return None
    from linrad_amd import lib as hiplib
    N1 = 1 << w["fft1_n"]
    M1 = N1 // 2
    ring_log2 = 24 if w["fft1_n"] >= 13 else 22
    nring = (1 << ring_log2) // 4
    s = hiplib.synth_defaults(N1, 0)
    iq = hiplib.synth_iq(s, 0, nring)
    lim = strong_liminfo(s, w["fft1_n"])
    runs = []
    with tempfile.TemporaryDirectory() as td:
        fi, fl, fo = (os.path.join(td, x) for x in ("in.bin", "lim.bin", "out.bin"))
        np.asarray(iq, np.int16).tofile(fi)
        lim.tofile(fl)
        for b in batches:
            nblk = args.glue_blocks * (1 if b < 4 else 2 if b < 16 else 4)
            cmd = [c for c in ref_harness_cmd(w, nblk, fi, fl, fo) if not c.startswith(("max_fft1n", "max_fft2n"))]
            cmd[0] = exe
            # rings as a patched Linrad sizes them for version 21 (integration/linrad_hip.patch, buf.c): fft1 ring of 256 transforms, timf2 of at
            # least 64 fft1 blocks -- the stages behind fft1_b take what has accumulated in one library call, up to 64 blocks
            t2log = int(np.log2(max(8 << max(w["fft2_n"], w["fft1_n"]), 256 * M1)))
            cmd += ["max_fft1n=256", "max_fft2n=64", f"timf2pow_log2={t2log}", f"timf1_log2={ring_log2}", "shim_threads=2", f"shim_workers={workers}", f"shim_batch={b}", "warm=512",
                    "shim_sparse=1"]                  # hip_open decides about the fft2 ring like in a patched xlinrad64 (AFC off here: sparse)
            try:
                # a run lasts 0.1 - 0.3 s: the median of three (the same input, a new process each) is what is reported, all three are listed
                reps = []
                for _ in range(max(1, args.glue_repeats)):
                    pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, check=True)
                    out = pr.stdout
                    for ln in pr.stderr.splitlines():        # HIPSHIM_PROF=1: the glue's own phase times
                        if ln.startswith("HIPSHIM_PROF"):
                            print(f"[{w['key']} fft1_batch_n {int(np.log2(b))}] {ln}", file=sys.stderr)
                    reps.append(json.loads(out.strip().splitlines()[-1]))
                reps.sort(key=lambda q: q["samples"] / q["loop_seconds"])
                r = reps[len(reps) // 2]
            except subprocess.CalledProcessError as e:
                runs.append({"fft1_batch_n": int(np.log2(b)), "error": (e.stderr or "")[-300:]})
                continue
            except Exception as e:  # noqa: BLE001
                runs.append({"fft1_batch_n": int(np.log2(b)), "error": repr(e)})
                continue
            v = r["samples"] / r["loop_seconds"] / 1e6
            runs.append({"fft1_batch_n": int(np.log2(b)), "blocks_per_fft1_b_call": b, "value": round(v, 1), "unit": "Msamples/s",
                         "us_per_block": round(1e6 * r["loop_seconds"] / r["blocks"], 2), "blocks": r["blocks"], "seconds": round(r["loop_seconds"], 3),
                         "realtime_factor": {k: round(v * 1e6 / rate, 2) for k, rate in (("10Msps", 10e6), ("40Msps", 40e6), ("160Msps", 160e6))},
                         "all_runs_Msamples_per_s": [round(q["samples"] / q["loop_seconds"] / 1e6, 1) for q in reps],
                         "stage_calls": r.get("stage_calls"), "threads": r.get("threads")})
    return {"what": "patched reference objects + integration/hipshim.c + liblinrad_hip.so (oracle/_ref/shim_harness_hip timing=1 shim_threads=2): "
                    "samples through the finish_rx_read hook, Linrad's stage threads, all read-backs of a running xlinrad64",
            "config": w["text"].split(":")[0], "fft1_size": N1, "fft2_size": 1 << w["fft2_n"], "fft3_size": (1 << w["fft3_n"]) if w["fft3_n"] else 0,
            "fft1_b_workers": workers, "runs": runs}


def cpu_worker(w, nblk, channel):
    """One single-thread oracle pipeline (child process of cpu_baseline_all_cores); prints its loop time."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import open_oracle
    from linrad_amd import lib as hiplib
    cfg = chain_config(w["fft1_n"], w["fft2_n"], batch=min(32, nblk), fft3_n=w["fft3_n"], mix2_n=w["mix2_n"])
    rx = setup_receiver(cfg, channel, open_oracle, hiplib)
    rx.wideband_dsp(min(32, nblk), cfg.max_batch)
    t0 = time.perf_counter()
    rx.wideband_dsp(nblk, cfg.max_batch)
    print(json.dumps({"seconds": time.perf_counter() - t0}))


def cpu_baseline_all_cores(args, w):
    """Upper bound for any threading of the CPU path on this host: one independent single-thread oracle pipeline per core
    (at most 16), one channel each, started together.  None if a child fails."""
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ncpu = min(ncpu, 16)
    nblk = max(256, args.cpu_blocks // 8)
    cmd = [sys.executable, os.path.abspath(__file__), "--fft1-n", str(w["fft1_n"]), "--fft2-n", str(w["fft2_n"]), "--fft3-n", str(w["fft3_n"]),
           "--cpu-blocks", str(nblk)]
    t0 = time.perf_counter()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    procs = [subprocess.Popen(cmd + ["--cpu-worker", str(ch)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
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
    M1 = (1 << w["fft1_n"]) // 2
    dt = max(times)
    return {"value": round(ncpu * nblk * M1 / dt / 1e6, 4), "unit": "Msamples/s", "cores": ncpu, "kind": "port",
            "sample": f"{ncpu} processes x {nblk} fft1 blocks, one channel each, slowest {dt:.1f} s "
                      f"(wall incl. start-up {time.perf_counter() - t0:.1f} s)"}


# --------------------------------------------------------------------------------------------------------- rank spawner
def spawn_ranks(args):
    """`python bench.py --gpus N` run by hand: start the N ranks as child processes, one per GPU, before anything in this
    process touches the GPU (no HIP call, no torch.cuda call here), wait for them and pass their status on.  Rank 0's
    JSON line goes to this process's stdout."""
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    print(f"bench.py: started {n} ranks, pids {[p.pid for p in procs]}, rendezvous 127.0.0.1:{port}", file=sys.stderr, flush=True)
    rc, failed = 0, []
    deadline = time.time() + args.spawn_timeout
    alive = list(range(n))
    while alive:
        for r in list(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.remove(r)
            if code != 0:
                failed.append((r, procs[r].pid, code))
        if failed or time.time() > deadline:
            for r in alive:                                    # a rank died (or the run overran): its peers would wait for ever
                procs[r].terminate()
            t_end = time.time() + 10
            for r in alive:
                try:
                    procs[r].wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            if not failed:
                failed.append((-1, 0, "timeout"))
            break
        time.sleep(0.05)
    if failed:
        for r, pid, code in failed:
            print(f"bench.py: rank {r} (pid {pid}) exited with {code}", file=sys.stderr)
        rc = 1
    return rc


# --------------------------------------------------------------------------------------------------------- one measurement
def measure(args, w, rank, local_rank, world, dist, torch, hiplib, steps, warmup, with_stage_times=True):
    """Time `steps` steps of workload `w` on this rank's GPU; returns the fields of the JSON line that depend on it."""
    N1, N2, M1 = 1 << w["fft1_n"], 1 << w["fft2_n"], (1 << w["fft1_n"]) // 2
    coupled, combine = w["mode"] == "coupled", w["mode"] == "combine"
    cfg = chain_config(w["fft1_n"], w["fft2_n"], batch=args.batch, device=local_rank, fft3_n=w["fft3_n"], mix2_n=w["mix2_n"], rounds=args.rounds)
    if args.real_input:
        cfg.timf1_real_input = 1
    # nothing on this path reads the fft1_float ring (make_timf2 is its consumer): the fused forward transform then stores only the
    # strong bins its second pass needs (cfg.fft1_float_sparse, include/linrad_hip.h); --fft1-float full keeps every bin
    cfg.fft1_float_sparse = 0 if args.fft1_float == "full" else 1
    # likewise fft2_float: power sums and waterfall lines are formed inside the transform kernels and mix1 cuts a band of mix1.size bins
    cfg.fft2_float_sparse = 0 if (args.fft2_float == "full" or args.spurs) else 1      # spur acquisition reads whole transforms
    if coupled:
        cfg.blanker_channels, cfg.timf1_channel_index = 2, rank & 1
    clever_g = None
    if args.clever:
        # the linear blanker needs the calibrated receiver's pulse response (init_blanker, buf.c:1771-2057): the tables the compiled
        # reference built from a synthetic amplitude calibration, carried as data by the golden fixture of the parity tests
        clever_g = dict(np.load(os.path.join(ROOT, "tests", "golden", "clever_n10_n12.npz")))
        cfg.blanker_pulsewidth, cfg.blnfit_range = int(clever_g["bln_ints"][1]), int(clever_g["bln_ints"][3])
    rx = setup_receiver(cfg, channel_of_rank(rank), hiplib.open_hip, hiplib)

    def clever_on():
        """The operator's click in the high-resolution graph (hires_graph.c:1160-1162): the linear blanker goes on in a receiver that
        is already running -- noise floor measured, selective limiter past its start-up (its noise-floor pass waits for spek_avgnum
        spectra, sellim.c:860; until then the carriers' skirts stay with the weak signal and most samples exceed any sensible limit,
        a case the reference's sample walk and this library's region walk both take minutes over)."""
        bi, bf = clever_g["bln_ints"], clever_g["bln_fparams"]
        floor = rx.blanker_state().timf2_noise_floor
        rx.set_blanker_tables(bln=clever_g["bln"].reshape(-1, 4)[:, :3], refpulse=clever_g["blanker_refpulse"], phasefunc=clever_g["blanker_phasefunc"],
                              pulindex=clever_g["blanker_pulindex"], largest_blnfit=int(bi[2]), clever_bln_factor=float(bf[1]),
                              clever_bln_limit=int(np.float32(floor) * np.float32(bf[1])), liminfo_amplitude_factor=rx.liminfo_amplitude_factor())
    samples_per_step = args.batch * args.rounds * M1
    dev = torch.device("cuda", local_rank)
    use_dist = dist is not None
    xchg = torch.zeros(N1, dtype=torch.float32, device=dev) if use_dist else None
    # the context's own stream, wrapped so that torch / RCCL work can be ordered against it without host waits
    # (and a side stream for the power-sum all-reduce: the legacy default stream would serialise with every blocking stream)
    lrh_stream = torch.cuda.ExternalStream(rx.stream_handle(), device=dev) if use_dist else None
    comm_stream = torch.cuda.Stream(device=dev) if use_dist else None

    host_ring = None
    if args.stream_host:
        s0 = hiplib.synth_defaults(N1, channel_of_rank(rank))
        host_ring = np.ascontiguousarray(hiplib.synth_iq(s0, 0, cfg.timf1_bytes // 4))
        rx.host_register(host_ring)
        if args.batch * args.rounds * M1 * 4 > cfg.timf1_bytes // 2:
            raise SystemExit("--stream-host: one step's samples must fit half the timf1 ring (use --rounds 1)")
    step_bytes = args.batch * args.rounds * M1 * 4
    wr = [0]

    if coupled:
        from linrad_amd.multichan import install_exchange, run_coupled
        rx.set_pol(0.8, 0.36, -0.48)
        if not args.coupled_stages:
            install_exchange(rx, dist, dev)                # lrh_wideband_dsp asks for the collectives at its exchange points (lrh_set_exchange)
    if combine:
        # phased array (BASELINE configs[4]): steer at the synthetic sky signal, whose phase on channel c is 0.7 c rad
        # (SURVEY 8d); beam B is the same aperture pointed half a beam away
        from linrad_amd.multichan import coupled_fft3_mix2
        th = 0.7 * channel_of_rank(rank)
        rx.set_combine_weights(np.exp(-1j * th) / world, np.exp(-1j * (th + np.pi * channel_of_rank(rank) / max(world, 1))) / world)

    def narrow_tail():
        """narrowband side with the coherent combine: fft3 of the own channel, the all-reduce of the weighted mix2 bins
        (stream-ordered on the context's stream, no host wait), filter + back transform of the combined beam"""
        k3 = rx.fft3_available()
        cap = max(1, cfg.max_fft3n // 2)
        while k3 > 0:
            k3b = min(k3, cap)
            rx.make_fft3_all(k3b)
            coupled_fft3_mix2(rx, k3b, dist, dev)
            k3 -= k3b

    # the selective limiter decides the strong / weak routing from the device-resident power spectra once per step
    # (fft1_update_liminfo, sellim.c:738; reference defaults, the one-second hold-off at the workload's sample rate); until its
    # first run the table is the hand-made one of setup_receiver
    sel = None
    if args.sellim and not coupled:
        from linrad_amd.abi import default_sellim
        # blanker_ston_fft1 is a slider of the high-resolution graph (hires_graph.c:710).  The SURVEY 8d test signal carries 20000-LSB
        # impulses whose spectrum ripples across the band at 36 x the noise power; 30 keeps that ripple with the weak signal (51 bins
        # routed strong: the carriers and their skirts), the 4 of a quiet band would route 38 % of the bins
        sel = default_sellim(cfg, fft1_blocktime=M1 / 160e6, blanker_ston_fft1=30.0, exact_stats=0,
                             blanker_ston_fft2=30.0, fft2_blocktime=(N2 // 2) / 160e6, sellim_par1=args.limiter2_par1)
        # inside lrh_wideband_dsp, at the end of every round: wideband_dsp's own limiter calls (wcw.c:1124-1133), once per pass of the loop
        rx.wideband_limiter(sel, args.limiter2)

    def step():
        if coupled and args.coupled_stages:
            run_coupled(rx, args.batch * args.rounds, args.batch, dist, device=dev, xy=True, pol=True)
            return
        if coupled:
            rx.wideband_dsp(args.batch * args.rounds, args.batch)
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
        if combine:
            narrow_tail()
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

    if sel is not None:
        # part of the set-up, like the hand-made table it replaces: the limiter's noise-floor pass only starts once spek_avgnum
        # spectra have been seen (sellim.c:866), until then the medium carriers stay in the weak stream and latch the blanker
        for _ in range(6):
            rx.wideband_dsp(args.batch, args.batch)
    spur_info = None
    if args.spurs and not coupled:
        # the carriers of the SURVEY 8d signal are taken out by the spur loop: acquired on the device-resident spectra (lrh_spur_acquire)
        # after one step, tracked and subtracted by k_spur inside every lrh_make_fft2 from then on
        from linrad_amd.spurs import spur_spectra
        step()
        rx.sync()
        rx.spur_config(args.spurs, args.spur_speknum, spur_spectra(2))
        s0 = hiplib.synth_defaults(N1, channel_of_rank(rank))
        order = np.argsort(-np.asarray(s0.carrier_amp[:s0.ncarriers]))
        locked = 0
        for i in order[:args.spurs]:
            f2 = N2 / 2 + s0.carrier_bin[i] * N2 / s0.fft_size
            locked += int(rx.spur_acquire(int(round(f2)) - 3))
        spur_info = {"requested": args.spurs, "locked": locked, "spur_speknum": args.spur_speknum}
    for _ in range(warmup):
        step()
    if clever_g is not None:
        clever_on()
        for _ in range(2):
            step()
    barrier()
    t0 = time.perf_counter()
    rx.timer_start()
    for _ in range(steps):
        step()
    t_enq = time.perf_counter() - t0                       # host-side enqueue time (the device runs behind it)
    ev_ms = rx.timer_stop()
    barrier()
    dt = time.perf_counter() - t0
    host = {k: rx.profile_get("host:" + k) for k in ("mix1_phases", "wideband_dsp", "wideband_dsp_cpu", "staging_wait")}
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = world * steps * samples_per_step / dt / 1e6
    res = {"value": round(value, 2), "ms_per_step": round(1e3 * dt / steps, 4), "steps": steps, "warmup": warmup,
           "event_ms_per_step": round(ev_ms / steps, 4), "host_enqueue_ms_per_step": round(1e3 * t_enq / steps, 4),
           "host_cpu": {"mix1_phase_ms_per_call": round(host["mix1_phases"][0] / max(host["mix1_phases"][1], 1), 4),
                        "wideband_dsp_ms_per_call": round(host["wideband_dsp"][0] / max(host["wideband_dsp"][1], 1), 4),
                        "wideband_dsp_cpu_ms_per_call": round(host["wideband_dsp_cpu"][0] / max(host["wideband_dsp_cpu"][1], 1), 4),
                        "staging_wait_ms_per_call": round(host["staging_wait"][0] / max(host["staging_wait"][1], 1), 4)},
           "samples_per_step": samples_per_step,
           "workload": workload_name(w, args.batch) + ("_full" if (args.fft1_float == "full" and args.fft2_float == "full") else ""),   # key into profiles/*_traffic.json
           "routing": ("selective limiter on the device at the end of every round inside lrh_wideband_dsp (lrh_wideband_limiter: fft1_update_liminfo%s, "
                       "wcw.c:1124-1133), strong bins now %d of %d" %
                       (" + fft2_update_liminfo (sellim_par1 = %d)" % args.limiter2_par1 if args.limiter2 else "", int(np.count_nonzero(rx.get_liminfo())), N1)) if sel is not None else "fixed table (strong carriers routed by hand)",
           "config_text": w["text"].format(N1=N1, N2=N2, Nm=N2 >> 6, N3=(1 << w["fft3_n"]) if w["fft3_n"] else 0,
                                           Nm2=(1 << w["mix2_n"]) if w["mix2_n"] else 0, rounds=args.rounds, batch=args.batch,
                                           samples=samples_per_step, world=world)}

    # ---- per-kernel timing with HIP events on the streams the kernels are launched on (rank 0), in the SAME two-stream
    # schedule as the timed loop (lrh_profile_enable(2)); a serial pass afterwards gives the stand-alone times
    stages, alone, roof = {}, {}, None
    # (a coupled pair issues its collectives inside lrh_wideband_dsp: with two ranks BOTH must make the profiled calls, or rank 0 waits for
    # a partner that has already left -- the stage-by-stage form, --coupled-stages, is not profiled)
    if (rank == 0 or (coupled and world > 1)) and with_stage_times and not (coupled and args.coupled_stages):
        nprof = max(3, min(steps, 10))
        rx.profile_enable(2)
        for _ in range(nprof):
            rx.wideband_dsp(args.batch * args.rounds, args.batch)
            if combine:
                k3 = rx.fft3_available()                   # keep the rings moving; the collective is not profiled here
                while k3 > 0:
                    k3b = min(k3, max(1, cfg.max_fft3n // 2))
                    rx.make_fft3_all(k3b); rx.mix2_pol_begin(k3b); rx.fft3_mix2(k3b)
                    k3 -= k3b
        rx.sync()
        for k in STAGES:
            ms, n = rx.profile_get(k)
            if n:
                stages[k] = {"ms_total": round(ms, 4), "launches": n, "avg_us": round(1e3 * ms / n, 2)}
        rx.profile_enable(1)
        rx.wideband_dsp(args.batch, args.batch)                # first serial pass after the two-stream run: discarded
        rx.sync()
        rx.profile_enable(1)                                   # resets the accumulators
        for _ in range(nprof):
            rx.wideband_dsp(args.batch, args.batch)
        rx.sync()
        for k in STAGES:
            ms, n = rx.profile_get(k)
            if n:
                alone[k] = round(1e3 * ms / n, 2)
        rx.profile_enable(0)
        fused_sums = "sumsq" not in stages

        def alg_of(k):                                       # algorithmic bytes per sample of a stage in the form this workload runs it
            if k == "fft2":                                  # with cfg.fft2_float_sparse the spectrum (8 B x 2 transforms per sample) is not written
                return ALG_BYTES["fft2_single" if w["fft2_n"] <= 14 else "fft2"] - (ALG_BYTES["fft2_spectrum_out"] if cfg.fft2_float_sparse else 0.0)
            return ALG_BYTES[k]
        for k, st in stages.items():
            st["avg_us_alone"] = alone.get(k)
            if k in ALG_BYTES:                             # stage rates (secondary metric): one batch per launch
                per = alg_of(k) + (ALG_BYTES["sumsq"] if (k == "timf2" and fused_sums) else 0.0)
                launches_per_round = st["launches"] / (nprof * args.rounds)
                st["Msamples_per_s"] = round(args.batch * M1 / (st["avg_us"] * launches_per_round * 1e-6) / 1e6, 1)
                st["alg_GBps"] = round(st["Msamples_per_s"] * per / 1e3, 1)
                tr = measured_traffic(k, res["workload"])
                if tr:
                    st["counter_GBps"] = round(tr["traffic_bytes_per_launch"] / (st["avg_us"] * 1e-6) / 1e9, 1)
                if st["alg_GBps"] > HBM_PEAK_GBS:
                    st["note"] = ("algorithmic bytes (SURVEY 8d: every stage one full pass) exceed what this kernel moves: the overlapped half "
                                  "of each transform stays in registers and the per-transform power ring is never written")
        # The HBM roofline is that of the stage which has to move the most HBM bytes per round (the algorithmic figures: deterministic, where
        # stand-alone times and counter bytes of two stages lie within a percent of each other): fft2 on both configs.  The stage with the longest stand-alone time
        # -- since round 3 a near tie between fft2 (376 us) and k_fft1w (378 us), which is bound by its two 16384-point transforms' LDS
        # exchanges and not by bytes -- is reported beside it as `longest_kernel` with its fp32 rate.  (In-schedule durations of the
        # side-stream stages are stretched by the kernels they overlap with and say nothing about them.)
        def bytes_per_round(k):
            per_launch = alg_of(k) * (args.batch * M1) / (stages[k]["launches"] / (nprof * args.rounds))
            return per_launch * stages[k]["launches"]
        cands = [k for k in stages if k in ALG_BYTES]
        dom = max(cands, key=bytes_per_round)
        longest = max(cands, key=lambda k: (alone.get(k) or 0.0) * stages[k]["launches"])
        per_sample = alg_of(dom) + (ALG_BYTES["sumsq"] if (dom == "timf2" and fused_sums) else 0.0)
        launches_per_round = stages[dom]["launches"] / (nprof * args.rounds)
        alg_bytes_launch = per_sample * (args.batch * M1) / launches_per_round
        avg_s = stages[dom]["avg_us"] * 1e-6
        achieved = alg_bytes_launch / avg_s / 1e9
        tr = measured_traffic(dom, res["workload"])
        traffic = tr["traffic_bytes_per_launch"] if tr else None
        chain_alg = alg_bytes_chain(w)
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_file": ("profiles/" + _TRAFFIC["file"]) if (traffic and _TRAFFIC.get("file")) else None,
                "frac_alg": round(achieved / HBM_PEAK_GBS, 4),
                "frac_counter": round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                "achieved_counter": round(traffic / avg_s / 1e9, 1) if traffic else None,
                "primary": "frac_counter: HBM bytes the kernel really moved (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/%s, same sources as this library) " % (_TRAFFIC.get("file") or "r0N_traffic.json: none matches these sources") +
                           "/ average launch time / 8 TB/s; frac (= frac_alg) prices the SURVEY 8d algorithmic bytes, part of which this "
                           "kernel eliminates (overlap read-modify-write 32 B, liminfo floats 8 B of 92 B per sample)",
                "timer": "HIP events around each launch on the stream it is launched on, two-stream schedule of the timed loop "
                         "(lrh_profile_enable(2)); avg_launch_us_alone = same kernel in the serial order",
                "alg_bytes_per_launch": int(alg_bytes_launch), "alg_bytes_per_sample": per_sample,
                "avg_launch_us": round(avg_s * 1e6, 2), "avg_launch_us_alone": alone.get(dom),
                "chain_alg_bytes_per_sample": chain_alg,
                "chain_alg_GBps": round(value / world * chain_alg / 1e3, 1),
                "chain_frac_alg": round(value / world * chain_alg / 1e3 / HBM_PEAK_GBS, 4),
                "selection": "the stage that moves the most HBM bytes per round; the stage with the longest stand-alone time is `longest_kernel`"}
        if True:                                             # always: the byte-dominant stage and the longest kernel are different questions
            lt = measured_traffic(longest, res["workload"])
            lp = stages[longest]["launches"] / (nprof * args.rounds)
            l_s = stages[longest]["avg_us"] * 1e-6
            lk = {"kernel": longest, "avg_launch_us": stages[longest]["avg_us"], "avg_launch_us_alone": alone.get(longest),
                  "hbm_frac_alg": round(alg_of(longest) * (args.batch * M1) / lp / l_s / 1e9 / HBM_PEAK_GBS, 4),
                  "hbm_frac_counter": round(lt["traffic_bytes_per_launch"] / l_s / 1e9 / HBM_PEAK_GBS, 4) if lt else None}
            if longest == "fft1w":
                # two fft1_size-point complex transforms per block (forward, weak-stream back transform) at 5 N log2 N flops each; the guide's
                # vector fp32 peak (157.3 TFLOP/s).  Neither bytes nor flops bound it: the transforms' register <-> LDS exchanges, the issue
                # of its memory operations and five workgroup barriers per block at two waves per SIMD (profiles/r04_sq_counters.txt)
                flops = 2 * 5.0 * N1 * w["fft1_n"] * args.batch / lp
                lk.update({"bound": "valu_fp32 / LDS exchange / memory issue", "fp32_TFLOPs": round(flops / l_s / 1e12, 1), "fp32_peak_TFLOPs": 157.3,
                           "fp32_frac": round(flops / l_s / 1e12 / 157.3, 4)})
            roof["longest_kernel"] = lk
        trs = [measured_traffic(k, res["workload"]) for k in stages]
        if all(t is not None for t, k in zip(trs, stages) if k in ALG_BYTES):
            # whole-round HBM traffic from the counters: sum over the profiled kernels x their launches per round
            tot = sum(t["traffic_bytes_per_launch"] * stages[k]["launches"] / (nprof * args.rounds) for t, k in zip(trs, stages) if t)
            roof["chain_counter_bytes_per_sample"] = round(tot / (args.batch * M1), 1)
            roof["chain_frac_counter"] = round(value / world * 1e6 * tot / (args.batch * M1) / 1e9 / HBM_PEAK_GBS, 4)
        # context for the 8 TB/s nominal peak: what a plain device-to-device copy reaches on this box (read + write bytes)
        try:
            nbytes = 1 << 30
            src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
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
    if rank == 0:
        bs = rx.blanker_state()
        res["blanker"] = {"noise_floor": bs.timf2_noise_floor, "limit": bs.stupid_bln_limit,
                          "cleared_rate_pct": round(bs.stupid_blanker_rate, 3), "slow_path_calls": bs.slow_path_calls}
        if args.clever:
            res["blanker"].update(clever_limit=bs.clever_bln_limit, clever_rate_pct=round(bs.clever_blanker_rate, 3), last_call_fitted=bs.last_call_fitted,
                                  last_call_rejected=bs.last_call_rejected, one_wave_replays=bs.clever_serial_calls)
    res["roofline"], res["stages"] = roof, stages
    if spur_info is not None:
        spur_info["locked_at_end"] = sum(1 for q in rx.spur_get() if q.spur_flag == 0)
        res["spurs"] = spur_info
    if host_ring is not None:
        rx.host_unregister(host_ring)
    rx.close()
    return res


LINE_LIMIT = 6000          # bytes: the driver keeps an 8 KB tail of stdout and parses the LAST line (round 5's 22.9 KB line did not survive that)


def compact_line(out):
    """The ONE line the driver parses, from the full result `out` (which goes to gpurun_out/bench_detail.json and stderr): the contract's
    fields, the roofline and cpu_baseline objects, and one number each for the objects measured beside the headline."""
    def pick(d, keys):
        return {k: d[k] for k in keys if isinstance(d, dict) and k in d}
    roof = out.get("roofline") or {}
    line = pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    cfg = dict(out.get("config") or {})
    cfg["workload"] = (cfg.get("workload") or "")[:420]
    for k in ("fft1_float", "fft2_float"):
        if k in cfg:
            cfg[k] = "sparse" if not str(cfg[k]).startswith("every bin") else "full"
    line["config"] = cfg
    line["roofline"] = pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_counter", "achieved_counter", "alg_bytes_per_launch", "alg_bytes_per_sample",
                                   "avg_launch_us", "avg_launch_us_alone", "chain_alg_bytes_per_sample", "chain_frac_alg", "chain_counter_bytes_per_sample", "chain_frac_counter",
                                   "device_copy_GBps", "traffic_file"))
    lk = roof.get("longest_kernel")
    if lk:
        line["roofline"]["longest_kernel"] = pick(lk, ("kernel", "avg_launch_us", "avg_launch_us_alone", "hbm_frac_counter", "fp32_frac"))
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = dict(pick(cb, ("value", "unit", "cores", "kind")), sample=(cb.get("sample") or "")[:200]) if cb else None
    for k in ("cpu_baseline_port", "cpu_baseline_reference_threads", "cpu_baseline_all_cores"):
        if out.get(k):
            line[k] = pick(out[k], ("value", "cores", "kind"))
    if out.get("mode"):
        line["mode"] = out["mode"][:160]
    if out.get("stages"):
        line["stage_us"] = {k: v.get("avg_us") for k, v in out["stages"].items()}      # average launch time per stage in the timed schedule (HIP events)
    line["realtime_factor"] = out.get("realtime_factor")
    line["event_ms_per_step"] = out.get("event_ms_per_step")
    if out.get("blanker"):
        line["blanker"] = pick(out["blanker"], ("noise_floor", "cleared_rate_pct"))
    for k in ("full_rings", "secondary"):
        o = out.get(k)
        if o:
            r = o.get("roofline") or {}
            line[k] = dict(pick(o, ("value", "ms_per_step", "steps", "error")), frac=r.get("frac"), frac_counter=r.get("frac_counter"), kernel=r.get("kernel"))
    if out.get("round_sweep"):
        line["round_sweep"] = out["round_sweep"]
    g = out.get("glue")
    if isinstance(g, list):
        line["glue"] = {"unit": "Msamples/s per gpu.fft1_batch_n (patched reference objects + hipshim.c + liblinrad_hip.so, PCIe inclusive)"}
        for q in g:
            if q:
                line["glue"][q["config"]] = {str(r["fft1_batch_n"]): r.get("value", "error") for r in q["runs"]}
    elif g:
        line["glue"] = g
    line["detail"] = "gpurun_out/bench_detail.json (stages, per-stage glue call tables, notes); also on stderr"
    txt = json.dumps(line, allow_nan=False)
    if len(txt) > LINE_LIMIT:                                # never again an unparseable line: drop the optional objects, largest first
        for k in ("glue", "round_sweep", "stage_us", "realtime_factor", "cpu_baseline_all_cores", "cpu_baseline_reference_threads", "cpu_baseline_port", "blanker", "secondary", "full_rings"):
            line.pop(k, None)
            txt = json.dumps(line, allow_nan=False)
            if len(txt) <= LINE_LIMIT:
                break
    return txt


def round_sweep(args, primary, rank, local_rank, world, dist, torch, hiplib):
    """The headline workload at rounds of 256 / 1024 / 4096 / 8192 fft1 blocks (the same samples per step): how much of `value` is the
    amortisation of launches and pipeline fill over a long round (8192 blocks = 0.42 s of signal at 160 Msps)."""
    total = args.batch * args.rounds
    res = {}
    for b in (256, 1024, 4096, 8192):
        if b > total or total % b:
            continue
        a = argparse.Namespace(**vars(args))
        a.batch, a.rounds = b, total // b
        try:
            res[str(b)] = measure(a, primary, rank, local_rank, world, dist, torch, hiplib, max(3, args.steps // 10), 2, with_stage_times=False)["value"]
        except Exception as e:  # noqa: BLE001
            res[str(b)] = repr(e)[:80]
    return {"unit": "Msamples/s by fft1 blocks per round", **res}


WORKLOADS = {
    # BASELINE configs[2] as SURVEY 8d C3 makes it concrete
    "c2": dict(mode="chain", fft3=True,
               text="BASELINE configs[2] (SURVEY 8d C3): 1 channel/GPU complex-int16 IQ, fft1_size={N1} sin^2 50% overlap, timf2 + stupid blanker, "
                    "fft2_size={N2} sin^2 (four-step), mix1 size {Nm}, fft3_size={N3} + mix2 size {Nm2}; {rounds} x {batch} fft1 blocks "
                    "({samples} samples) per step, device-resident ring"),
    # BASELINE configs[1]
    "c1": dict(mode="chain", fft3=False,
               text="BASELINE configs[1]: 1 channel/GPU complex-int16 IQ, fft1_size={N1} sin^2 50% overlap, timf2 + stupid blanker, "
                    "fft2_size={N2} sin^2, mix1 size {Nm}; {rounds} x {batch} fft1 blocks ({samples} samples) per step, device-resident ring"),
    # BASELINE configs[4]: the configs[2] chain per channel + the coherent combine
    "c4": dict(mode="combine", fft3=True,
               text="BASELINE configs[4]: {world}-channel phased array, one channel per GPU, each through fft1_size={N1} / timf2 + blanker / "
                    "fft2_size={N2} / mix1 {Nm} / fft3 {N3}; RCCL all-reduce of the per-bin channel power sums and of the weighted mix2 bins "
                    "(coherent combine, two beams, mix2 size {Nm2}) inside the timed region; {rounds} x {batch} fft1 blocks ({samples} samples) "
                    "per channel and step"),
    # BASELINE configs[3]
    "c3": dict(mode="coupled", fft3=True,
               text="BASELINE configs[3]: polarisation pair (ui.rx_rf_channels = 2 sharded one channel per GPU), fft1_size={N1}, coupled blanker "
                    "(2 all-reduces per call), fft2_size={N2} cross products (all-gather), mix1 {Nm}, fft3 {N3} + polarisation transform in "
                    "mix2 {Nm2} (all-reduce); {rounds} x {batch} fft1 blocks ({samples} samples) per channel and step"),
}


def make_workload(key, fft1_n, fft2_n, fft3_n, mix2_n):
    w = dict(WORKLOADS[key], key=key, fft1_n=fft1_n, fft2_n=fft2_n)
    w["fft3_n"], w["mix2_n"] = (fft3_n, mix2_n) if (w["fft3"] and fft3_n) else (0, 0)
    return w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    # batch = fft1 blocks handed to every kernel launch.  Throughput grows with it (fuller waves of workgroups, launch and
    # host overheads amortised: 23.3 Gsamples/s at 1024, 26.9 at 2048, 29.2 at 4096, 29.3 at 8192, round 1) at the price of
    # batch*8192 samples of latency; 4096 blocks are 1.1 ms of signal at the rate the chain sustains.
    # Round 5: with the kernels twice as fast as then, what a round costs beside them shows again -- the prologue block of every workgroup's run in k_fft1v
    # (1/16 of a run of 16), five kernel-to-kernel hand-overs on the main stream (~6 us each), the tails of the launches: 41.0 Gsamples/s at 4096 blocks per round,
    # 45.4 at 8192 (same 268 435 456 samples per step: 4 rounds of 8192); 16384 does not fit the 32-bit ring offsets.
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--rounds", type=int, default=4, help="batches of --batch fft1 blocks per step (pipelined on two streams; the pipeline fills and drains once per step)")
    ap.add_argument("--fft1-n", type=int, default=14)
    ap.add_argument("--fft2-n", type=int, default=None, help="log2 fft2_size; default 16 (configs[2]); 12 = configs[1]")
    ap.add_argument("--fft3-n", type=int, default=12, help="log2 fft3_size behind mix1 (0: chain ends at mix1)")
    ap.add_argument("--mix2-n", type=int, default=8)
    ap.add_argument("--cpu-blocks", type=int, default=16384)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the second workload of the default run")
    ap.add_argument("--no-sellim", dest="sellim", action="store_false",
                    help="keep the hand-made routing table instead of running the selective limiter (lrh_fft1_update_liminfo) once per step")
    ap.add_argument("--limiter2", action="store_true",
                    help="also run the second limiter (fft2_update_liminfo) at the end of every round; lrh_wideband_dsp then leaves its one-round-late schedule")
    ap.add_argument("--limiter2-par1", type=int, default=2, choices=(0, 1, 2),
                    help="with --limiter2: hg.sellim_par1, the second limiter's variant (2 = the reference's setting)")
    ap.add_argument("--stream-host", action="store_true",
                    help="PCIe-inclusive variant (never the headline value): every step first hands its samples over from "
                         "page-locked host memory with lrh_timf1_write_async, overlapped with the previous step's kernels")
    ap.add_argument("--coupled", action="store_true",
                    help="BASELINE configs[3] as the primary workload: ranks 0/1 are the two channels of a polarisation pair: coupled blanker "
                         "(2 all-reduces per call), fft2 cross products (all-gather), fft3 + polarisation transform in mix2 (all-reduce)")
    ap.add_argument("--combine", action="store_true",
                    help="BASELINE configs[4]'s mode (one channel per rank + coherent combine) also on a single rank: the pre-flight of the multi-GPU "
                         "run (with LRH_BENCH_FORCE_DIST=1 the collectives go through a one-rank RCCL group)")
    ap.add_argument("--fft1-float", choices=("sparse", "full"), default="sparse",
                    help="sparse (default): the fft1_float ring keeps only the strong bins (nobody on the path reads it); full: every bin is stored")
    ap.add_argument("--fft2-float", choices=("sparse", "full"), default="sparse",
                    help="sparse (default): of every fft2 transform only the band mix1 cuts out is stored (cfg.fft2_float_sparse); full: every bin")
    ap.add_argument("--clever", action="store_true", help="linear (\"clever\") blanker in front of the stupid one: pulse search, fit and subtraction (blank1.c:765-1003)")
    ap.add_argument("--spurs", type=int, default=0, help="track and subtract this many of the signal's carriers (eliminate_spurs inside lrh_make_fft2)")
    ap.add_argument("--spur-speknum", type=int, default=16)
    ap.add_argument("--coupled-stages", action="store_true", help="--coupled through the stage calls of linrad_amd.multichan.run_coupled instead of one lrh_wideband_dsp call per step")
    ap.add_argument("--real-input", action="store_true",
                    help="real samples (fft1 version 2): every fft1 block takes 2*M1 reals; value still counts M1 complex-rate samples per block")
    ap.add_argument("--glue-repeats", type=int, default=3, help="runs per fft1_batch_n of the drop-in measurement; the median is reported")
    ap.add_argument("--glue-blocks", type=int, default=8192, help="fft1 blocks per run of the drop-in measurement (`glue` object; x2 / x4 at the larger fft1_b batches)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the `round_sweep` object (the headline workload at rounds of 256 / 1024 / 4096 / 8192 blocks)")
    ap.add_argument("--no-glue", action="store_true", help="skip the `glue` object (patched reference + hipshim.c + liblinrad_hip.so)")
    ap.add_argument("--glue-only", action="store_true", help="only the `glue` object, as one JSON line")
    ap.add_argument("--spawn-timeout", type=float, default=1500.0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker", type=int, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    explicit_fft2 = args.fft2_n is not None
    if args.fft2_n is None:
        args.fft2_n = 16
    if args.cpu_worker is not None:                        # child of cpu_baseline_all_cores: no GPU, no torch
        cpu_worker(make_workload("c2" if args.fft3_n else "c1", args.fft1_n, args.fft2_n, args.fft3_n, args.mix2_n), args.cpu_blocks, args.cpu_worker)
        return 0
    if args.glue_only:                                     # the harness processes own the GPU; nothing here touches it
        print(json.dumps({"glue": [glue_rate(args, make_workload("c2", args.fft1_n, args.fft2_n, args.fft3_n, args.mix2_n)),
                                   glue_rate(args, make_workload("c1", args.fft1_n, 12, 0, 0))]}), flush=True)
        return 0
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:   # started by hand: become the launcher, never touch the GPU here
        return spawn_ranks(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("LRH_BENCH_WATCHDOG"):               # diagnostics: every thread's Python stack to stderr after that many seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["LRH_BENCH_WATCHDOG"]), exit=True)
    if os.environ.get("LRH_BENCH_SAME_DEVICE") == "1":     # rehearsal of the N > 1 logic on a one-GPU box (backend gloo)
        local_rank = 0
    import torch
    dist = None
    force_dist = os.environ.get("LRH_BENCH_FORCE_DIST") == "1"      # exercise the collective path on a single GPU
    backend = os.environ.get("LRH_BENCH_BACKEND", "nccl")
    if world > 1 or force_dist or args.coupled:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        # lazy communicator: an RCCL communicator on the device costs this pipeline ~6 % (measured, round 1) even when
        # idle, so it is only created by the first collective, i.e. when there really is more than one rank
        dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != world or (world > 1 and world != args.gpus):
            raise SystemExit(f"bench.py: the process group has {dist.get_world_size()} ranks, WORLD_SIZE {world}, --gpus {args.gpus}")
    from linrad_amd import lib as hiplib

    if args.coupled and world > 2:
        raise SystemExit("--coupled: a polarisation pair has two channels (Linrad's maximum, SURVEY F4)")
    # primary workload: configs[2] on one GPU; one channel per GPU with the coherent combine on several (configs[4])
    if args.coupled:
        primary = make_workload("c3", args.fft1_n, args.fft2_n if explicit_fft2 else 12, 10, 8)
    elif world > 1 or args.combine:
        primary = make_workload("c4", args.fft1_n, args.fft2_n, args.fft3_n or 12, args.mix2_n)
    else:
        primary = make_workload("c2" if args.fft3_n else "c1", args.fft1_n, args.fft2_n, args.fft3_n, args.mix2_n)
        if args.fft2_n <= 14 and not args.fft3_n:
            primary = make_workload("c1", args.fft1_n, args.fft2_n, 0, 0)
    res = measure(args, primary, rank, local_rank, world, dist, torch, hiplib, args.steps, args.warmup)

    # second workload of the default run: configs[1] beside configs[2] on one GPU; configs[3] (coupled pair) on two
    secondary = None
    default_run = not explicit_fft2 and not args.coupled and not args.combine and not args.stream_host and not args.real_input and not args.no_secondary
    if default_run and world <= 2:
        sw = make_workload("c1", args.fft1_n, 12, 0, 0) if world == 1 else make_workload("c3", args.fft1_n, 12, 10, 8)
        try:
            s = measure(args, sw, rank, local_rank, world, dist, torch, hiplib, max(5, args.steps // 5), max(2, args.warmup // 2))
            secondary = {"config": {"workload": s["config_text"]}, "value": s["value"], "unit": "Msamples/s", "ms_per_step": s["ms_per_step"],
                         "steps": s["steps"], "roofline": s["roofline"], "stages": s["stages"], "blanker": s.get("blanker")}
        except Exception as e:  # noqa: BLE001
            secondary = {"error": repr(e)}

    # third object of the default run: the configuration the Linrad glue opens (integration/hipshim.c: cfg.fft1_float_sparse =
    # cfg.fft2_float_sparse = 0 -- Linrad's graphs, fft1_mix1_* and the AFC read the full rings), same workload as the headline
    full_rings = None
    if default_run and world == 1 and args.fft1_float == "sparse" and args.fft2_float == "sparse":
        fa = argparse.Namespace(**vars(args))
        fa.fft1_float = fa.fft2_float = "full"
        try:
            s = measure(fa, primary, rank, local_rank, world, dist, torch, hiplib, max(5, args.steps // 5), max(2, args.warmup // 2))
            full_rings = {"config": {"workload": s["config_text"], "fft1_float": "every bin stored", "fft2_float": "every bin stored",
                                     "note": "cfg.fft1_float_sparse = cfg.fft2_float_sparse = 0: what integration/hipshim.c opens"},
                          "value": s["value"], "unit": "Msamples/s", "ms_per_step": s["ms_per_step"], "steps": s["steps"],
                          "roofline": s["roofline"], "stages": s["stages"]}
        except Exception as e:  # noqa: BLE001
            full_rings = {"error": repr(e)}

    sweep = None
    if default_run and world == 1 and not args.no_sweep:
        sweep = round_sweep(args, primary, rank, local_rank, world, dist, torch, hiplib)

    cpu = cpu_all = cpu_port = cpu_threads = None
    if rank == 0 and world == 1 and not args.no_cpu:            # (the contract: the CPU baseline is timed on rank 0 of the N = 1 run only)
        cw = dict(primary)
        cpu_port = cpu_baseline(args, cw)
        cpu = cpu_reference(args, cw) or cpu_port
        cpu_threads = cpu_reference(args, cw, threads=-1)
        cpu_all = cpu_baseline_all_cores(args, cw)
    # fourth object of the default run: the drop-in's own rate (patched reference objects + hipshim.c + liblinrad_hip.so as child
    # processes, one per gpu.fft1_batch_n; this process is idle meanwhile) for configs[2] and configs[1]
    glue = None
    if rank == 0 and world == 1 and default_run and not args.no_glue:
        try:
            glue = [glue_rate(args, make_workload("c2", args.fft1_n, args.fft2_n, args.fft3_n, args.mix2_n)),
                    glue_rate(args, make_workload("c1", args.fft1_n, 12, 0, 0))]
            if glue[0] is None:
                glue = None
        except Exception as e:  # noqa: BLE001
            glue = {"error": repr(e)}
    if dist is not None:
        dist.barrier()
        nranks = dist.get_world_size()
        dist.destroy_process_group()
    else:
        nranks = 1
    if rank == 0:
        if nranks != world or (world > 1 and world != args.gpus):      # the line must describe the group the collectives really ran in
            raise SystemExit(f"bench.py: collectives ran in a group of {nranks}, WORLD_SIZE {world}, --gpus {args.gpus}")
        value = res["value"]
        N1, N2 = 1 << primary["fft1_n"], 1 << primary["fft2_n"]
        out = {
            "metric": METRIC, "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": res["config_text"], "fft1_size": N1, "fft2_size": N2,
                       "fft3_size": (1 << primary["fft3_n"]) if primary["fft3_n"] else 0, "batch_blocks": args.batch,
                       "rounds_per_step": args.rounds, "channels": world, "parallelism": f"1 RF channel per GPU x{world}",
                       "collective_world_size": nranks, "backend": (backend if dist is not None else None),
                       "fft1_float": "strong bins only (cfg.fft1_float_sparse: no reader on this path)" if args.fft1_float == "sparse" else "every bin stored",
                       "fft2_float": "the band mix1 cuts out (cfg.fft2_float_sparse: power sums / waterfall inside the transform kernels)" if args.fft2_float == "sparse" else "every bin stored"},
            "mode": {"chain": "lrh_wideband_dsp", "combine": "lrh_wideband_dsp + coherent combine (lrh_mix2_pol_begin / all-reduce / lrh_fft3_mix2)",
                     "coupled": "two coupled channels (polarisation pair): lrh_wideband_dsp with the collectives registered through lrh_set_exchange (linrad_amd.multichan.install_exchange)"}[primary["mode"]],
            "input": "page-locked host ring over PCIe, lrh_timf1_write_async per step" if args.stream_host else "device-resident ring",
            "event_ms_per_step": res["event_ms_per_step"], "host_enqueue_ms_per_step": res["host_enqueue_ms_per_step"], "host_cpu": res["host_cpu"],
            "realtime_factor": {k: round(value / world * 1e6 / r, 1) for k, r in (("10Msps", 10e6), ("40Msps", 40e6), ("160Msps", 160e6))},
            "routing": res.get("routing"), "roofline": res["roofline"], "cpu_baseline": cpu, "cpu_baseline_port": cpu_port, "cpu_baseline_reference_threads": cpu_threads,
            "cpu_baseline_all_cores": cpu_all, "stages": res["stages"], "blanker": res.get("blanker"), "spurs": res.get("spurs"), "secondary": secondary, "full_rings": full_rings, "round_sweep": sweep, "glue": glue,
        }
        # everything measured: a file (merged back from the GPU box) and stderr; stdout carries ONE short line (compact_line)
        detail = json.dumps(out)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as f:
                f.write(detail + "\n")
        except OSError:
            pass
        print(detail, file=sys.stderr, flush=True)
        print(compact_line(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
