#!/usr/bin/env python3
"""Condense a scripts/profile_round.sh output directory into the summaries that are committed under profiles/:
   summary text on stdout; with a second argument, the machine-readable HBM traffic per stage launch and workload
   (read by bench.py for roofline.traffic / frac_counter)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
# stage (bench.py's HIP-event scopes) -> (anchor kernel: one dispatch per stage launch, all kernels of the stage)
STAGE_KERNELS = {
    "fft1": ("k_fft1<", ["k_fft1<", "k_realsplit", "k_foldcorr"]),
    "fft1w": (("k_fft1w<", "k_fft1v<"), ["k_fft1w<", "k_fft1v<"]),
    "timf2s": ("k_timf2<14, 1, false, true>", ["k_timf2<14, 1, false, true>", "k_timf2_sd<"]),   # the direct sum and the transform kernel behind it (one of them returns at once)
    "timf2": ("k_timf2<", ["k_timf2<"]),
    "spur": ("k_spur(", ["k_spur(", "k_spur_patch"]),
    "sumsq": ("k_sumsq(", ["k_sumsq("]),
    "sumsq_join": ("k_sumsq_join", ["k_sumsq_join"]),
    "slowsum": ("k_slowsum", ["k_slowsum"]),
    "clever": ("k_clever_regions", ["k_clever_prep", "k_clever_count", "k_clever_regions", "k_clever(", "k_clever_check", "k_clever_restore"]),
    "blanker": ("k_blank_scan", ["k_blank_scan", "k_blank_runs_pre", "k_blank_runs(", "k_blank_serial", "k_blank_walk", "k_blank_apply", "k_blank_stats", "k_blank_update"]),
    "fft2": (("k_fft2_cols<", "k_fft2_cols16<", "k_fft2<"), ["k_fft2_cols<", "k_fft2_cols16<", "k_fft2_rows<", "k_fft2_rows16x<", "k_fft2<"]),
    "powersum2": ("k_powersum2", ["k_powersum2"]),
    "waterfall": ("k_waterfall", ["k_waterfall"]),
    "mix1": ("k_mix1_back<", ["k_mix1_back<", "k_mix1_out"]),
    "fft3": ("k_fft3<", ["k_fft3<"]),
    "mix2": ("k_mix2_back<", ["k_mix2_back<"]),
}


def first(pattern):
    f = glob.glob(os.path.join(d, pattern), recursive=True)
    return max(f, key=os.path.getmtime) if f else None      # gpurun merges runs into one tree: the newest


def counter_table(tag, ctr):
    """kernel name -> [MEDIAN of the counter over the kernel's dispatches x dispatches, dispatches, mean].
    The median, because the first calls of a run are not the steady state: while the blanker's noise floor settles its
    long-run replay and its apply kernel move GBs (k_blank_runs_pre: 1.5 GB in each of the first 4 calls, 11 KiB afterwards),
    which a mean would spread over every launch."""
    f = first(f"{tag}/**/*counter_collection.csv")
    vals = defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == ctr:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    acc = defaultdict(lambda: [0.0, 0, 0.0])
    for k, v in vals.items():
        v.sort()
        acc[k] = [v[len(v) // 2] * len(v), len(v), sum(v) / len(v)]
    return acc


print("# rocprofv3 summary for", d)
for sub, title in (("stats_headline", "python3 bench.py --steps 10 --warmup 3 --no-cpu --no-secondary: the HEADLINE workload alone, rounds of 8192 fft1 blocks -- "
                                      "one launch shape per kernel, the averages to hold against roofline.avg_launch_us"),
                   ("stats", "python3 bench.py --steps 10 --warmup 3 --no-cpu --no-glue --no-sweep: the default run's three workloads -- headline, configs[1] (k_fft2<12>), "
                             "full_rings (k_fft1v<.., KEEP>; the same fft2 kernels with the spectrum store) -- share the fft2 / timf2 / blanker kernel names")):
    f = first(sub + "/**/*kernel_stats.csv")
    if f:
        print(f"\n## kernel-trace --stats ({title})")
        print("%-58s %6s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
        for r in csv.DictReader(open(f)):
            print("%-58s %6s %12.1f %10.2f %6s" % (r["Name"][:58], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                   float(r["AverageNs"]) / 1e3, r["Percentage"]))
workloads = {}
for wdir in sorted(glob.glob(os.path.join(d, "pmc_fetch_*"))):
    wl = os.path.basename(wdir)[len("pmc_fetch_"):]
    fetch, write = counter_table("pmc_fetch_" + wl, "FETCH_SIZE"), counter_table("pmc_write_" + wl, "WRITE_SIZE")
    print(f"\n## workload {wl}: --pmc FETCH_SIZE / WRITE_SIZE (separate passes, serial schedule), per dispatch in KiB as rocprofv3 reports them: median (mean)")
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch[k][0] + write[k][0])):
        fv, fn, fm = fetch.get(k, [0, 0, 0]); wv, wn, wm = write.get(k, [0, 0, 0])
        print("%-50s n=%4d fetch=%10.1f (%10.1f)  n=%4d write=%10.1f (%10.1f)" % (k.replace("void lrh::", "")[:50], fn, fv / max(fn, 1), fm, wn, wv / max(wn, 1), wm))
    kernels = {}
    for stage, (anchor, pats) in STAGE_KERNELS.items():
        anchors = (anchor,) if isinstance(anchor, str) else anchor
        strong_only = "k_timf2<14, 1, false, true>"           # the sparse second pass behind k_fft1w is a stage of its own

        def has(k, pp):
            return any(q in k for q in pp) and not (stage == "timf2" and strong_only in k)
        nf = sum(v[1] for k, v in fetch.items() if has(k, anchors))
        nw = sum(v[1] for k, v in write.items() if has(k, anchors))
        if not nf or not nw:
            continue
        fsum = sum(v[0] for k, v in fetch.items() if has(k, pats))
        wsum = sum(v[0] for k, v in write.items() if has(k, pats))
        names = sorted({k.split("(")[0].replace("void lrh::", "") for k in fetch if has(k, pats)})
        kernels[stage] = {"kernels": names, "FETCH_SIZE_KiB": round(fsum / nf, 1), "WRITE_SIZE_KiB": round(wsum / nw, 1), "stage_launches": nf,
                          "traffic_bytes_per_launch": int(2 * fsum / nf * 1024 + wsum / nw * 1024)}
    try:
        bj = json.loads(open(os.path.join(d, f"bench_fetch_{wl}.json")).read().strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        bj = None
    workloads[wl] = {"kernels": kernels, "config": bj["config"]["workload"] if bj else None}
for tag in ("bench_stats_headline", "bench_stats", "bench_plain"):
    f = os.path.join(d, tag + "_detail.json")               # bench.py's full result (gpurun_out/bench_detail.json of that run); stdout has the short line
    if not os.path.exists(f):
        f = os.path.join(d, tag + ".json")
    if os.path.exists(f):
        try:
            j = json.loads(open(f).read().strip().splitlines()[-1])
            print(f"\n## {tag}: value {j['value']} {j['unit']}, ms/step {j['ms_per_step']}\n   roofline {j['roofline']}")
            for k, v in (j.get("stages") or j.get("stage_us") or {}).items():
                print("   ", k, v)
            if j.get("secondary"):
                print("   secondary:", j["secondary"].get("value"), j["secondary"].get("config"), j["secondary"].get("roofline"))
        except Exception as e:  # noqa: BLE001
            print(tag, "unreadable", e)

if len(sys.argv) > 2:
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hsh = hashlib.sha256()
    for fn in ("lrh_kernels.hip", "lrh_fft.hip.h", "lrh_kernels.hip.h", "lrh_host.hip", "lrh_timf2_sd.hip"):
        hsh.update(open(os.path.join(root, "linrad_amd", "csrc", fn), "rb").read())
    out = {"source_sha16": hsh.hexdigest()[:16],          # bench.py quotes these byte counts only for the library built from these sources
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes per workload, python3 bench.py --steps 5 "
                     "--warmup 2 --no-cpu [workload flags] (scripts/profile_round.sh)",
           "correction": "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE reports half of a coalesced streaming read, "
                         "MI355X_MICROARCH.md HBM section; calibrated in round 1 on k_sumsq's float2 reads: 67 MB reported for 134 MB read). "
                         "A stage = the kernels bench.py's HIP-event scope of that name covers; per stage launch = sum over those kernels of "
                         "(median per dispatch x dispatches) / dispatches of the stage's first kernel; median because the first calls of a run "
                         "(blanker start-up transient) move GBs that the steady state does not",
           "workloads": workloads}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
