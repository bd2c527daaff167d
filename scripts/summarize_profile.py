#!/usr/bin/env python3
"""Condense a scripts/profile_round.sh output directory into the summary that is committed under profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]


def first(pattern):
    f = glob.glob(os.path.join(d, pattern), recursive=True)
    return f[0] if f else None


print("# rocprofv3 summary for", d)
f = first("stats/**/*kernel_stats.csv")
if f:
    print("\n## kernel-trace --stats (bench.py --steps 10 --warmup 3 --no-cpu)")
    print("%-58s %6s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for r in csv.DictReader(open(f)):
        print("%-58s %6s %12.1f %10.2f %6s" % (r["Name"][:58], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                               float(r["AverageNs"]) / 1e3, r["Percentage"]))
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = first(f"{tag}/**/*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == ctr:
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    print(f"\n## --pmc {ctr}: mean per dispatch (counter unit: KiB as reported by rocprofv3)")
    for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        print("%-58s n=%4d mean=%14.1f" % (k[:58], n, v / n))
for tag in ("bench_stats", "bench_plain"):
    f = os.path.join(d, tag + ".json")
    if os.path.exists(f):
        try:
            j = json.loads(open(f).read().strip().splitlines()[-1])
            print(f"\n## {tag}: value {j['value']} {j['unit']}, ms/step {j['ms_per_step']}, roofline {j['roofline']}")
            for k, v in j["stages"].items():
                print("   ", k, v)
        except Exception as e:  # noqa: BLE001
            print(tag, "unreadable", e)

# ---- machine-readable HBM traffic per launch (read by bench.py's roofline.traffic)
if len(sys.argv) > 2:
    per = {}
    for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        f = first(f"{tag}/**/*counter_collection.csv")
        if not f:
            continue
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == ctr:
                a = acc[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
        for k, (v, n) in acc.items():
            per.setdefault(k, {})[ctr + "_KiB"] = round(v / n, 1)
    stage_of = {"k_fft1": "fft1", "k_sumsq": "sumsq", "k_slowsum": "slowsum", "k_timf2": "timf2", "k_blank_scan": "blanker",
                "k_fft2<": "fft2", "k_powersum2": "powersum2", "k_mix1_back": "mix1"}
    kernels = {}
    for name, v in per.items():
        for pat, stage in stage_of.items():
            if pat in name and "FETCH_SIZE_KiB" in v and "WRITE_SIZE_KiB" in v:
                kernels[stage] = dict(kernel=name.split("(")[0] + "(...)", **v,
                                      traffic_bytes_per_launch=int(2 * v["FETCH_SIZE_KiB"] * 1024 + v["WRITE_SIZE_KiB"] * 1024))
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 10 --warmup 3 --no-cpu "
                     "(scripts/profile_round.sh)",
           "workload": {"fft1_n": 14, "fft2_n": 12, "batch": int(os.environ.get("LRH_PROFILE_BATCH", "4096"))},
           "correction": "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE reports half of a coalesced streaming "
                         "read, MI355X_MICROARCH.md HBM section; k_sumsq confirms it: ~67 MB reported for 134 MB read)",
           "kernels": kernels}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
