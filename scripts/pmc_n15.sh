#!/bin/bash
# HBM traffic of the four-step fft1 / timf2 kernels at fft1_size 32768 (FETCH_SIZE and WRITE_SIZE in separate passes, serial schedule)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp LRH_PIPELINE=0
OUT=gpurun_out/pmc_n15; rm -rf $OUT; mkdir -p $OUT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$ctr -- python3 bench.py --no-cpu --no-secondary --no-sellim --fft1-n 15 --fft2-n 17 --batch 2048 --steps 3 --warmup 1 > $OUT/bench_$ctr.json 2> $OUT/$ctr.log
done
python3 - <<'PY'
import csv, glob, collections, statistics
res = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_n15/{ctr}/**/*counter_collection.csv", recursive=True)[0]
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr: continue
        vals[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lrh::", "")[:40]].append(float(r["Counter_Value"]))
    for k, v in vals.items(): res[k][ctr] = statistics.median(v)
print("kernel, median KB fetched (raw counter; x2 for the gfx950 correction), median KB written, MB moved (2 x fetch + write)")
for k, v in sorted(res.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)))[:12]:
    fe, wr = v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)
    print(f"{k:42s} {fe:12.0f} {wr:12.0f} {(2 * fe + wr) / 1024:10.1f}")
PY
find $OUT -name "*.csv" -size +4M -delete
