#!/bin/bash
# quick per-kernel timing of the bench via rocprofv3 --stats; usage: scripts/kstats.sh [bench args]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/ks; mkdir -p gpurun_out/ks
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 bench.py --steps 10 --warmup 3 --no-cpu "$@" > gpurun_out/ks/bench.json 2> gpurun_out/ks/log.txt
f=$(find gpurun_out/ks -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-56s %5s %9.2f us %6s%%" % (r["Name"][:56], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
find gpurun_out/ks -name "*kernel_trace.csv" -delete
