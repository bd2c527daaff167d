#!/bin/bash
# per-kernel times in the serial schedule (LRH_PIPELINE=0: stand-alone kernels); usage: scripts/serial_kstats.sh [bench args]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp LRH_PIPELINE=0
rm -rf gpurun_out/ks0; mkdir -p gpurun_out/ks0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks0 -- python3 bench.py --steps 4 --warmup 2 --no-cpu --no-secondary --rounds 1 "$@" > gpurun_out/ks0/bench.json 2> gpurun_out/ks0/log.txt
f=$(find gpurun_out/ks0 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-56s %5s %9.2f us %6s%%" % (r["Name"].replace("void lrh::","").replace("lrh::","")[:56], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
find gpurun_out/ks0 -name "*kernel_trace.csv" -delete
