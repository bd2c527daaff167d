#!/bin/bash
# kernel timeline of a few pipelined rounds (start offset, duration, queue) -- for schedule tuning
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/tl; mkdir -p gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 3 --warmup 2 --no-cpu "$@" > gpurun_out/tl/bench.json 2> gpurun_out/tl/log.txt
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
idx=[i for i,n in enumerate(names) if "k_timf2" in n]
i0=idx[10]; i1=idx[12]
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1+1]:
    n=r["Kernel_Name"].split("(")[0].replace("void lrh::","").replace("lrh::","")[:22]
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print("%-22s q%-3s start %8.1f end %8.1f dur %7.1f"%(n,r.get("Queue_Id","?"),s,e,e-s))
PY
find gpurun_out/tl -name "*kernel_trace.csv" -delete
