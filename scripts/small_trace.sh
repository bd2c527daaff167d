#!/bin/bash
# kernel trace of small calls: launches per call and device-side time (serial schedule, 4 blocks per call)
cd /tmp && export TMPDIR=/tmp
B=${B:-4}
rm -rf $GRAFT_REPO_ROOT/gpurun_out/small_trace
LRH_PIPELINE=${PL:-0} rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/small_trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-secondary --batch $B --rounds 1 --steps 400 --warmup 20 > $GRAFT_REPO_ROOT/gpurun_out/small_trace.json 2> $GRAFT_REPO_ROOT/gpurun_out/small_trace.log
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/small_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 100 calls: take the tail of the trace and count kernels between successive k_fft1 launches
names = [r["Kernel_Name"].split("(")[0][:60] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("void k_fft1<") or n.startswith("k_fft1")]
print("kernels in trace", len(rows), "k_fft1 launches", len(idx))
tail = idx[-101:]
per = [tail[i + 1] - tail[i] for i in range(len(tail) - 1)]
print("launches per call (last 100): min", min(per), "max", max(per), "mean", sum(per) / len(per))
a, b = tail[-2], tail[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r, n in zip(rows[a:b], names[a:b]):
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1000:9.1f} {(e-s)/1000:7.1f} us  {n}")
span = [int(rows[tail[i + 1]]["Start_Timestamp"]) - int(rows[tail[i]]["Start_Timestamp"]) for i in range(len(tail) - 1)]
busy = []
for i in range(len(tail) - 1):
    busy.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[tail[i]:tail[i + 1]]))
print("call period us: mean", sum(span) / len(span) / 1000, " device busy per call us:", sum(busy) / len(busy) / 1000)
PY
