#!/bin/bash
# kernel trace of small calls: launches per call and device-side time (serial schedule, 4 blocks per call)
cd /tmp && export TMPDIR=/tmp
B=${B:-4}
rm -rf $GRAFT_REPO_ROOT/gpurun_out/small_trace
LRH_PIPELINE=${PL:-0} rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/small_trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-secondary --batch $B --rounds 1 --steps 400 --warmup 20 > $GRAFT_REPO_ROOT/gpurun_out/small_trace.json 2> $GRAFT_REPO_ROOT/gpurun_out/small_trace.log
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/small_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("void ", "").replace("lrh::", "").split("(")[0][:50] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("k_fft1<")]
tail = idx[-101:]
per = [tail[i + 1] - tail[i] for i in range(len(tail) - 1)]
print("launches per call (last 100 calls): min", min(per), "max", max(per), "mean", sum(per) / len(per))
for i in range(len(tail) - 1, 0, -1):          # a call that holds a limiter update and a whole narrowband tail
    seg = names[tail[i - 1]:tail[i]]
    if any(n.startswith("k_sellim") for n in seg) and any(n.startswith("k_fft3") for n in seg):
        break
a, b = tail[i - 1], tail[i]
t0 = int(rows[a]["Start_Timestamp"])
print(" start us   dur us  kernel")
for r, n in zip(rows[a:b], names[a:b]):
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1000:9.1f} {(e-s)/1000:7.1f}  {n}  grid {r['Grid_Size_X']} wg {r['Workgroup_Size_X']}")
span = [int(rows[tail[i + 1]]["Start_Timestamp"]) - int(rows[tail[i]]["Start_Timestamp"]) for i in range(len(tail) - 1)]
busy = [sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[tail[i]:tail[i + 1]]) for i in range(len(tail) - 1)]
print("call period us (mean):", round(sum(span) / len(span) / 1000, 1), " sum of kernel durations per call us:", round(sum(busy) / len(busy) / 1000, 1))
PY
