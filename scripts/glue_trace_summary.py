"""Summary of scripts/glue_trace.sh's traces: per kernel count / total / mean, the device's busy time (union of the kernel and copy
intervals) against the span from the first to the last kernel.  usage: python3 scripts/glue_trace_summary.py gpurun_out/glue_prof0"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], "k", r.get("Stream_Id", r.get("Queue_Id", ""))))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", ""), "c", ""))
rows.sort()
ks = [r for r in rows if r[3] == "k"]
t0, t1 = ks[0][0], max(r[1] for r in ks)
# skip the warm-up: the last 80 % of the span
cut = t0 + (t1 - t0) // 5
per = defaultdict(lambda: [0, 0])
busy = 0
end = cut
kb = 0
kend = cut
for s, e, n, kind, q in rows:
    if s < cut:
        continue
    per[n][0] += 1
    per[n][1] += e - s
    if e > end:
        busy += e - max(s, end)
        end = e
    if kind == "k" and e > kend:
        kb += e - max(s, kend)
        kend = e
span = t1 - cut
print(f"span {span / 1e6:.2f} ms (last 80 % of the run)  device busy {busy / 1e6:.2f} ms = {busy / span:.2f}  kernels alone {kb / span:.2f}")
print(f"{'name':60s} {'calls':>7s} {'total ms':>9s} {'mean us':>8s} {'of span':>7s}")
for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:60s} {c:7d} {t / 1e6:9.2f} {t / c / 1e3:8.2f} {t / span:7.3f}")
print("launches per ms:", round(sum(c for c, _ in per.values()) / (span / 1e6), 1))
