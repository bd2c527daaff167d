#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace run: for the last rounds, start / end of every kernel relative to the first k_timf2 shown.
usage: python3 scripts/timeline.py <dir with *kernel_trace.csv> [rounds]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void lrh::", "").replace("lrh::", ""), r.get("Stream_Id", r.get("Queue_Id", "")))
        for r in csv.DictReader(open(f))]
rows.sort()
t2 = [i for i, r in enumerate(rows) if r[2].startswith("k_timf2<")]
first = t2[-(nr + 1)]
t0 = rows[first][0]
for s, e, k, q in rows[first:]:
    print("%9.1f %9.1f %8.1f  q%-4s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, k[:40]))
