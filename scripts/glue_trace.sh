#!/bin/bash
# GPU box: rocprofv3 kernel trace of the timed glue harness at one fft1_b batch size (arguments: batch workers), stats into gpurun_out/glue_trace/
B=${1:-1}; W=${2:-3}
mkdir -p gpurun_out/glue_trace
python - <<PY
import sys, os, numpy as np
sys.path.insert(0, ".")
import bench
from linrad_amd import lib as hiplib
from linrad_amd.workload import strong_liminfo
N1 = 1 << 14
s = hiplib.synth_defaults(N1, 0)
np.asarray(hiplib.synth_iq(s, 0, (1 << 24) // 4), np.int16).tofile("/tmp/g_in.bin")
strong_liminfo(s, 14).tofile("/tmp/g_lim.bin")
w = bench.make_workload("c2", 14, 16, 12, 8)
cmd = [c for c in bench.ref_harness_cmd(w, 4096, "/tmp/g_in.bin", "/tmp/g_lim.bin", "/tmp/g_out.bin") if not c.startswith(("max_fft1n", "max_fft2n"))]
cmd[0] = "oracle/_ref/shim_harness_hip"
cmd += ["max_fft1n=256", "max_fft2n=64", "timf2pow_log2=21", "timf1_log2=24", "shim_threads=2", "shim_workers=$W", "shim_batch=$B", "warm=512"]
open("/tmp/g_cmd.txt", "w").write(" ".join(cmd))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/glue_trace -o glue -- $GRAFT_REPO_ROOT/$(cat /tmp/g_cmd.txt | cut -d' ' -f1) $(cat /tmp/g_cmd.txt | cut -d' ' -f2-) > $GRAFT_REPO_ROOT/gpurun_out/glue_trace/run.log 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/glue_trace | head
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/glue_trace/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:25]:
        print("%-60s calls %7s avg %9.1f us tot %8.2f ms %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
t = glob.glob("gpurun_out/glue_trace/**/*kernel_trace.csv", recursive=True)
if t:
    rows = list(csv.DictReader(open(t[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[len(rows) // 2]["Start_Timestamp"])
    print("--- 120 launches from the middle of the run: start us, dur us, queue, kernel")
    for r in rows[len(rows) // 2: len(rows) // 2 + 120]:
        print("%9.1f %7.1f q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:50]))
PY
tail -2 gpurun_out/glue_trace/run.log
