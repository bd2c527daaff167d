"""Writes the drop-in measurement's input files and prints the harness command line of one gpu.fft1_batch_n, so that
oracle/_ref/shim_harness_hip can be started directly under rocprofv3 (the profiler must see the program itself, not a launcher):
  python3 scripts/glue_trace.py gpurun_out/glue_in 0 > gpurun_out/glue_in/cmd0.txt
  rocprofv3 --kernel-trace --stats -d gpurun_out/glue_prof0 -- $(cat gpurun_out/glue_in/cmd0.txt)
Same command as bench.py's glue_rate builds (configs[2] unless a third argument says c1)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from linrad_amd import lib as hiplib  # noqa: E402

out, n = sys.argv[1], int(sys.argv[2])
key = sys.argv[3] if len(sys.argv) > 3 else "c2"
w = bench.make_workload("c2", 14, 16, 12, 8) if key == "c2" else bench.make_workload("c1", 14, 12, 0, 0)
os.makedirs(out, exist_ok=True)
N1 = 1 << w["fft1_n"]
M1 = N1 // 2
ring_log2 = 24
s = hiplib.synth_defaults(N1, 0)
fi, fl, fo = (os.path.join(out, x) for x in ("in.bin", "lim.bin", "out.bin"))
if not os.path.exists(fi):
    np.asarray(hiplib.synth_iq(s, 0, (1 << ring_log2) // 4), np.int16).tofile(fi)
    bench.strong_liminfo(s, w["fft1_n"]).tofile(fl)
b = 1 << n
nblk = 8192 * (1 if b < 4 else 2 if b < 16 else 4)
cmd = [c for c in bench.ref_harness_cmd(w, nblk, fi, fl, fo) if not c.startswith(("max_fft1n", "max_fft2n"))]
cmd[0] = os.path.join(ROOT, "oracle", "_ref", "shim_harness_hip")
t2log = int(np.log2(max(8 << max(w["fft2_n"], w["fft1_n"]), 256 * M1)))
cmd += ["max_fft1n=256", "max_fft2n=64", f"timf2pow_log2={t2log}", f"timf1_log2={ring_log2}", "shim_threads=2", "shim_workers=3", f"shim_batch={b}", "warm=512", "shim_sparse=1"]
print(" ".join(cmd))
