"""GPU diagnostic: oracle/_ref/shim_harness_hip on one golden case, timf2 error per block and per stream"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from paritylib import load_golden               # noqa: E402
from refcases import case_params, harness_args  # noqa: E402
from refdump import load_dump                   # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "n10_n12"
extra = sys.argv[2:]
d, g = case_params(name), load_golden(name)
td = tempfile.mkdtemp()
g["iq"].tofile(td + "/in.bin"); g["liminfo"].tofile(td + "/lim.bin")
for exe in ("shim_harness_hip", "shim_harness"):
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", exe)] + harness_args(d, td + "/in.bin", td + "/lim.bin", td + "/o.bin") + extra, capture_output=True, text=True)
    print(exe, r.returncode, r.stderr[-300:])
    dump = load_dump(td + "/o.bin")
    n1 = 1 << d["n1"]
    blk = n1 // 2
    for key in ("fft1_float", "timf2_float", "timf2_pwr_float", "fft2_float", "timf3_float"):
        a, b = dump[key].astype(np.float64), g[key][:dump[key].size].astype(np.float64)
        print(" ", key, "err", np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30), "norm", np.linalg.norm(b))
    a, b = dump["timf2_float"].reshape(-1, blk, 4).astype(np.float64), g["timf2_float"].reshape(-1, blk, 4).astype(np.float64)
    ew = np.linalg.norm((a - b)[:, :, :2], axis=(1, 2)); es = np.linalg.norm((a - b)[:, :, 2:], axis=(1, 2))
    nw = np.linalg.norm(b[:, :, :2], axis=(1, 2)); ns = np.linalg.norm(b[:, :, 2:], axis=(1, 2))
    print("  timf2_pa", dump["final"][3] // 4 // blk, "weak err/norm per block:", np.round(ew / np.maximum(nw, 1e-9), 4).tolist())
    print("  strong err/norm per block:", np.round(es / np.maximum(ns, 1e-9), 4).tolist())
    print("  liminfo nonzero", np.count_nonzero(g["liminfo"]))
