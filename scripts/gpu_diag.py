#!/usr/bin/env python3
"""Non-asserting GPU diagnostic: runs every parity case through the HIP path and prints per-ring errors."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from paritylib import RINGS, golden_itrace, load_golden, relerr, run_case  # noqa: E402
from refcases import CASES  # noqa: E402
from linrad_amd.lib import open_hip  # noqa: E402

for name in (sys.argv[1:] or list(CASES)):
    print("=====", name, flush=True)
    try:
        g = load_golden(name)
        for stupid in (0, None):
            out = run_case(open_hip, name, golden=g, stupid=stupid)
            tag = "noblank" if stupid == 0 else "blank"
            if stupid == 0:
                print(tag, "timf2", relerr(out["timf2_float"], g["timf2_float_noblank"]),
                      "pwr", relerr(out["timf2_pwr_float"], g["timf2_pwr_float_noblank"]))
                continue
            gi, oi = golden_itrace(g), out["itrace"]
            print("ptr trace equal:", np.array_equal(gi[:, [0, 1, 2, 3, 6, 7, 8, 9, 10]], oi[:, [0, 1, 2, 3, 6, 7, 8, 9, 10]]),
                  "floor maxdiff", np.abs(gi[:, 4] - oi[:, 4]).max(), "limit maxdiff", np.abs(gi[:, 5] - oi[:, 5]).max())
            if not np.array_equal(gi[:, 4], oi[:, 4]):
                bad = np.nonzero(gi[:, 4] != oi[:, 4])[0][:5]
                print("  floor first diffs", bad, gi[bad, 4], oi[bad, 4])
            for _, key in RINGS:
                print(f"  {key:22s} rel {relerr(out[key], g[key][:out[key].size]):.3e}")
            a, b = out["timf2_pwr_float"] == 0, g["timf2_pwr_float"] == 0
            print("  cleared: hip", a.sum(), "ref", b.sum(), "equal", np.array_equal(a, b), "jaccard", (a & b).sum() / max((a | b).sum(), 1))
            gw, ow = g["wf_lines"].reshape(-1, out["cfg"].wf_xpixels), out["wf_lines"]
            if gw.shape == ow.shape and gw.size:
                d = np.abs(gw.astype(int) - ow.astype(int))
                print("  waterfall lines", gw.shape, "mismatch frac", (d != 0).mean(), "max", d.max())
            else:
                print("  waterfall shape", gw.shape, ow.shape)
            gm, om = g["mixtrace"].reshape(-1, 8), out["mixtrace"]
            n = min(len(gm), len(om))
            print("  mixtrace maxdiff", np.abs(gm[:n] - om[:n]).max(axis=0) if n else None)
            bs = out["api"].blanker_state()
            print("  slow path calls", bs.slow_path_calls)
    except Exception:
        traceback.print_exc()
