"""GPU box: the library's host-built tables against the compiled reference's (goldens), element by element (diagnostics)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from paritylib import run_case, load_golden
from refcases import CASES
from linrad_amd.lib import open_hip
for name in CASES:
    g = load_golden(name)
    out = run_case(open_hip, name, golden=g)
    api = out["api"]
    row = []
    for t in ("fft1_window", "fft2_window", "mix1_fqwin", "fft1_filtercorr", "wg_waterf_yfac", "fft1_inverted_window", "fft3_window"):
        if t not in g or (t == "fft2_window" and out["cfg"].fft2_sinpow == 0):
            continue
        try:
            got = api.get_table(t, g[t].size)
        except Exception as e:
            row.append(f"{t}: {e}"); continue
        ref = g[t][:got.size]
        bad = got != ref
        ulp = np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
        row.append(f"{t}: {int(bad.sum())}/{got.size} differ, max {int(ulp.max()) if ulp.size else 0} ulp")
    print(name, "|", " | ".join(row))
