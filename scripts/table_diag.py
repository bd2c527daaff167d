"""GPU box: which entries of the library's host tables differ from the goldens, with both values (diagnostics)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from paritylib import run_case, load_golden
from linrad_amd.lib import open_hip
from oracle_binding import open_oracle
for name in ("n8_n10", "n10_n12"):
    g = load_golden(name)
    for opener in (open_hip, open_oracle):
        out = run_case(opener, name, golden=g)
        api = out["api"]
        for t in ("fft1_filtercorr", "wg_waterf_yfac"):
            got = api.get_table(t, g[t].size); ref = g[t][:got.size]
            idx = np.nonzero(got != ref)[0]
            print(name, opener.__name__, t, idx.tolist(), [float.hex(float(v)) for v in got[idx]], [float.hex(float(v)) for v in ref[idx]])
