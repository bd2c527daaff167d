import sys
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import test_gpu_random_configs as t
from paritylib import run_case
from refcases import make_input, make_liminfo
for s in [int(x) for x in sys.argv[1:]]:
    d, batch = t.random_case(s)
    g = {"iq": make_input(d), "liminfo": make_liminfo(d)}
    a = run_case(t._open_hip, "random", golden=g, batch=batch, params=d)
    b = run_case(t._open_oracle, "random", golden=g, batch=batch, params=d)
    print("seed", s, {k: d[k] for k in ("n1", "n2", "sinpow1", "sinpow2", "mixred", "nblk", "dword", "direction", "wf_mode", "wf_avgnum", "att_n", "avg1num", "bln_interval", "bln_avgnum")}, "batch", batch)
    print("  floors hip", a["itrace"][:, 4].tolist()[:20]); print("  floors ora", b["itrace"][:, 4].tolist()[:20])
    wa, wb = a["wf_lines"].astype(int), b["wf_lines"].astype(int)
    print("  wf shape", wa.shape, "maxdiff", np.abs(wa - wb).max() if wa.size else None)
    if wa.size:
        bad = np.argwhere(np.abs(wa - wb) > 2)
        print("  bad count", len(bad), "first", bad[:6].tolist())
        for (l, x) in bad[:6]:
            print("    line", l, "pix", x, "hip", wa[l, x], "ora", wb[l, x], "neighbours hip", wa[l, max(0, x - 2):x + 3].tolist(), "ora", wb[l, max(0, x - 2):x + 3].tolist())
        print("  cfg wf", a["cfg"].wf_xpixels, a["cfg"].wf_mode, a["cfg"].wf_first_xpoint, a["cfg"].waterfall_avgnum)
