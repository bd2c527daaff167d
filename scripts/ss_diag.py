import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spurlib
from linrad_amd.lib import open_hip
from oracle_binding import open_oracle
for name in ("spur_n10_n12",):
    g = spurlib.load(name)
    for fn in (open_oracle, open_hip):
        out = spurlib.run(fn, name, g)
        sp, thr, done, cnt = out["ss"]
        a, b = out["ss_range"]
        ref = g["spursearch_spectrum"][a:b + 1]
        print(fn.__name__, "thr", thr, "ref thr", g["spursearch_thresholds"], "done", done, "cnt", cnt, g["spursearch_info"], "at", g["spursearch_at"])
        print("   max", sp.max(), ref.max(), "neg", int((sp < 0).sum()), int((ref < 0).sum()), "zeros", int((sp == 0).sum()), int((ref == 0).sum()), "maxerr/scale", np.abs(sp - ref).max() / ref.max())
