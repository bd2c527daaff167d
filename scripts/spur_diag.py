import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spurlib
from linrad_amd.lib import open_hip
from oracle_binding import open_oracle
name = sys.argv[1] if len(sys.argv) > 1 else "spur_n10_fft1"
g = spurlib.load(name)
ref = g["spur_trace"].reshape(-1, 12)
np.set_printoptions(precision=6, suppress=True, linewidth=220)
for fn in (open_oracle, open_hip):
    for acq in (False, True):
        out = spurlib.run(fn, name, g, acquire=acq)
        t = out["trace"]
        d = np.abs(t[:, 2:9] - ref[:, 2:9])
        print(fn.__name__, "acquire" if acq else "handed", "rows", t.shape, "first rows where ampl differs > 1e-4:", np.nonzero(d[:, 4] > 1e-4 * np.abs(ref[:, 6]))[0][:10])
        for i in range(4):
            print("   ", i, "got", t[i, :9], "\n       ref", ref[i, :9])
