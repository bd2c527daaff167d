import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import test_gpu_random_configs as T
from paritylib import run_case
from refcases import make_input, make_liminfo
seed = int(sys.argv[1])
d, batch = T.random_case(seed)
g = {"iq": make_input(d), "liminfo": make_liminfo(d)}
if d["foldcorr_seed"]:
    from refcases import make_foldcorr
    g["foldcorr"] = make_foldcorr(d)
if d["afc"]:
    r3 = np.random.default_rng(5600 + seed)
    tt = np.arange(64 * d["nblk"] + 64)
    amp, per, at, step = r3.uniform(0.3, 2.0), r3.uniform(15, 60), int(r3.integers(10, 40)), r3.uniform(-3, 3)
    f = (d["fq"] + amp * np.sin(2 * np.pi * tt / per) + step * ((tt >= at) & (tt < at + 30))).astype(np.float32)
    g["afc_fq0"], g["afc_supplied"] = f[:1], f[1:]
a = run_case(T._open_hip, "random", golden=g, batch=batch, params=d)
b = run_case(T._open_oracle, "random", golden=g, batch=batch, params=d)
print({k: d[k] for k in ("n1", "n2", "sinpow1", "sinpow2", "mixred", "nblk", "stupid", "afc", "blockpower_block", "pulsewidth", "foldcorr_seed", "real", "dword", "direction", "sample_shift", "pulse_period", "att_n")}, "batch", batch)
x, y = a["timf2_float"].astype(np.float64), b["timf2_float"].astype(np.float64)
dd = np.abs(x - y)
w = np.argsort(dd)[-8:][::-1]
print("timf2 rel", np.linalg.norm(x - y) / np.linalg.norm(y), "largest diffs at", w, "hip", x[w], "oracle", y[w])
pw_a, pw_b = a["timf2_pwr_float"], b["timf2_pwr_float"]
f = np.nonzero((pw_a == 0) != (pw_b == 0))[0]
print("flips", f, "pwr hip", pw_a[f], "oracle", pw_b[f], "floor", a["itrace"][-1, 4], b["itrace"][-1, 4], "timf2_pa", a["api"].p.timf2_pa, "N1", a["api"].N1)
blk = dd.reshape(-1, 4 * (a["api"].N1 // 2)).max(axis=1)
print("blocks with a difference > 1e-3 of the rms:", np.nonzero(blk > 1e-3 * np.sqrt(np.mean(y * y)))[0], "of", blk.size)
