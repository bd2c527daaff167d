"""Diagnostics for the waterfall lines of tests/test_gpu_random_configs.py::test_random_configuration_matches_the_oracle: HIP, oracle and the float64 build for
the given seeds, where the lines differ by more than a few counts.  usage (GPU box): python3 scripts/wf_diag.py seed [seed ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_gpu_random_configs as T  # noqa: E402
from paritylib import run_case  # noqa: E402
from refcases import make_input, make_liminfo  # noqa: E402

for seed in [int(x) for x in sys.argv[1:]]:
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
    t = run_case(T._open_truth, "random", golden=g, batch=batch, params=d)
    pre = np.array(t["api"].wf_pre_lines, np.float64).reshape(-1, a["cfg"].wf_xpixels)
    ha, hb, ht = a["wf_lines"].astype(int), b["wf_lines"].astype(int), t["wf_lines"].astype(int)
    print("seed", seed, {k: d[k] for k in ("n1", "n2", "sinpow1", "sinpow2", "wf_mode", "wf_avgnum", "stupid", "afc", "blockpower_block", "pulsewidth", "foldcorr_seed")}, "batch", batch, "lines", ha.shape)
    print("  first_xpoint", a["cfg"].wf_first_xpoint, "xpixels", a["cfg"].wf_xpixels, "itrace equal", np.array_equal(a["itrace"], b["itrace"]))
    for ln in range(min(ha.shape[0], 6)):
        dv = np.abs(ha[ln] - hb[ln])
        w = np.argsort(dv)[-5:][::-1]
        print(f"  line {ln}: max |hip - oracle| {dv.max()} at pixels {w}: hip {ha[ln][w]} oracle {hb[ln][w]} truth {ht[ln][w]} pre {np.round(pre[ln][w], 1)}; line max {hb[ln].max()} min {hb[ln].min()}; pixels differing by > 3: {int((dv > 3).sum())}")
    for key in ("fft2_powersum_float", "fft2_float", "timf2_float"):
        x, y = a[key].astype(np.float64), b[key].astype(np.float64)
        print(f"  {key} rel {np.linalg.norm(x - y) / np.linalg.norm(y):.2e}")
