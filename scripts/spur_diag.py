"""Diagnostics for tests/test_gpu_random_configs.py::test_random_spur_is_acquired_and_tracked_like_the_oracle: where in the fft2 ring HIP and the oracle differ.
usage (GPU box): python3 scripts/spur_diag.py seed [seed ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import refcases  # noqa: E402
import spurlib  # noqa: E402
import test_gpu_random_configs as T  # noqa: E402

for seed in [int(x) for x in sys.argv[1:]]:
    t, batch = T.random_spur_case(seed)
    refcases.SPUR["x"] = t
    case = refcases.spur_case("x")
    gg = spurlib.load("spur_n10_n12")
    st = np.zeros(16); st[10] = t["spur_speknum"]
    g = {"iq": case[2], "spur_init_state": st, "spur_locked": np.array([t["spur_start"]]), "spur_spectra": gg["spur_spectra"]}
    h = spurlib.run(T._open_hip, "x", g, batch=batch, acquire=True, case=case)
    o = spurlib.run(T._open_oracle, "x", g, batch=batch, acquire=True, case=case)
    cfg = h["cfg"]
    n2 = 1 << cfg.fft2_n
    fh, fo = h["fft2"].reshape(cfg.max_fft2n, n2, 2).astype(np.float64), o["fft2"].reshape(cfg.max_fft2n, n2, 2).astype(np.float64)
    e = np.sqrt(((fh - fo) ** 2).sum(axis=2))
    print("seed", seed, t["tone"], "speknum", t["spur_speknum"], "start", t["spur_start"], "batch", batch, "locations", sorted(set(o["trace"][:, 0].astype(int))), "fft2_na", h["api"].p.fft2_na)
    print("  loop state, last transform: hip", h["trace"][-1, :9], "oracle", o["trace"][-1, :9])
    print("  phase difference per transform (mrad):", np.round(1e3 * ((h["trace"][:, 3] - o["trace"][:, 3] + np.pi) % (2 * np.pi) - np.pi), 2)[:40])
    rows = np.argsort(e.sum(axis=1))[-4:][::-1]
    for r in rows:
        b = np.argsort(e[r])[-6:][::-1]
        print(f"  ring row {r}: error norm {np.linalg.norm(e[r]):.4g} of row norm {np.linalg.norm(fo[r]):.4g}; largest at bins {b}: {np.round(e[r][b], 3)}")
