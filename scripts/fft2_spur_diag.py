"""How clean is fft2 far from a strong carrier?  One carrier (no noise) through the chain, HIP / oracle / float64 build: the error of the fft2 spectrum against the
float64 one, in dB relative to the carrier, by distance and by residue of the bin index.  usage (GPU box): python3 scripts/fft2_spur_diag.py [n2 ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_gpu_random_configs as T  # noqa: E402
from paritylib import run_case  # noqa: E402
from refcases import case_params, level_gain, make_liminfo  # noqa: E402

for n2 in [int(x) for x in sys.argv[1:]] or [12, 14]:
    n1 = 12
    d = case_params("n10_n12")
    d.update(n1=n1, n2=n2, mixred=5, nblk=64, timf2pow_log2=max(n1, n2) + 3, sumsq_blocks=8, pulse_period=0, strong=[(300.3, 9000.0)], weak=[], sigma=0.0, blockpower_block=0, fq=0.3 * (1 << n2), stupid=0)
    d["gain"] = level_gain(n1, d["att_n"], 64.0)
    N1 = 1 << n1
    n = (N1 // 2) * d["nblk"] + 2 * N1
    t = np.arange(n)
    x = 9000.0 * np.exp(2j * np.pi * 300.3 * t / N1)
    iq = np.empty(2 * n, np.int16)
    iq[0::2], iq[1::2] = np.round(x.real), np.round(x.imag)
    g = {"iq": iq, "liminfo": make_liminfo(d)}
    res = {k: run_case(fn, "x", golden=g, params=d) for k, fn in (("hip", T._open_hip), ("oracle", T._open_oracle), ("truth", T._open_truth))}
    N2 = 1 << n2
    tr = res["truth"]["fft2_float"].astype(np.float64).reshape(-1, N2, 2)[-1]
    tz = tr[:, 0] + 1j * tr[:, 1]
    pk = int(np.argmax(np.abs(tz)))
    print(f"fft2_size {N2}: carrier at bin {pk}, |X| {np.abs(tz[pk]):.4g}")
    for k in ("hip", "oracle"):
        z = res[k]["fft2_float"].astype(np.float64).reshape(-1, N2, 2)[-1]
        e = np.abs((z[:, 0] + 1j * z[:, 1]) - tz) / np.abs(tz[pk])
        far = np.ones(N2, bool); far[max(0, pk - 64):pk + 64] = False
        edb = 20 * np.log10(np.maximum(e, 1e-30))
        worst = np.argsort(e * far)[-8:][::-1]
        print(f"  {k:6s} error / carrier: rms {20 * np.log10(np.sqrt(np.mean(e[far] ** 2))):.1f} dB, max {edb[far].max():.1f} dB at bins {worst} ({[int(w - pk) % 16 for w in worst]} mod 16 from the carrier)")
        by = [20 * np.log10(np.sqrt(np.mean(e[far & ((np.arange(N2) - pk) % 16 == r)] ** 2))) for r in range(16)]
        print("         rms by (bin - carrier) mod 16:", np.round(by, 1))
