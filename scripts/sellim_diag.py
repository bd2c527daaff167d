"""Diagnostics for tests/test_gpu_random_configs.py::test_random_selective_limiter_matches_the_oracle: the given seeds HIP and oracle in lock step, the
limiter calls made by the caller; at the first round where the routing tables differ: which update did it, the bins, and how far the statistics the
two updates read were apart just before.  usage (GPU box): LRO_SELLIM_DEBUG=<bin> python3 scripts/sellim_diag.py seed [seed ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import refcases  # noqa: E402
import test_gpu_random_configs as T  # noqa: E402
from linrad_amd import abi  # noqa: E402
from linrad_amd.abi import default_sellim  # noqa: E402
from refcases import interleave, lrh_config, sellim_case  # noqa: E402


def rel(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


for seed in [int(x) for x in sys.argv[1:]]:
    t, how = T.random_sellim_case(seed)
    refcases.SELLIM["x"] = t
    d, sl, iq = sellim_case("x")
    batch = how["batch"]
    cfg = lrh_config(d, iq, max_batch=max(4, batch))
    N1, N2 = 1 << d["n1"], 1 << d["n2"]
    bt2 = float(np.float32(sl["blocktime"]) * np.float32(N2 - interleave(d["n2"], d["sinpow2"])) / np.float32(N1 - interleave(d["n1"], d["sinpow1"])))
    rxs = [T._open_hip(cfg), T._open_oracle(cfg)]
    pars = []
    for rx in rxs:
        rx.timf1_write(iq)
        rx.set_mix1_selfreq(how["fq"] * N2)
        pars.append(default_sellim(cfg, sellim_maxlevel=sl["maxlevel"], liminfo_group_points=max(1, N1 // sl["lim_groups"]), fft1_blocktime=sl["blocktime"],
                                   blanker_ston_fft1=sl["ston_fft1"], baseband_bw_fftxpts=sl["bw_fftxpts"], blanker_ston_fft2=sl["ston_fft2"], fft2_blocktime=bt2,
                                   exact_stats=1, fft1_first_point=4, fft1_first_inband=4, fft1_last_point=N1 - 5, fft1_last_inband=N1 - 5, **{f"sellim_par{i}": sl[f"par{i}"] for i in range(1, 9)}))
    print("seed", seed, {k: v for k, v in t.items() if k not in ("strong", "weak")}, how, flush=True)
    warm, c = 24 // batch, [[0, 0], [0, 0]]
    done = False
    for r in range(d["nblk"] // batch):
        for rx in rxs:
            rx.wideband_dsp(batch, batch)
        if r < warm:
            continue
        stats = [(rx.export(abi.RING_FFT1_SLOWSUM), rx.export(abi.RING_FFT1_SUMSQ), rx.export(abi.RING_FFT2_POWERSUM)) for rx in rxs]
        for which in (1, 2):
            if which == 2 and not sl["sellim2"]:
                continue
            sys.stderr.write(f"-- round {r} update {which}\n"); sys.stderr.flush()
            for k, rx in enumerate(rxs):
                cnt = rx.p.fft1_liminfo_cnt if which == 1 else rx.p.fft2_liminfo_cnt
                if cnt != c[k][which - 1]:
                    (rx.fft1_update_liminfo if which == 1 else rx.fft2_update_liminfo)(pars[k])
                    c[k][which - 1] = cnt
            a, b = rxs[0].get_liminfo(), rxs[1].get_liminfo()
            bad = np.nonzero(np.sign(a) != np.sign(b))[0]
            if bad.size:
                print(f"  round {r}: tables differ after update {which} (1 = fft1_update_liminfo, 2 = fft2_update_liminfo) at bins {bad[:12]}: hip {a[bad[:6]]} oracle {b[bad[:6]]}")
                print("  statistics just before: slowsum", rel(stats[0][0], stats[1][0]), "sumsq", rel(stats[0][1], stats[1][1]), "fft2 powersum", rel(stats[0][2], stats[1][2]))
                for i in bad[:4]:
                    nn = N2 // N1
                    print(f"   bin {i}: slowsum hip {stats[0][0][i]:.9g} oracle {stats[1][0][i]:.9g}; fft2 powersum of its sub-bins hip {stats[0][2][nn * i:nn * i + nn]} oracle {stats[1][2][nn * i:nn * i + nn]}")
                done = True
                break
        if done:
            break
    if not done:
        print("  no difference with the calls made by the caller")
    for rx in rxs:
        rx.close()
