"""Diagnostics: one seed of test_random_selective_limiter_matches_the_oracle run as the test runs it, then where the rings differ.
usage (GPU box): python3 scripts/sellim_diag2.py seed"""
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


seed = int(sys.argv[1])
t, how = T.random_sellim_case(seed)
refcases.SELLIM["x"] = t
d, sl, iq = sellim_case("x")
batch = how["batch"]
cfg = lrh_config(d, iq, max_batch=max(4, batch))
N1, N2 = 1 << d["n1"], 1 << d["n2"]
bt2 = float(np.float32(sl["blocktime"]) * np.float32(N2 - interleave(d["n2"], d["sinpow2"])) / np.float32(N1 - interleave(d["n1"], d["sinpow1"])))
erng = np.random.default_rng(6400 + seed)
ends = [4 + int(erng.integers(0, 5)), 0, 0, N1 - 5 - int(erng.integers(0, 5))]
ends[1], ends[2] = ends[0] + int(erng.integers(0, 9)), ends[3] - int(erng.integers(0, 9))
print("seed", seed, how, "ends", ends, {k: v for k, v in t.items() if k not in ("strong", "weak")})
warm, nr = 24 // batch, d["nblk"] // batch
rings = [(abi.RING_TIMF2_FLOAT, "timf2"), (abi.RING_TIMF2_PWR, "pwr"), (abi.RING_FFT2_FLOAT, "fft2"), (abi.RING_FFT2_POWERSUM, "ps2"), (abi.RING_TIMF3_FLOAT, "timf3"), (abi.RING_FFT1_SLOWSUM, "slowsum")]
res = []
for fn in (T._open_hip, T._open_oracle):
    rx = fn(cfg)
    rx.timf1_write(iq)
    rx.set_mix1_selfreq(how["fq"] * N2)
    par = default_sellim(cfg, sellim_maxlevel=sl["maxlevel"], liminfo_group_points=max(1, N1 // sl["lim_groups"]), fft1_blocktime=sl["blocktime"],
                         blanker_ston_fft1=sl["ston_fft1"], baseband_bw_fftxpts=sl["bw_fftxpts"], blanker_ston_fft2=sl["ston_fft2"], fft2_blocktime=bt2,
                         exact_stats=1, fft1_first_point=ends[0], fft1_first_inband=ends[1], fft1_last_inband=ends[2], fft1_last_point=ends[3],
                         **{f"sellim_par{i}": sl[f"par{i}"] for i in range(1, 9)})
    c1 = c2 = 0
    hist = []
    for r in range(nr):
        if r == warm and how["in_call"]:
            rx.wideband_limiter(par, bool(sl["sellim2"]))
        rx.wideband_dsp(batch, batch)
        if r >= warm and not how["in_call"]:
            if rx.p.fft1_liminfo_cnt != c1:
                rx.fft1_update_liminfo(par)
                c1 = rx.p.fft1_liminfo_cnt
            if sl["sellim2"] and rx.p.fft2_liminfo_cnt != c2:
                rx.fft2_update_liminfo(par)
                c2 = rx.p.fft2_liminfo_cnt
        st = rx.blanker_state()
        hist.append((rx.p.timf2_pa, rx.p.timf2_px, rx.p.timf3_pa, st.timf2_noise_floor, st.timf2_cleared_points, rx.p.fft2_na, rx.liminfo_amplitude_factor()))
    out = {k: rx.export(rg) for rg, k in rings}
    out["hist"], out["p"], out["lim"] = np.array(hist, np.float64), rx.p.as_dict(), rx.get_liminfo()
    res.append(out)
    rx.close()
h, o = res
print("pointers equal", h["p"] == o["p"], "tables: sign", int(np.sum(np.sign(h["lim"]) != np.sign(o["lim"]))), "values", rel(h["lim"], o["lim"]))
bad = np.nonzero((h["hist"] != o["hist"]).any(axis=1))[0]
print("per-round state (timf2_pa, timf2_px, timf3_pa, noise floor, cleared, fft2_na, amp factor) first differs at round", None if not bad.size else (int(bad[0]), h["hist"][bad[0]], o["hist"][bad[0]]))
for k in [x[1] for x in rings]:
    print(f"  {k:8s} rel {rel(h[k], o[k]):.3e}", "zero pattern differs at", int(np.sum((h[k] == 0) != (o[k] == 0))))
for k, chunk in (("timf2", 4 * (N1 // 2)), ("timf3", 2 * max(8, N2 >> d["mixred"])), ("fft2", 2 * N2)):
    a, b = h[k].reshape(-1, chunk).astype(np.float64), o[k].reshape(-1, chunk).astype(np.float64)
    e = np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-300)
    print(f"  {k} per block of {chunk}: worst {np.argsort(e)[-4:][::-1]} {np.sort(e)[-4:][::-1]}  median {np.median(e):.2e}")
f = np.nonzero((h["pwr"] == 0) != (o["pwr"] == 0))[0]
print("  cleared differently:", f[:10], "powers hip", h["pwr"][f[:6]], "oracle", o["pwr"][f[:6]])
