"""GPU diagnostic: k_fft1v against k_fft1w on the bench-shape signal of tests/test_gpu_clever.py, blanker off: largest sample-wise
difference of the weak stream, and the values at given positions"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from linrad_amd import abi                                   # noqa: E402
from linrad_amd.lib import open_hip, synth_defaults, synth_iq  # noqa: E402
from linrad_amd.workload import chain_config, strong_liminfo  # noqa: E402

N1 = 16384
batch, warm = 4096, 64
cfg = chain_config(14, 16, batch=batch, fft3_n=12, mix2_n=8, rounds=2)
cfg.stupid_bln_mode = 0
s = synth_defaults(N1, 0)
iq = synth_iq(s, 0, cfg.timf1_bytes // 4)
lim = strong_liminfo(s, 14)
res = []
for v in ("1", "0"):
    os.environ["LRH_FFT1V"] = v
    rx = open_hip(cfg)
    rx.timf1_write(iq)
    rx.set_liminfo(lim)
    rx.set_mix1_selfreq(0.31 * 65536 + 0.3)
    rx.wideband_dsp(warm, warm)
    rx.wideband_dsp(batch, batch)
    res.append(rx.export(abi.RING_TIMF2_FLOAT).reshape(-1, 4))
    rx.close()
a, b = res
d = np.abs(a[:, :2] - b[:, :2]).max(axis=1)
print("weak: max abs diff", d.max(), "at", int(d.argmax()), "rms", np.sqrt(np.mean(d * d)), "rms of signal", np.sqrt(np.mean(b[:, :2] ** 2)))
worst = np.argsort(d)[-8:]
for i in worst:
    print(int(i), a[i, :2], b[i, :2])
print("strong max abs diff", np.abs(a[:, 2:] - b[:, 2:]).max())
