import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from linrad_amd import abi
from linrad_amd.workload import chain_config, strong_liminfo
from linrad_amd.lib import open_hip, synth_defaults, synth_iq
from oracle_binding import open_oracle
N1 = 16384
cfg = chain_config(14, 12, batch=16)
s = synth_defaults(N1, 0)
iq = synth_iq(s, 0, cfg.timf1_bytes // 4)
lim = strong_liminfo(s, 14)
res = []
for fn in (open_hip, open_oracle):
    rx = fn(cfg)
    rx.timf1_write(iq); rx.set_liminfo(lim); rx.set_mix1_selfreq(0.31 * 4096 + 0.3)
    rx.wideband_dsp(48, 16)
    res.append((rx.export(abi.RING_TIMF2_PWR), rx.blanker_state(), rx.export(abi.RING_FFT1_FLOAT)))
(hp, hb, hf), (op, ob, of) = res
print("limit", hb.stupid_bln_limit, ob.stupid_bln_limit, "floor", hb.timf2_noise_floor, ob.timf2_noise_floor)
print("fft1 relerr", np.linalg.norm(hf - of) / np.linalg.norm(of))
d = np.abs(hp - op)
bad = np.nonzero(d > 1e-3 * np.maximum(op, 1))[0]
print("pwr mismatches", len(bad), bad[:40])
for i in bad[:10]:
    print(i, hp[i - 2:i + 3], op[i - 2:i + 3])
