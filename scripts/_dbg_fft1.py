import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from linrad_amd import abi
from linrad_amd.workload import chain_config
from linrad_amd.lib import open_hip, synth_defaults, synth_iq
from oracle_binding import open_oracle
N1 = 16384
for batch in (16, 300):
    cfg = chain_config(14, 12, batch=batch)
    s = synth_defaults(N1, 0)
    iq = synth_iq(s, 0, cfg.timf1_bytes // 4)
    ref = None
    for fn in (open_oracle, open_hip, open_hip, open_hip):
        rx = fn(cfg)
        rx.timf1_write(iq)
        rx.fft1_b(batch) if hasattr(rx, "fft1_b") else None
        f = rx.export(abi.RING_FFT1_FLOAT).reshape(-1, N1 * 2)[:batch]
        if ref is None:
            ref = f
            continue
        err = np.linalg.norm(f - ref, axis=1) / np.linalg.norm(ref, axis=1)
        bad = np.nonzero(err > 1e-5)[0]
        print("batch", batch, "max relerr %.3g" % err.max(), "bad transforms", bad[:20], len(bad))
        for t in bad[:3]:
            d = np.abs(f[t] - ref[t]).reshape(-1, 2).max(axis=1)
            idx = np.nonzero(d > 1e-3 * np.abs(ref[t]).max())[0]
            print("   t", t, "nbad bins", len(idx), idx[:16], "...", idx[-8:])
