"""diagnostic: time of one lrh_wideband_dsp call whose stupid blanker takes the serial walk (calibrated, limit below the noise), one-lane
form against the one-wave form:  python scripts/blank_serial_time.py [blocks]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linrad_amd.lib import open_hip
from linrad_amd.workload import chain_config

nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for one_lane in ("1", "0"):
    os.environ["LRH_BLN_SERIAL"] = one_lane
    cfg = chain_config(14, 12, batch=nblk)
    cfg.blanker_pulsewidth = 3
    cfg.timf2_noise_floor = 30
    n = cfg.timf1_bytes // 4
    rng = np.random.default_rng(9)
    iq = np.clip(np.round(rng.normal(0, 64.0, 2 * n)), -32767, 32767).astype(np.int16)
    rx = open_hip(cfg)
    rx.timf1_write(iq)
    rx.set_liminfo(np.zeros(1 << 14, np.float32))
    rx.sync()
    t0 = time.perf_counter()
    rx.wideband_dsp(nblk, nblk)
    rx.sync()
    dt = time.perf_counter() - t0
    st = rx.blanker_state()
    print("LRH_BLN_SERIAL=%s: %d blocks (%.1f M samples) %.1f ms, %.1f ns per sample, slow-path calls %d, cleared %d" %
          (one_lane, nblk, nblk * 8192 / 1e6, 1e3 * dt, 1e9 * dt / (nblk * 8192), st.slow_path_calls, st.timf2_cleared_points))
    rx.close()
