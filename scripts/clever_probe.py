"""diagnostic: the linear blanker's state call by call on the bench signal (bench.py --clever), HIP library.
   python scripts/clever_probe.py fft1_n fft2_n batch calls [noise_floor [rounds [fft3_n mix2_n]]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from linrad_amd import lib as hiplib
from linrad_amd.workload import chain_config
f1, f2, batch, calls = (int(v) for v in sys.argv[1:5])
nf = int(sys.argv[5]) if len(sys.argv) > 5 else 500
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 1
f3, m2 = (int(sys.argv[7]), int(sys.argv[8])) if len(sys.argv) > 8 else (0, 0)
cfg = chain_config(f1, f2, batch=batch, device=0, fft3_n=f3, mix2_n=m2, rounds=rounds)
cfg.fft1_float_sparse = cfg.fft2_float_sparse = int(os.environ.get("PROBE_SPARSE", "0"))
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "clever_n10_n12.npz")))
cfg.blanker_pulsewidth, cfg.blnfit_range = int(g["bln_ints"][1]), int(g["bln_ints"][3])
cfg.timf2_noise_floor = nf
rx = bench.setup_receiver(cfg, 0, hiplib.open_hip, hiplib)
bi, bf = g["bln_ints"], g["bln_fparams"]
if not os.environ.get("PROBE_NOCLEVER"):
  rx.set_blanker_tables(bln=g["bln"].reshape(-1, 4)[:, :3], refpulse=g["blanker_refpulse"], phasefunc=g["blanker_phasefunc"], pulindex=g["blanker_pulindex"],
                      largest_blnfit=int(bi[2]), clever_bln_factor=float(bf[1]), clever_bln_limit=int(np.float32(nf) * np.float32(bf[1])), liminfo_amplitude_factor=float(bf[0]))
if os.environ.get("PROBE_SELLIM"):
    from linrad_amd.abi import default_sellim
    M1, N2 = (1 << f1) // 2, 1 << f2
    sel = default_sellim(cfg, fft1_blocktime=M1 / 160e6, blanker_ston_fft1=30.0, exact_stats=0, blanker_ston_fft2=30.0, fft2_blocktime=(N2 // 2) / 160e6)
    rx.wideband_limiter(sel, int(os.environ["PROBE_SELLIM"]) > 1)
for k in range(calls):
    t = time.time()
    rx.wideband_dsp(batch * rounds, batch)
    rx.sync()
    bs = rx.blanker_state()
    print(batch, k, "floor", bs.timf2_noise_floor, "climit", bs.clever_bln_limit, "slimit", bs.stupid_bln_limit, "fitted", bs.last_call_fitted, "rejected", bs.last_call_rejected,
          "cleared", bs.last_call_cleared, "lowlevel", round(rx.p.fft1_lowlevel_fraction, 4), "despiked", [round(v, 2) for v in bs.timf2_despiked_pwr], "serial", bs.clever_serial_calls,
          "amp_factor", rx.liminfo_amplitude_factor(), "strong bins", int(np.count_nonzero(rx.get_liminfo())), "%.3fs" % (time.time() - t), flush=True)
    if bs.last_call_rejected > 50 * max(1, bs.last_call_fitted) + 1000 or (bs.clever_bln_limit < 2000 and not os.environ.get("PROBE_NOCLEVER")):
        print("limit has collapsed: stopping before the next call"); break
