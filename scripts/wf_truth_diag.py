"""GPU box: the HIP path, the reference golden and the float64 truth side by side (diagnostics): per ring the three relative errors, for the
waterfall the joint histogram of |hip - trunc(truth)| and |ref - trunc(truth)|."""
import sys, os, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from paritylib import run_case, load_golden, relerr, RINGS
from refcases import CASES
from oracle_binding import open_truth
from linrad_amd.lib import open_hip
for name in sys.argv[1:] or list(CASES):
    g = load_golden(name)
    try:
        t = run_case(open_truth, name, golden=g)
        h = run_case(open_hip, name, golden=g)
    except Exception as e:
        print(name, "ERROR", repr(e)); continue
    st = int(g["__stride"]) if "__stride" in g else 1
    print(name, "stride", st, "itrace truth==hip", np.array_equal(t["itrace"], h["itrace"]))
    pre = np.array(t["api"].wf_pre_lines)
    if pre.size:
        gw = g["wf_lines"].reshape(pre.shape).astype(int); hw = h["wf_lines"].astype(int); tt = np.clip(np.trunc(pre), -32767, 32767).astype(int)
        dh, dr = np.abs(hw - tt), np.abs(gw - tt)
        print("   wf bins %d  hip!=ref %d max %d | hip!=truth %d max %d | ref!=truth %d max %d | hist (|h-t|,|r-t|): %s" % (
            pre.size, np.count_nonzero(hw - gw), np.abs(hw - gw).max(), np.count_nonzero(dh), dh.max(), np.count_nonzero(dr), dr.max(),
            dict(sorted(collections.Counter(zip(dh.ravel().tolist(), dr.ravel().tolist())).items()))))
        worse = np.argwhere((dh > 1) & (dh > dr))
        for (l, i) in worse[:10]:
            print("      hip further than ref: line %d pix %d truth %.3f hip %d ref %d depth %d" % (l, i, pre[l, i], hw[l, i], gw[l, i], gw.max() - gw[l, i]))
    keys = [k for _, k in RINGS] + (["fft3", "baseb_raw"] if "fft3" in h else [])
    for k in keys:
        strided = st > 1 and k in ("fft1_float", "fft1_sumsq", "timf2_float", "timf2_pwr_float", "fft2_float", "fft2_power_float")
        b = g[k] if strided else g[k][:h[k].size]
        hh, tk = (h[k][::st], t[k][::st]) if strided else (h[k], t[k])
        cl = "" if k != "timf2_pwr_float" else "  cleared sets truth==ref %s hip==ref %s" % (np.array_equal(tk == 0, b == 0), np.array_equal(hh == 0, b == 0))
        print("   %-20s hip-vs-ref %.3e  hip-vs-truth %.3e  ref-vs-truth %.3e  ratio %.3f%s" % (k, relerr(hh, b), relerr(hh, tk), relerr(b, tk), relerr(hh, tk) / max(relerr(b, tk), 1e-30), cl))
