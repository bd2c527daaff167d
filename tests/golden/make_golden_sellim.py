#!/usr/bin/env python3
"""Goldens of the selective limiter from the COMPILED REFERENCE (harness sellim=1: fft1_update_liminfo of sellim.c:738 runs in
the single-CPU order of wcw.c:1124-1128 and make_timf2 routes with its table).  Data only: the seeded input, the limiter's
parameters, liminfo after every update, and the rings the routing shapes.  usage: python tests/golden/make_golden_sellim.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from refcases import SELLIM, harness_args, sellim_case  # noqa: E402
from refdump import load_dump  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
KEEP = ["liminfo_trace", "liminfo_trace_blk", "sellim_params", "sellim_fparams", "liminfo_final", "fft1_sumsq", "fft1_slowsum",
        "timf2_float", "timf2_pwr_float", "fft2_powersum_float", "timf3_float", "itrace", "trace", "final", "wg_waterf_yfac",
        "amp_factor_trace", "liminfo_trace2", "liminfo_trace2_blk", "sellim2_fparams", "sellim2_par1"]


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    for name in sys.argv[1:] or list(SELLIM):
        d, sl, iq = sellim_case(name)
        with tempfile.TemporaryDirectory() as td:
            fi, fo = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
            iq.tofile(fi)
            args = [a for a in harness_args(d, fi, "none", fo) if not a.startswith("liminfo=")]
            subprocess.check_call([HARNESS] + args + ["sellim=1"] + [f"{k}={v}" for k, v in sl.items()])
            ref = load_dump(fo)
        out = {k: ref[k] for k in KEEP if k in ref}
        out["iq"] = iq
        path = os.path.join(os.environ.get("LRH_GOLDEN_OUT", HERE), f"{name}.npz")
        np.savez_compressed(path, **out)
        if "sellim2_par1" in out and int(out["sellim2_par1"][0]) == 0:
            # the median variant decides by a strict compare: no decision of the case may sit within float32 noise of its threshold
            # (the float32 oracle, which follows the reference to 1e-6, measures the margins)
            import sellimlib
            from oracle_binding import open_oracle
            m32 = []
            sellimlib.run(open_oracle, name, dict(np.load(path)), margins=m32)
            # calls made while the fft2 sums are still those of an all but empty input (median far below its steady value) work on the float32
            # noise floor: two float32 implementations were seen 4e-4 apart there, against 1e-7 in the steady state
            steady = np.median([x[1] for x in m32[len(m32) // 2:]])
            ok = [x[0] > (1.5e-3 if x[1] < 1e-3 * steady else 2e-5) for x in m32]
            print("   decision margins", ["%.1e" % x[0] for x in m32[:10]], "medians", ["%.1e" % x[1] for x in m32[:6]], "all clear:", all(ok), "min %.1e" % min(x[0] for x in m32))
            assert all(ok), "a decision of this case sits within float32 noise of its threshold: change the case (tests/refcases.py) and regenerate"
        tr = out["liminfo_trace"].reshape(-1, 1 << d["n1"])
        if "liminfo_trace2" in out:
            t2 = out["liminfo_trace2"].reshape(-1, 1 << d["n1"])
            print("   fft2_update_liminfo:", t2.shape[0], "updates; strong bins:", [int(np.count_nonzero(r)) for r in t2[::max(1, t2.shape[0] // 8)]],
                  "differs from the table before it in", [int(np.count_nonzero(t2[i] != t2[i - 1])) for i in range(1, t2.shape[0], max(1, t2.shape[0] // 8))],
                  "amp factors", sorted(set(np.round(out["amp_factor_trace"], 4).tolist()))[:6])
        print(name, os.path.getsize(path) // 1024, "KiB;", tr.shape[0], "updates; strong bins per update:",
              [int(np.count_nonzero(r)) for r in tr[::max(1, tr.shape[0] // 8)]], "attenuated:", int(np.count_nonzero(tr[-1] > 0)))


if __name__ == "__main__":
    main()
