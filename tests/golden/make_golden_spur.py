#!/usr/bin/env python3
"""Goldens of the spur subtraction from the COMPILED REFERENCE (harness spur=1): the reference's own store_new_spur /
spur_phase_lock acquire a carrier, eliminate_spurs (spur.c:36-494) then tracks and subtracts it inside make_fft2.  Data only:
seeded input, the loop state and histories at hand-over, the line-shape table, the PLL state after every transform and the rings
behind it.  usage: python tests/golden/make_golden_spur.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from refcases import SPUR, SPUR_CLICKS, harness_args, make_liminfo, spur_case  # noqa: E402
from refdump import load_dump  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
KEEP = ["spur_init_state", "spur_init_table", "spur_init_signal", "spur_init_ind", "spur_spectra", "spur_trace", "spur_locked",
        "fft2_float", "fft2_powersum_float", "timf3_float", "wf_lines", "mixtrace", "final", "itrace", "fft1_float", "fft1_sumsq", "fft1_slowsum",
        "spursearch_spectrum", "spursearch_thresholds", "spursearch_info", "spursearch_at", "spur_trace_all", "spur_clicks"]


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    for name in sys.argv[1:] or list(SPUR) + list(SPUR_CLICKS):
        d, sp, iq, lim = spur_case(name)
        with tempfile.TemporaryDirectory() as td:
            fi, fl, fo = (os.path.join(td, x) for x in ("in.bin", "lim.bin", "out.bin"))
            iq.tofile(fi)
            lim.tofile(fl)
            r = subprocess.run([HARNESS] + harness_args(d, fi, fl, fo) + ["spur=1"] + [f"{k}={v}" for k, v in sp.items()], stderr=subprocess.PIPE, text=True)
            assert r.returncode == 0, r.stderr
            ref = load_dump(fo)
        assert ref["spur_locked"][0] > 0, "the reference did not lock the spur: " + r.stderr
        out = {k: ref[k] for k in KEEP if k in ref}
        out["iq"], out["liminfo"] = iq, lim
        if iq.size > (1 << 20):                       # (the seeded input of the four-step case is 2 MB: tests/refcases.py regenerates it, the golden keeps a checksum)
            del out["iq"]
            out["iq_sum"] = np.array([int(iq.astype(np.int64).sum()), int((iq.astype(np.int64) * (np.arange(iq.size) % 251)).sum())], np.int64)
        path = os.path.join(os.environ.get("LRH_GOLDEN_OUT", HERE), f"{name}.npz")
        np.savez_compressed(path, **out)
        if name in SPUR_CLICKS:                       # the operator's clicks through init_spur_elimination: every spur's loop state after every transform
            ta = out["spur_trace_all"].reshape(-1, 40)
            print(name, os.path.getsize(path) // 1024, "KiB; transforms", ta.shape[0], "spurs held", sorted(set(ta[:, 1].astype(int))))
            for ln in r.stderr.strip().splitlines():
                if ln.startswith("click"):
                    print("   " + ln[:200])
            continue
        tr = out["spur_trace"].reshape(-1, 12)
        print(name, os.path.getsize(path) // 1024, "KiB;", r.stderr.strip().splitlines()[-1][:150])
        print("   transforms tracked", tr.shape[0], "flags", sorted(set(tr[:, 1].astype(int))), "locations", sorted(set(tr[:, 0].astype(int))),
              "freq %.3f -> %.3f" % (tr[0, 2], tr[-1, 2]), "ampl %.4g -> %.4g" % (tr[0, 6], tr[-1, 6]))

if __name__ == "__main__":
    main()
