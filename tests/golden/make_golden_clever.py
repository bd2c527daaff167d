#!/usr/bin/env python3
"""Goldens of the linear ("clever") noise blanker from the COMPILED REFERENCE (harness clever=1): init_blanker (buf.c:1771) builds
the pulse-response tables from a synthetic amplitude calibration, first_noise_blanker (blank1.c:684-1003) then finds, fits and
subtracts pulses (subtract_onechan_pulse, blank1.c:36).  Data only: seeded input, the tables as the reference built them, the
per-call blanker scalars and the rings behind.  usage: python tests/golden/make_golden_clever.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from refcases import CLEVER, clever_case, harness_args  # noqa: E402
from refdump import load_dump  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
KEEP = ["bln", "bln_ints", "bln_fparams", "blanker_refpulse", "blanker_phasefunc", "blanker_pulindex", "timf2_float", "timf2_pwr_float",
        "fft2_float", "fft2_powersum_float", "timf3_float", "trace", "itrace", "mixtrace", "final", "wf_lines",
        "liminfo_trace", "liminfo_trace_blk", "sellim_params", "sellim_fparams", "amp_factor_trace"]


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    for name in sys.argv[1:] or list(CLEVER):
        d, cl, iq, lim, des = clever_case(name)
        with tempfile.TemporaryDirectory() as td:
            fi, fl, fd, fo = (os.path.join(td, x) for x in ("in.bin", "lim.bin", "des.bin", "out.bin"))
            iq.tofile(fi)
            lim.tofile(fl)
            des.tofile(fd)
            args = harness_args(d, fi, fl, fo)
            if cl["sellim"]:                 # the limiter builds the table: no initial liminfo file
                args = [a for a in args if not a.startswith("liminfo=")] + ["sellim=1"] + [f"{k}={v}" for k, v in cl["sellim"].items()]
            r = subprocess.run([HARNESS] + args + ["clever=1", f"desired={fd}", f"clever_factor={cl['clever_factor']}"],
                               stderr=subprocess.PIPE, text=True)
            assert r.returncode == 0, r.stderr
            ref = load_dump(fo)
        out = {k: ref[k] for k in KEEP if k in ref}
        out["iq"], out["liminfo"], out["desired"] = iq, lim, des
        path = os.path.join(os.environ.get("LRH_GOLDEN_OUT", HERE), f"{name}.npz")
        np.savez_compressed(path, **out)
        tr = ref["trace"].reshape(-1, 16)
        it = ref["itrace"].reshape(-1, 16)
        print(name, os.path.getsize(path) // 1024, "KiB;", r.stderr.strip().splitlines()[0][:150])
        print("   bln", ref["bln"].reshape(-1, 4)[:, :3].tolist())
        print("   fitted per period", tr[:, 8].tolist()[::4], "clever rate", sorted(set(np.round(tr[:, 7], 3)))[-3:], "limit", sorted(set(tr[:, 6]))[:4])
        if "amp_factor_trace" in out:
            print("   amplitude factors", sorted(set(np.round(out["amp_factor_trace"], 4).tolist()))[:8])
        print("   cleared", it[:, 5].tolist()[::8], "pfit-pend lag", sorted(set(((it[:, 0] // 4 - it[:, 1]) & (ref['timf2_pwr_float'].size - 1)).tolist()))[:6])


if __name__ == "__main__":
    main()
