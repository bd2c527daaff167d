#!/usr/bin/env python3
"""Goldens of the linear ("clever") noise blanker on TWO channels from the COMPILED REFERENCE (harness channels=2 blanker2=1 clever=1):
first_noise_blanker with get_pulse_pol, transform_timf2_pol and subtract_twochan_pulse (blank1.c:232-609, 984-992) after every
block.  Data only: seeded frames, the tables as init_blanker built them, per-call scalars, the two-channel timf2 rings.
usage: python tests/golden/make_golden_clever2.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from refcases import CLEVER2, clever2_case, harness_args  # noqa: E402
from refdump import load_dump  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
KEEP = ["bln", "bln_ints", "bln_fparams", "blanker_refpulse", "blanker_phasefunc", "blanker_pulindex", "timf2_float", "timf2_pwr_float", "trace", "itrace"]


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    for name in sys.argv[1:] or list(CLEVER2):
        d, cl, frames, lim, des = clever2_case(name)
        with tempfile.TemporaryDirectory() as td:
            fi, fl, fd, fo = (os.path.join(td, x) for x in ("in.bin", "lim.bin", "des.bin", "out.bin"))
            frames.tofile(fi)
            lim.tofile(fl)
            des.tofile(fd)
            args = harness_args(d, fi, fl, fo) + ["channels=2", "ch2_c1=1.0", "ch2_c2=0.0", "blanker2=1", "clever=1", f"desired={fd}", f"clever_factor={cl['clever_factor']}"]
            r = subprocess.run([HARNESS] + args, stderr=subprocess.PIPE, text=True)
            assert r.returncode == 0, r.stderr
            ref = load_dump(fo)
        out = {k: ref[k] for k in KEEP}
        out["frames"], out["liminfo"], out["desired"] = frames, lim, des
        path = os.path.join(os.environ.get("LRH_GOLDEN_OUT", HERE), f"{name}.npz")
        np.savez_compressed(path, **out)
        tr = ref["trace"].reshape(-1, 16)
        it = ref["itrace"].reshape(-1, 16)
        print(name, os.path.getsize(path) // 1024, "KiB;", r.stderr.strip().splitlines()[0][:150])
        print("   fitted per period", tr[:, 10].tolist()[::4], "clever rate", sorted(set(np.round(tr[:, 9], 3)))[-3:], "limit", sorted(set(tr[:, 8]))[:4])
        print("   cleared", it[:, 5].tolist()[::8], "noise floor", it[:, 12].tolist()[::8])


if __name__ == "__main__":
    main()
