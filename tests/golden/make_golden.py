#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/ref_harness).

Runs only in the build container (needs /root/reference to build the harness: `make -C oracle ref`).
Each fixture is data only: the seeded int16 IQ input, the liminfo routing table, and the reference's
output rings / scalar traces for the call pattern of tests/refcases.py.  No reference source is stored.

usage: python tests/golden/make_golden.py [case ...]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from refcases import CASES, case_params, harness_args, make_foldcorr, make_input, make_liminfo  # noqa: E402
from refdump import load_dump  # noqa: E402

BIG_RINGS = ["fft1_float", "fft1_sumsq", "timf2_float", "timf2_pwr_float", "fft2_float", "fft2_power_float",
             "timf2_float_noblank", "timf2_pwr_float_noblank"]
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
KEEP = ["hdr", "fft1_window", "fft1_filtercorr", "fft2_window", "mix1_fqwin", "wg_waterf_yfac",
        "fft1_inverted_window", "fft1_first_raw", "fft1_float", "fft1_sumsq", "fft1_slowsum", "timf2_float",
        "timf2_pwr_float", "fft2_float", "fft2_power_float", "fft2_powersum_float", "timf3_float", "wf_lines",
        "trace", "itrace", "mixtrace", "final", "timf2_blockpower", "blockpower_ptrs", "fft3", "fft3_window",
        "fft3_ptrs", "baseb_raw", "bg_filterfunc", "baseb_ptrs", "basebraw_fir", "timf3_py", "afc_fq0", "afc_supplied", "afc_fq_mid", "afc_fq_slope", "afc_fq_curv", "afc_fq_start"]


def run_case(name, **override):
    d = case_params(name)
    d.update(override)
    iq, lim = make_input(d), make_liminfo(d)
    with tempfile.TemporaryDirectory() as td:
        fi, fl, fo = (os.path.join(td, x) for x in ("in.bin", "lim.bin", "out.bin"))
        iq.tofile(fi)
        lim.tofile(fl)
        extra = []
        if d["foldcorr_seed"]:
            ff = os.path.join(td, "fold.bin")
            make_foldcorr(d).tofile(ff)
            extra = [f"foldcorr={ff}"]
        subprocess.check_call([HARNESS] + harness_args(d, fi, fl, fo) + extra)
        ref = load_dump(fo)
    return d, iq, lim, ref


def main():
    if not os.path.exists(HARNESS):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    names = sys.argv[1:] or list(CASES)
    for name in names:
        d, iq, lim, ref = run_case(name)
        # second run with the blanker off: timf2 ring straight out of make_timf2
        _, _, _, ref_nb = run_case(name, stupid=0)
        out = {k: ref[k] for k in KEEP if k in ref}
        out["timf2_float_noblank"] = ref_nb["timf2_float"]
        out["timf2_pwr_float_noblank"] = ref_nb["timf2_pwr_float"]
        stride = d.get("golden_stride", 1)
        if stride > 1:
            # large case: keep every stride-th element of the big rings (the comparison subsamples the same way)
            for k in BIG_RINGS:
                out[k] = np.ascontiguousarray(out[k][::stride])
            out["__stride"] = np.array(stride)
        out["iq"] = iq
        out["liminfo"] = lim
        if d["foldcorr_seed"]:
            out["foldcorr"] = make_foldcorr(d)
        path = os.path.join(os.environ.get("LRH_GOLDEN_OUT", HERE), f"{name}.npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
