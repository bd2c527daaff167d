#!/usr/bin/env python3
"""Two-RF-channel golden from the COMPILED REFERENCE (oracle/_ref/ref_harness channels=2): fft1_b (fft1win_dif_chan,
dif_permute_chan, channel-2 phasing), fft1_c (|X0|^2 + |X1|^2), update_fft1_slowsum, make_timf2 (fft1back_two,
two-channel fft1back_fp_finish).  Data only: the frame-interleaved int16 input {I0,Q0,I1,Q1}, liminfo, the reference's
output rings.  Runs only in the build container.

usage: python tests/golden/make_golden_2ch.py
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

from refcases import case_params, harness_args, make_input, make_liminfo, twochan_case  # noqa: E402
from refdump import load_dump  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    for name in ("twochan_n10", "twochan_n9_sin3", "twochan_real_n9"):
        d, frames, lim = twochan_case(name)
        with tempfile.TemporaryDirectory() as td:
            fi, fl, fo = (os.path.join(td, x) for x in ("in.bin", "lim.bin", "out.bin"))
            frames.tofile(fi)
            lim.tofile(fl)
            args = harness_args(d, fi, fl, fo) + ["channels=2", f"ch2_c1={d['ch2_c1']!r}", f"ch2_c2={d['ch2_c2']!r}"]
            subprocess.check_call([HARNESS] + args)
            ref = load_dump(fo)
            subprocess.check_call([HARNESS] + args + ["corr=1"])            # correlation spectrum on (genparm[FFT1_CORRELATION_SPECTRUM] = 1)
            refc1 = load_dump(fo)
            subprocess.check_call([HARNESS] + args + ["blanker2=1"])       # second run: two-channel first_noise_blanker after every block
            refb = load_dump(fo)
        out = {k: ref[k] for k in ("hdr", "fft1_filtercorr", "fft1_float", "fft1_sumsq", "fft1_slowsum", "timf2_float",
                                   "timf2_pwr_float", "itrace", "trace")}
        for k in ("timf2_float", "timf2_pwr_float", "itrace", "trace"):
            out["bln_" + k] = refb[k]
        for k in ("fft1_corrsum", "fft1_slowcorr", "fft1_slowcorr_tot", "slowcorr_tot_avgnum"):
            out[k] = refc1[k]
        assert np.array_equal(refc1["fft1_sumsq"], ref["fft1_sumsq"])      # the correlation sums ride beside the power sums, nothing else changes
        out["frames"], out["liminfo"] = frames, lim
        path = os.path.join(os.environ.get("LRH_GOLDEN_OUT", HERE), f"{name}.npz")
        if "--chain-only" not in sys.argv and ("--only" not in sys.argv or name in sys.argv):
            np.savez_compressed(path, **out)
            print(name, os.path.getsize(path) // 1024, "KiB")
        if d["real"]:
            continue
        # third run: the whole two-channel chain (two-channel first_noise_blanker, make_fft2 with fft2_xypower / fft2_xysum
        # and the polarisation-independent waterfall line, fft2_mix1_fixed); input = the same frames
        d, frames, lim = twochan_case(name, chain=True)
        with tempfile.TemporaryDirectory() as td:
            fi, fl, fo = (os.path.join(td, x) for x in ("in.bin", "lim.bin", "out.bin"))
            frames.tofile(fi)
            lim.tofile(fl)
            args = harness_args(d, fi, fl, fo) + ["channels=2", f"ch2_c1={d['ch2_c1']!r}", f"ch2_c2={d['ch2_c2']!r}", "chain2=1"]
            args += [f"pol_c{i + 1}={v!r}" for i, v in enumerate(d["pol"])]
            subprocess.check_call([HARNESS] + args)
            refc = load_dump(fo)
        outc = {k: refc[k] for k in ("hdr", "itrace", "trace", "fft2_float", "fft2_xypower", "fft2_xysum", "wf_lines", "timf3_float",
                                     "mixtrace", "final", "wg_waterf_yfac", "timf2_pwr_float", "fft3", "fft3_ptrs", "baseb_raw",
                                     "baseb_raw_orthog", "bg_filterfunc", "baseb_ptrs")}
        path = os.path.join(os.environ.get("LRH_GOLDEN_OUT", HERE), f"{name}_chain.npz")
        np.savez_compressed(path, **outc)
        print(name + "_chain", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
