#!/usr/bin/env python3
"""Generate tests/golden/rawdat_18bit.npz from the COMPILED REFERENCE's 18-bit packing (csplit.c, via
oracle/_ref/ref_rawdat).  Data only: seeded int32 samples, the reference's packed bytes for them, and the reference's
expansion of those bytes into a ring image.  Runs only in the build container."""
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
TOOL = os.path.join(ROOT, "oracle", "_ref", "ref_rawdat")


def main():
    if not os.path.exists(TOOL):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    rng = np.random.default_rng(20)
    n = 4 * 1024                                           # int32 components (two 8192-byte blocks)
    v = rng.integers(-131072, 131072, n).astype(np.int64)
    samples = ((v << 14) | rng.integers(0, 1 << 14, n)).astype(np.int32)   # garbage below bit 14 must be dropped
    ring_log2, pa = 15, 16384
    with tempfile.TemporaryDirectory() as td:
        fs, fp, fe = (os.path.join(td, x) for x in ("s.bin", "p.bin", "e.bin"))
        samples.tofile(fs)
        subprocess.check_call([TOOL, "compress", f"in={fs}", f"out={fp}"])
        packed = np.fromfile(fp, np.uint8)
        subprocess.check_call([TOOL, "expand", f"in={fp}", f"out={fe}", f"pa={pa}", f"ring_log2={ring_log2}"])
        ring = np.fromfile(fe, np.uint8)
    path = os.path.join(os.environ.get("LRH_GOLDEN_OUT", HERE), "rawdat_18bit.npz")
    np.savez_compressed(path, samples=samples, packed=packed, ring=ring, pa=np.array(pa), ring_log2=np.array(ring_log2))
    print(path, os.path.getsize(path) // 1024, "KiB", packed.size, "packed bytes")


if __name__ == "__main__":
    main()
