"""GPU diagnostic: the stage-call pattern of integration/hipshim.c from Python, with its ingredients switched one at a time, against the
oracle: which of them makes timf2 differ?  usage: python scripts/shim_diag.py [case]"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from linrad_amd import abi                      # noqa: E402
from linrad_amd.lib import open_hip             # noqa: E402
from oracle_binding import open_oracle          # noqa: E402
from paritylib import load_golden               # noqa: E402
from refcases import case_params, lrh_config    # noqa: E402


def run(fn, d, g, max_batch, async_write, exports, nblk):
    iq = g["iq"]
    cfg = lrh_config(d, iq, max_batch=max_batch)
    rx = fn(cfg)
    host = np.ascontiguousarray(iq)
    if async_write:
        rx.host_register(host)
        rx.timf1_write_async(host)
    else:
        rx.timf1_write(host)
    rx.set_liminfo(g["liminfo"][: 1 << d["n1"]])
    rx.set_mix1_selfreq(d["fq"])
    for _ in range(nblk):
        rx.fft1_b(1), rx.fft1_c(1)
        if exports:
            rx.export(abi.RING_FFT1_SUMSQ, 0, 1 << d["n1"])
            rx.export(abi.RING_FFT1_SLOWSUM)
        rx.make_timf2(1)
        rx.first_noise_blanker()
        if exports:
            rx.blanker_state()
        for _ in range(rx.fft2_available()):
            rx.make_fft2(1)
            rx.fft2_mix1_fixed(1)
    out = rx.export(abi.RING_TIMF2_FLOAT), rx.p.timf2_pa
    if async_write:
        rx.timf1_write_wait()
        rx.host_unregister(host)
    rx.close()
    return out


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "n10_n12"
    d, g = case_params(name), load_golden(name)
    nblk = d["nblk"]
    ref, pa = run(open_oracle, d, g, 4, False, False, nblk)
    n1 = 1 << d["n1"]
    keep = np.ones(ref.size, bool)
    keep[(pa + np.arange(4 * (n1 // 2))) % keep.size] = False
    for mb, aw, ex in itertools.product((4, 1), (False, True), (False, True)):
        got, pa2 = run(open_hip, d, g, mb, aw, ex, nblk)
        err = np.linalg.norm((got - ref)[keep].astype(np.float64)) / np.linalg.norm(ref[keep].astype(np.float64))
        blk = 4 * (n1 // 2)
        e = ((got - ref) * keep).reshape(-1, blk).astype(np.float64)
        bad = np.nonzero(np.linalg.norm(e, axis=1) > 1e-4 * np.linalg.norm(ref.astype(np.float64)) / np.sqrt(e.shape[0]))[0]
        print(f"max_batch={mb} async_write={aw} exports={ex}: timf2 err {err:.3e} pa {pa2 == pa} bad blocks {bad[:16].tolist()} of {e.shape[0]}", flush=True)


if __name__ == "__main__":
    main()
