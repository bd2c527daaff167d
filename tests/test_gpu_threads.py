"""GPU: the C ABI called from Linrad's thread topology (examples/lrh_threads.c: input thread, wideband dispatcher, six fft1_b
workers with their own handles, timf2 / second-fft / narrowband threads, pthread condition events like lxsys.c:415-447) against
the same stage calls made from one thread: every ring, the pointer block and the blanker state bit for bit."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, threaded, nblk, extra=()):
    exe = os.path.join(ROOT, "examples", "lrh_threads")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "linrad_amd", "csrc"), "example"])   # gcc only; no-op when fresh
    out = tmp_path / f"dump_{threaded}.bin"
    r = subprocess.run([exe, str(threaded), str(nblk), str(out), *map(str, extra)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    return np.fromfile(out, np.uint8), r.stdout


@pytest.mark.parametrize("sizes", [(12, 13, 6), (10, 12, 3)])
def test_stage_threads_equal_the_single_thread_run(tmp_path, sizes):
    a, la = _run(tmp_path, 0, 700, sizes)
    b, lb = _run(tmp_path, 1, 700, sizes)
    assert a.size == b.size and a.size > 1 << 20
    assert np.count_nonzero(a) > a.size // 4
    bad = np.nonzero(a != b)[0]
    assert bad.size == 0, (la, lb, bad[:8])


def test_worker_handles_from_python_threads():
    """lrh_fft1_b with handles 1..6 called from six Python threads at once (ctypes releases the GIL inside the call); the
    transforms must equal those of handle 0, and the next reader waits for all of them"""
    import threading
    from linrad_amd import abi
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    from linrad_amd.workload import chain_config
    cfg = chain_config(fft1_n=12, fft2_n=10, batch=4)
    cfg.max_fft1n = 64
    s = synth_defaults(1 << 12, 0)
    iq = synth_iq(s, 0, cfg.timf1_bytes // 4)
    ref = open_hip(cfg)
    ref.timf1_write(iq)
    for _ in range(12):
        ref.fft1_b(4)
    ref.fft1_c(4)
    want = ref.export(abi.RING_FFT1_FLOAT)
    rx = open_hip(cfg)
    rx.timf1_write(iq)
    M1b, blk = rx.timf1_blockbytes, 2 * rx.N1
    errs = []

    def worker(h):
        try:
            for j in range(2):                              # worker h takes batches h-1 and h+5
                b = (h - 1) + 6 * j
                rc = rx.lib.lrh_fft1_b(rx.ctx, h, (b * 4 * M1b) & (cfg.timf1_bytes - 1), (b * 4 * blk) & (cfg.max_fft1n * blk - 1), 4)
                if rc:
                    errs.append((h, rc))
        except Exception as e:  # noqa: BLE001
            errs.append((h, repr(e)))
    th = [threading.Thread(target=worker, args=(h,)) for h in range(1, 7)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    rx.p.timf1p_px, rx.p.fft1_pa = ref.p.timf1p_px, ref.p.fft1_pa
    rx.p.fft1_na, rx.p.fft1_nm = ref.p.fft1_na, ref.p.fft1_nm
    rx.fft1_c(4)                                            # a reader of fft1_float: joins the worker streams
    assert np.array_equal(rx.export(abi.RING_FFT1_FLOAT), want)
    assert rx.lib.lrh_fft1_b(rx.ctx, 7, 0, 0, 1) == abi.LRH_EINVAL
