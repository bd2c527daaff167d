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


def test_read_backs_into_page_locked_destinations_and_lagged_stage_waits():
    """round 6: (a) a read-back whose destination lies in a span the caller has page-locked (lrh_host_register) is written by the copy engine
    itself -- same bytes as the staged path, any size; (b) lrh_export_begin / _end tickets of several generations stay valid while later
    stage calls are enqueued (the stage that rewrites the ring queues behind the copies still out); (c) lrh_stage_wait_lag returns for every
    lag 0..3, with and without that many calls behind it, and refuses a lag outside that"""
    import ctypes as C
    from linrad_amd import abi
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    from linrad_amd.workload import chain_config
    cfg = chain_config(fft1_n=12, fft2_n=10, batch=4)
    s = synth_defaults(1 << 12, 0)
    iq = synth_iq(s, 0, cfg.timf1_bytes // 4)
    rx = open_hip(cfg)
    lib = rx.lib
    lib.lrh_stage_wait_lag.argtypes, lib.lrh_stage_wait_lag.restype = [C.c_void_p, C.c_int, C.c_int], C.c_int
    lib.lrh_export_begin.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]
    lib.lrh_export_begin.restype = C.c_int
    lib.lrh_export_end.argtypes, lib.lrh_export_end.restype = [C.c_void_p, C.c_int], C.c_int
    lib.lrh_host_unregister.argtypes, lib.lrh_host_unregister.restype = [C.c_void_p, C.c_void_p], C.c_int
    for lag in range(4):
        assert lib.lrh_stage_wait_lag(rx.ctx, 0, lag) == 0          # nothing enqueued yet: returns at once
    assert lib.lrh_stage_wait_lag(rx.ctx, 0, 4) == abi.LRH_EINVAL and lib.lrh_stage_wait_lag(rx.ctx, 9, 0) == abi.LRH_EINVAL
    rx.timf1_write(iq)
    n1 = rx.N1
    rows, tickets = [], []
    pinned = np.zeros(cfg.fft1_sumsq_bufsize, np.float32)           # (a): one page-locked array takes the whole sumsq ring in one copy
    rx.host_register(pinned)
    for call in range(6):
        pa = rx.p.fft1_sumsq_pa
        rx.fft1_b(4)
        rx.fft1_c(4)
        rx.make_timf2(4)
        rx.first_noise_blanker()
        assert lib.lrh_stage_wait_lag(rx.ctx, 0, call % 4) == 0
        if rx.p.fft1_sumsq_pa != pa:                                # (b) a ticket per completed period, collected two calls later
            dst = np.zeros(n1, np.float32)
            t = C.c_int()
            assert lib.lrh_export_begin(rx.ctx, abi.RING_FFT1_SUMSQ, dst.ctypes.data_as(C.c_void_p), pa, n1, C.byref(t)) == 0
            rows.append((pa, dst)); tickets.append(t.value)
        while len(tickets) > 2:
            assert lib.lrh_export_end(rx.ctx, tickets.pop(0)) == 0
    for t in tickets:
        assert lib.lrh_export_end(rx.ctx, t) == 0
    full = rx.export(abi.RING_FFT1_SUMSQ)
    assert rows and np.any(full)
    t = C.c_int()
    assert lib.lrh_export_begin(rx.ctx, abi.RING_FFT1_SUMSQ, pinned.ctypes.data_as(C.c_void_p), 0, pinned.size, C.byref(t)) == 0 and t.value > 0
    assert lib.lrh_export_end(rx.ctx, t.value) == 0
    assert np.array_equal(pinned, full)
    for pa, dst in rows:                                            # (24 blocks: no row has been overwritten by a later lap)
        assert np.array_equal(dst, full[pa:pa + n1]) and np.any(dst)
    assert lib.lrh_host_unregister(rx.ctx, pinned.ctypes.data_as(C.c_void_p)) == 0
    rx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("merge_kb", ["0", "16", "256"])
def test_producer_running_ahead_block_by_block(merge_kb, monkeypatch):
    """round 6: lrh_timf1_write_async out of a page-locked arena notes its copies and issues them merged (LRH_IN_MERGE_KB), and a reader waits for the
    issue that carries its own samples.  A producer that hands over one block per call and runs a random distance ahead of the reader, three laps
    of a small timf1 ring, against the same stream written synchronously block by block: transforms and sums bit-identical"""
    import numpy as np
    from linrad_amd import abi
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    from linrad_amd.workload import chain_config
    monkeypatch.setenv("LRH_IN_MERGE_KB", merge_kb)
    cfg = chain_config(fft1_n=12, fft2_n=10, batch=16)
    cfg.timf1_bytes = 1 << 18
    nblk = 100
    rx = open_hip(cfg)
    bb, mask = rx.timf1_blockbytes, cfg.timf1_bytes - 1
    ring_blocks = cfg.timf1_bytes // bb
    per = bb // 2                                              # int16 values per block
    arena = np.ascontiguousarray(synth_iq(synth_defaults(1 << cfg.fft1_n, 0), 0, nblk * per // 2))
    assert arena.size == nblk * per and ring_blocks == 32
    px0 = rx.p.timf1p_px

    def result(r):
        out = (r.export(abi.RING_FFT1_FLOAT), r.export(abi.RING_FFT1_SUMSQ), r.export(abi.RING_FFT1_SLOWSUM), r.p.as_dict())
        r.close()
        return out
    for k in range(nblk):                                      # reference: block k on the device before its transform is asked for
        rx.timf1_write(arena[k * per:(k + 1) * per], (px0 + k * bb) & mask)
        rx.fft1_b(1), rx.fft1_c(1)
    ref = result(rx)
    rx = open_hip(cfg)
    rx.host_register(arena)
    rng = np.random.default_rng(5)
    written = consumed = 0
    while consumed < nblk:
        for _ in range(int(rng.integers(0, 13))):
            if written == nblk or written - consumed >= ring_blocks // 2:
                break
            rx.timf1_write_async(arena[written * per:(written + 1) * per], (px0 + written * bb) & mask)
            written += 1
        if written > consumed:
            n = int(rng.integers(1, min(4, written - consumed) + 1))
            rx.fft1_b(n), rx.fft1_c(n)
            consumed += n
    rx.timf1_write_wait()
    rx.host_unregister(arena)
    got = result(rx)
    assert got[3] == ref[3]
    for a, b, name in zip(got[:3], ref[:3], ("fft1_float", "fft1_sumsq", "fft1_slowsum")):
        assert np.array_equal(a, b), name
