"""Multi-RF-channel sharding: one channel per rank / GPU, no data-path collective.

The only cross-channel step on the hot path is the per-bin channel power sum that fft1_c forms for two channels in one
loop (fft1.c:4132-4145: fft1_sumsq = |X_0|^2 + |X_1|^2); with channels sharded over ranks it is an all-reduce(sum) of
one fft1_sumsq block (N1 floats) per averaging period over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the CPU tests).
"""
import numpy as np


def channel_of_rank(rank):
    """RF channel handled by a rank (seed and sky phase of the synthetic signal follow it)."""
    return rank


def newest_sumsq_block(rx):
    """offset (in floats) of the most recently completed fft1_sumsq block of a StageAPI receiver"""
    return (rx.p.fft1_sumsq_pa - rx.N1) & (rx.cfg.fft1_sumsq_bufsize - 1)


def cross_channel_power_sum(block, dist):
    """all-reduce(sum) of one averaged power spectrum across the channel ranks; `block` is a torch tensor (in place)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(block)
    return block


def reference_two_channel_sumsq(spec0, spec1):
    """what fft1_c computes for two interleaved channels in one array (fft1.c:4139-4145), for tests"""
    p0 = spec0[0::2].astype(np.float32) ** 2 + spec0[1::2].astype(np.float32) ** 2
    p1 = spec1[0::2].astype(np.float32) ** 2 + spec1[1::2].astype(np.float32) ** 2
    return p0 + p1


# ---- two coupled RF channels (cfg.blanker_channels = 2): the blanker decides on the channel power sum -------------
class _DevSpan:
    """float32 device span as a __cuda_array_interface__ object, so that torch wraps it without a copy"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def _lrh_stream(rx, device):
    """the receiver's own HIP stream as a torch stream: a collective issued under it is ordered between the stage calls on the
    device (RCCL's stream waits for it and it waits for RCCL), with no host wait on either side"""
    import torch
    st = getattr(rx, "_torch_stream", None)
    if st is None:
        st = rx._torch_stream = torch.cuda.ExternalStream(rx.stream_handle(), device=device)
    return st


def exchange_sum(rx, which, count, dist, device=None):
    """all-reduce(sum) of one exchange buffer of a StageAPI receiver across the channel ranks.

    HIP receiver: in place on the device buffer (`lrh_exchange_ptr`), stream-ordered on the receiver's own stream.
    CPU oracle (gloo tests): through host memory."""
    import torch
    if count == 0:
        return
    if device is not None:
        t = torch.as_tensor(_DevSpan(rx.exchange_ptr(which), count), device=device)
        with torch.cuda.stream(_lrh_stream(rx, device)):
            dist.all_reduce(t)
    else:
        t = torch.from_numpy(rx.exchange_read(which, count))
        dist.all_reduce(t)
        rx.exchange_write(which, t.numpy())


def exchange_gather(rx, which, count, dist, device=None):
    """all-gather of the two `count`-float slots of an exchange buffer: slot r is the one rank r filled, afterwards every
    rank holds both.  HIP receiver: in place on the device buffer (RCCL recognises the input as its own slot of the
    output); CPU oracle (gloo tests): through host memory.  A group of one rank has nothing to fetch."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return
    assert world == 2, "Linrad has at most two RF channels (SURVEY F4)"
    if device is not None:
        t = torch.as_tensor(_DevSpan(rx.exchange_ptr(which), 2 * count), device=device)
        with torch.cuda.stream(_lrh_stream(rx, device)):
            if dist.get_backend() == "gloo":                # rehearsal on one GPU box: gloo has no in-place gather into one tensor
                slots = [torch.empty(count, dtype=torch.float32, device=device) for _ in range(2)]
                dist.all_gather(slots, t[rank * count:(rank + 1) * count].clone())
                t[(1 - rank) * count:(2 - rank) * count].copy_(slots[1 - rank])
            else:
                dist.all_gather_into_tensor(t, t[rank * count:(rank + 1) * count])
    else:
        own = torch.from_numpy(rx.exchange_read(which, count, rank * count))
        slots = [torch.empty_like(own), torch.empty_like(own)]
        dist.all_gather(slots, own)
        rx.exchange_write(which, slots[1 - rank].numpy(), (1 - rank) * count)


def coupled_fft2(rx, batch, dist, device=None):
    """make_fft2 of one of two coupled channels: the own transforms, then both channels' bins gathered for the cross
    products, fft2_xysum and the polarisation-independent waterfall line (fft2.c:1622-1640, 1700-1815)."""
    at = rx.ptrs_copy()
    rx.make_fft2(batch)
    n = rx.fft2_xy_begin(at, batch)
    exchange_gather(rx, rx.X_BINS, n, dist, device)
    rx.fft2_xy_finish(at, batch)


def coupled_fft3_mix2(rx, batch, dist, device=None):
    """fft3_mix2 of one of two coupled channels: the coherent combine A = c1 X + (c2 - j c3) Y, B = c1 Y - (c2 + j c3) X
    (mix2.c:340-343, 377-380) as an all-reduce of the channels' weighted bins; rank 0 then filters and back-transforms A
    (baseb_raw), rank 1 B (baseb_raw_orthog).  Needs rx.set_pol()."""
    n = rx.mix2_pol_begin(batch)
    exchange_sum(rx, rx.X_POL, n, dist, device)
    rx.fft3_mix2(batch)


def coupled_blanker(rx, dist, device=None):
    """first_noise_blanker of one of two coupled channels: power-sum exchange (with the linear blanker's tables installed also an
    all-gather of the channels' weak samples), scan, noise-statistic exchange, update (include/linrad_hip.h; blank1.c:984-992, 1017,
    1236-1300, 1510-1545, 1570)."""
    n = rx.blanker_begin()
    exchange_sum(rx, rx.X_PWR, n, dist, device)
    nw = rx.blanker_weak_span() if n else 0            # linear blanker: both channels' weak samples around the span (blank1.c:984-992)
    if nw:
        exchange_gather(rx, rx.X_WEAK, nw, dist, device)
    rx.first_noise_blanker()
    if n:
        exchange_sum(rx, rx.X_STAT, 2, dist, device)
        rx.blanker_finish()
    return n


def run_coupled(rx, nblocks, batch, dist, device=None, mix1=True, xy=False, pol=False):
    """single-CPU order of wideband_dsp (wcw.c:1036-1118) for one of two coupled channels, `batch` fft1 blocks per round,
    with the cross-channel exchanges between the stage calls (xy: also the fft2 cross products, an all-gather of the new
    transforms' bins)."""
    c = rx.cfg
    N2, M2 = rx.N2, rx.N2 - rx.fft2_interleave_points
    tmask = 4 * c.timf2pow_size - 1
    while nblocks > 0:
        b = min(batch, nblocks)
        rx.fft1_b(b), rx.fft1_c(b), rx.make_timf2(b)
        coupled_blanker(rx, dist, device)
        avail = (rx.p.timf2_pn2 - rx.p.timf2_px + 4 * c.timf2pow_size) & tmask        # wcw.c:265-266
        k = 0 if avail < 4 * N2 else 1 + (avail - 4 * N2) // (4 * M2)
        while k > 0:
            kb = min(k, c.max_fft2n)
            if xy:
                coupled_fft2(rx, kb, dist, device)
            else:
                rx.make_fft2(kb)
            if mix1:
                rx.fft2_mix1_fixed(kb)
                k3 = rx.fft3_available() if pol else 0
                while k3 > 0:
                    k3b = min(k3, rx.cfg.max_fft3n // 2)
                    rx.make_fft3_all(k3b)
                    coupled_fft3_mix2(rx, k3b, dist, device)
                    k3 -= k3b
            k -= kb
        nblocks -= b


XOP_SUM, XOP_GATHER = 0, 1


def install_exchange(rx, dist, device=None):
    """Two coupled channels through lrh_wideband_dsp: register the collectives the library asks for at its exchange points
    (include/linrad_hip.h, lrh_set_exchange).  HIP receiver (device given): the collective is issued under the stream the library
    names -- its own -- wrapped as a torch stream, in place on the library's device buffer, no host wait.  CPU oracle (gloo tests):
    the buffer is host memory."""
    import ctypes
    import torch
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    streams = {}

    def fn(which, op, ptr, count, stream, own=None):
        if world == 1:
            return 0                                            # nobody to talk to: the buffers already hold the sums / both slots are ours
        n = count if op == XOP_SUM else 2 * count
        if device is not None:
            st = streams.get(stream)
            if st is None:
                st = streams[stream] = torch.cuda.ExternalStream(stream, device=device)
            t = torch.as_tensor(_DevSpan(ptr, n), device=device)
            with torch.cuda.stream(st):
                if op == XOP_SUM:
                    dist.all_reduce(t)
                else:
                    # the rank's contribution: its slot (in-place all-gather), or -- `own` -- still where the library's previous stage
                    # left it (then the send buffer is that span and the own slot is merely overwritten with the same values)
                    mine = t[rank * count:(rank + 1) * count] if not own else torch.as_tensor(_DevSpan(own, count), device=device)
                    if dist.get_backend() == "gloo":
                        slots = [torch.empty(count, dtype=torch.float32, device=device) for _ in range(2)]
                        dist.all_gather(slots, mine.clone())
                        t[(1 - rank) * count:(2 - rank) * count].copy_(slots[1 - rank])
                    else:
                        dist.all_gather_into_tensor(t, mine)
            return 0
        buf = (ctypes.c_float * n).from_address(ptr)
        t = torch.frombuffer(buf, dtype=torch.float32)
        if op == XOP_SUM:
            dist.all_reduce(t)
        else:
            assert world == 2, "Linrad has at most two RF channels (SURVEY F4)"
            mine = (t[rank * count:(rank + 1) * count] if not own else torch.frombuffer((ctypes.c_float * count).from_address(own), dtype=torch.float32)).clone()
            slots = [torch.empty_like(mine), torch.empty_like(mine)]
            dist.all_gather(slots, mine)
            t[(1 - rank) * count:(2 - rank) * count] = slots[1 - rank]
        return 0
    rx.set_exchange(fn)


def install_pair_exchange(rxs, device=None):
    """Both channels of a coupled pair in ONE process (one context each, one caller thread each -- Linrad's two RF channels on one
    GPU): the exchange points of lrh_wideband_dsp meet at a barrier and trade through device memory (HIP, device given) or host
    memory (the CPU oracle).  Returns the barrier; run rxs[0].wideband_dsp and rxs[1].wideband_dsp on two threads."""
    import ctypes
    import threading
    import torch
    bar = threading.Barrier(2)
    box = [None, None]

    def make(ch):
        def fn(which, op, ptr, count, stream, own=None):
            n = count if op == XOP_SUM else 2 * count
            if device is not None:
                torch.cuda.ExternalStream(stream, device=device).synchronize()
                t = torch.as_tensor(_DevSpan(ptr, n), device=device)
                mine = torch.as_tensor(_DevSpan(own, count), device=device) if (own and op != XOP_SUM) else None
            else:
                t = torch.frombuffer((ctypes.c_float * n).from_address(ptr), dtype=torch.float32)
                mine = torch.frombuffer((ctypes.c_float * count).from_address(own), dtype=torch.float32) if (own and op != XOP_SUM) else None
            box[ch] = (t, mine)
            bar.wait()
            other, other_mine = box[1 - ch]
            lo, hi = (1 - ch) * count, (2 - ch) * count
            tmp = (box[0][0] + box[1][0]) if op == XOP_SUM else (other_mine.clone() if other_mine is not None else other[lo:hi].clone())
            if device is not None:
                torch.cuda.synchronize(device)
            bar.wait()
            if op == XOP_SUM:
                t.copy_(tmp)
            else:
                t[lo:hi].copy_(tmp)
            if device is not None:
                torch.cuda.synchronize(device)
            bar.wait()
            return 0
        return fn
    for ch, rx in enumerate(rxs):
        rx.set_exchange(make(ch))
    return bar
