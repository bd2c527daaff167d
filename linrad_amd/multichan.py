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
