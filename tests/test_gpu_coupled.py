"""GPU: the collective path of the coupled two-channel blanker.  One GPU is all a test box has, so the group has one rank:
what is checked is the plumbing -- RCCL reducing the library's own device buffers in place (lrh_exchange_ptr wrapped as a
torch tensor) gives the same bits as the exchange through host memory, and batched rounds work."""
import os
import socket

import numpy as np
import pytest

from linrad_amd import abi
from refcases import lrh_config, twochan_case

pytestmark = pytest.mark.gpu


def test_device_exchange_equals_host_exchange():
    import torch
    import torch.distributed as dist
    from linrad_amd.lib import open_hip
    from linrad_amd.multichan import run_coupled
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("cpu:gloo,cuda:nccl", rank=0, world_size=1)
    try:
        d, frames, lim = twochan_case("twochan_n10")
        iq = np.ascontiguousarray(frames.reshape(-1, 4)[:, 0:2]).ravel()
        res = []
        for device in (torch.device("cuda:0"), None):
            cfg = lrh_config(d, iq, blanker_channels=2, timf1_channel_index=0)
            rx = open_hip(cfg)
            rx.timf1_write(iq)
            rx.set_liminfo(lim)
            rx.set_mix1_selfreq(2200.3)
            run_coupled(rx, d["nblk"], 4, dist, device=device)
            bs = rx.blanker_state()
            res.append((rx.export(abi.RING_TIMF2_FLOAT), rx.export(abi.RING_FFT2_FLOAT), rx.export(abi.RING_TIMF3_FLOAT),
                        bs.timf2_noise_floor, bs.stupid_bln_limit, rx.p.as_dict()))
        for a, b in zip(res[0][:3], res[1][:3]):
            assert np.array_equal(a, b)
        assert res[0][3:] == res[1][3:]
        assert np.count_nonzero(res[0][2]) > 0 and (res[0][0].reshape(-1, 4)[:, :2] == 0).all(axis=1).sum() > 50
    finally:
        dist.destroy_process_group()


def test_gather_exchange_and_batched_cross_products():
    """Plumbing of multichan.coupled_fft2 on the device path with a one-rank group: lrh_fft2_xy_begin leaves the batch's
    transforms in the own slot of LRH_X_BINS, the gather leaves it alone, lrh_fft2_xy_finish forms x2 from it (the partner
    slot was never filled, so every product with y is zero).  Parity of the products: tests/test_twochan.py."""
    import torch
    import torch.distributed as dist
    from linrad_amd.lib import open_hip
    from linrad_amd.multichan import coupled_fft2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("cpu:gloo,cuda:nccl", rank=0, world_size=1)
    try:
        d, frames, lim = twochan_case("twochan_n10", chain=True)
        iq = np.ascontiguousarray(frames.reshape(-1, 4)[:, 0:2]).ravel()
        cfg = lrh_config(d, iq, blanker_channels=2, timf1_channel_index=0)
        rx = open_hip(cfg)
        rx.timf1_write(iq)
        rx.set_liminfo(lim)
        for _ in range(24):
            rx.fft1_b(1), rx.fft1_c(1), rx.make_timf2(1)
        rx.p.timf2_pn2 = rx.p.timf2_pa                      # release everything (no blanker here: plumbing test)
        k = rx.fft2_available()
        assert k >= 3
        coupled_fft2(rx, 3, dist, torch.device("cuda:0"))
        own = rx.exchange_read(rx.X_BINS, 3 * 2 * rx.N2)
        assert np.array_equal(own.reshape(3, -1), rx.export(abi.RING_FFT2_FLOAT).reshape(-1, 2 * rx.N2)[:3])
        xyp = rx.export(abi.RING_FFT2_XYPOWER).reshape(-1, rx.N2, 4)[:3]
        f = own.reshape(3, rx.N2, 2).astype(np.float32)
        assert np.allclose(xyp[:, :, 0], f[:, :, 0] ** 2 + f[:, :, 1] ** 2, rtol=1e-6)
        assert np.count_nonzero(xyp[:, :, 1:]) == 0          # the partner slot was never filled: y = 0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch,nblk,calls,fuse", [(32, 128, 3, "0"), (4096, 8192, 1, "0")])
def test_coupled_wideband_dsp_equals_the_stage_calls(batch, nblk, calls, fuse):
    """(second case: the call shape bench.py --coupled times -- 2 x 4096 blocks, sparse rings)
    lrh_wideband_dsp with cfg.blanker_channels = 2 and the exchange function registered (lrh_set_exchange): one call enqueues what
    multichan.run_coupled does with stage calls and collectives in between -- same rings to float32 rounding (one-rank group: the collectives
    have nobody to talk to, the sequencing and the fused kernels are what is compared), at the sizes of BASELINE configs[3]"""
    import torch
    import torch.distributed as dist
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    from linrad_amd.multichan import install_exchange, run_coupled
    from linrad_amd.workload import chain_config, strong_liminfo
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("cpu:gloo,cuda:nccl", rank=0, world_size=1)
    try:
        dev = torch.device("cuda:0")
        res = []
        for entry in ("dsp", "stages"):
            if fuse is not None:
                os.environ["LRH_FUSE_FFT1"] = fuse
            cfg = chain_config(14, 12, batch=batch, fft3_n=10, mix2_n=8, rounds=max(2, nblk // batch))
            cfg.blanker_channels, cfg.timf1_channel_index = 2, 0
            if batch >= 4096:
                cfg.fft1_float_sparse = cfg.fft2_float_sparse = 1
            rx = open_hip(cfg)
            os.environ.pop("LRH_FUSE_FFT1", None)
            sy = synth_defaults(1 << 14, 0)
            rx.timf1_write(synth_iq(sy, 0, cfg.timf1_bytes // 4))
            rx.set_liminfo(strong_liminfo(sy, 14))
            rx.set_mix1_selfreq(0.31 * 4096 + 0.3)
            rx.set_pol(0.8, 0.36, -0.48)
            if entry == "dsp":
                install_exchange(rx, dist, dev)
                for _ in range(calls):
                    rx.wideband_dsp(nblk, batch)
            else:
                for _ in range(calls):
                    run_coupled(rx, nblk, batch, dist, device=dev, xy=True, pol=True)
            bs = rx.blanker_state()
            big = batch >= 4096                             # sparse rings there: the full-length ones stand in as zeros of the same shape
            res.append(([np.zeros(1, np.float32) if (big and r in (abi.RING_TIMF2_FLOAT, abi.RING_FFT2_FLOAT)) else rx.export(r)
                         for r in (abi.RING_FFT1_SUMSQ, abi.RING_TIMF2_FLOAT, abi.RING_TIMF2_PWR, abi.RING_FFT2_FLOAT, abi.RING_FFT2_XYSUM,
                                   abi.RING_WG_WATERF, abi.RING_TIMF3_FLOAT, abi.RING_BASEB_RAW)], rx.p.as_dict(), (bs.timf2_noise_floor, bs.stupid_bln_limit)))
            rx.close()
        assert res[0][1] == res[1][1] and res[0][2] == res[1][2]
        # (k_fft1w is switched off for this comparison, LRH_FUSE_FFT1=0 below: its spectra differ from k_fft1's in the last bit, and a blanker
        # decision on a sample within float32 rounding of the limit then differs too; fused against unfused is tests/test_gpu_fused.py)
        names = ("sumsq", "timf2", "pwr", "fft2", "xysum", "wf", "timf3", "baseb")
        for nm, a, b in zip(names, res[0][0], res[1][0]):
            if nm == "sumsq":                               # the sums ride inside k_timf2 in the one call: association of a group's adds
                assert np.max(np.abs(a - b) / (np.abs(b) + 1e-30)) < 1e-6
            else:
                assert np.array_equal(a, b), nm
        assert np.count_nonzero(res[0][0][-1]) > 50 and np.any(res[0][0][5])
    finally:
        dist.destroy_process_group()
