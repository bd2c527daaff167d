"""GPU: the correlation spectrum of two coupled channels (fft1_corrsum, fft1_slowcorr, fft1_slowcorr_tot; fft1.c:4146-4150, 4189-4193,
4584-4603, wide_graph.c:1033-1050), HIP library against the compiled two-channel reference run with
genparm[FFT1_CORRELATION_SPECTRUM] = 1 (tests/golden/twochan_*.npz), by stage calls and through lrh_wideband_dsp."""
import numpy as np
import pytest

import corrlib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,batch", [("twochan_n10", 1), ("twochan_n9_sin3", 1), ("twochan_real_n9", 1), ("twochan_n10", 3)])
def test_hip_correlation_spectrum_matches_reference(name, batch):
    from linrad_amd.lib import open_hip
    d, g, out = corrlib.run(open_hip, name, batch)
    print(name, corrlib.compare(d, g, out, 1e-5))


@pytest.mark.parametrize("name,batch", [("twochan_n10", 1), ("twochan_n10", 4)])
def test_hip_correlation_spectrum_through_wideband_dsp(name, batch):
    """both channels' contexts on one GPU, one caller thread each; the exchange points trade through device memory"""
    import torch
    from linrad_amd.lib import open_hip
    from oracle_binding import open_oracle
    d, g, out = corrlib.run_dsp(open_hip, name, batch, torch.device("cuda:0"))
    print(name, corrlib.compare(d, g, out, 1e-5))
    _, _, ref = corrlib.run_dsp(open_oracle, name, batch)
    for ch in (0, 1):
        assert out[ch]["p"] == ref[ch]["p"]
        # the blanker's borderline decisions may differ by float rounding (tests/test_twochan.py): the cleared sets agree to 97 %,
        # and what neither side cleared agrees to the float tolerance
        a, b = out[ch]["timf2"].reshape(-1, 2, 2).astype(np.float64), ref[ch]["timf2"].reshape(-1, 2, 2).astype(np.float64)
        ca, cb = (a[:, 0, :] == 0).all(axis=1), (b[:, 0, :] == 0).all(axis=1)
        assert (ca & cb).sum() / max((ca | cb).sum(), 1) > 0.97
        keep = ~(ca | cb)
        assert keep.sum() > 1000 and np.linalg.norm(a[keep, 1] - b[keep, 1]) <= 1e-5 * np.linalg.norm(b[keep, 1])


def test_correlation_calls_need_the_switch():
    from linrad_amd import abi
    from linrad_amd.lib import open_hip
    from refcases import lrh_config, twochan_case
    d, frames, lim = twochan_case("twochan_n10")
    iq = np.ascontiguousarray(frames.reshape(-1, 4)[:, 0:2]).ravel()
    rx = open_hip(lrh_config(d, iq, blanker_channels=2, timf1_channel_index=0))
    rx.timf1_write(iq)
    rx.fft1_b(1)
    with pytest.raises(RuntimeError):
        rx.fft1_corr_begin(rx.ptrs_copy(), 1)
    with pytest.raises(RuntimeError):
        rx.export(abi.RING_FFT1_CORRSUM)
    rx.set_correlation(True)
    rx.set_correlation(False)
    with pytest.raises(RuntimeError):
        rx.export(abi.RING_FFT1_SLOWCORR)
    rx.close()
