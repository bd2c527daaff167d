"""Correlation spectrum of two coupled channels (genparm[FFT1_CORRELATION_SPECTRUM] = 1): machinery shared by the oracle (CPU) and
HIP (GPU) tests.  Two contexts, one per channel, the all-gather of LRH_X_SPEC by hand, in the harness's call pattern; against the
compiled two-channel reference's fft1_corrsum / fft1_slowcorr / fft1_slowcorr_tot (goldens tests/golden/twochan_*.npz)."""
import os

import numpy as np

from linrad_amd import abi
from refcases import lrh_config, twochan_case

HERE = os.path.dirname(os.path.abspath(__file__))


def run(open_fn, name, batch=1):
    d, frames, lim = twochan_case(name)
    g = np.load(os.path.join(HERE, "golden", f"{name}.npz"))
    fr = frames.reshape(-1, 4)
    rxs = []
    for ch in (0, 1):
        iq = np.ascontiguousarray(frames[ch::2]) if d["real"] else np.ascontiguousarray(fr[:, 2 * ch:2 * ch + 2]).ravel()
        cfg = lrh_config(d, iq, blanker_channels=2, timf1_channel_index=ch, max_batch=max(4, batch))
        rx = open_fn(cfg)
        rx.timf1_write(iq)
        rx.set_liminfo(lim)
        if ch == 1:
            rx.set_ch2_phasing(d["ch2_c1"], d["ch2_c2"])
        rx.set_correlation(True)
        rxs.append(rx)
    left = d["nblk"]
    while left > 0:
        b = min(batch, left)
        ats = []
        for rx in rxs:
            rx.fft1_b(b)
            ats.append(rx.ptrs_copy())
            rx.fft1_c(b)
        n = [rx.fft1_corr_begin(at, b) for rx, at in zip(rxs, ats)]
        slots = [rx.exchange_read(rx.X_SPEC, n[0], ch * n[0]) for ch, rx in enumerate(rxs)]      # the all-gather, by hand
        for rx in rxs:
            for ch in (0, 1):
                rx.exchange_write(rx.X_SPEC, slots[ch], ch * n[0])
        for rx, at in zip(rxs, ats):
            rx.fft1_corr_finish(at, b)
            rx.make_timf2(b)
        left -= b
    out = [dict(corrsum=rx.export(abi.RING_FFT1_CORRSUM), slowcorr=rx.export(abi.RING_FFT1_SLOWCORR), tot=rx.export(abi.RING_FFT1_SLOWCORR_TOT),
                avgnum=rx.slowcorr_tot_avgnum(), sumsq=rx.export(abi.RING_FFT1_SUMSQ)) for rx in rxs]
    for rx in rxs:
        rx.close()
    return d, g, out


def compare(d, g, out, tol):
    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    rep = {}
    for ch in (0, 1):
        assert out[ch]["avgnum"] == int(g["slowcorr_tot_avgnum"][0]) and out[ch]["avgnum"] > 0
        # the period still being summed at the end of the run sits in the ring as a partial sum on both sides
        rep[f"corrsum{ch}"] = rel(out[ch]["corrsum"], g["fft1_corrsum"])
        rep[f"slowcorr{ch}"] = rel(out[ch]["slowcorr"], g["fft1_slowcorr"])
        rep[f"tot{ch}"] = rel(out[ch]["tot"], g["fft1_slowcorr_tot"])
        assert rep[f"corrsum{ch}"] < tol and rep[f"slowcorr{ch}"] < 4 * tol and rep[f"tot{ch}"] < tol, rep
    assert np.array_equal(out[0]["corrsum"], out[1]["corrsum"]) and np.array_equal(out[0]["tot"], out[1]["tot"])   # both contexts: the same rings
    assert np.count_nonzero(g["fft1_slowcorr"]) > 100
    return rep


def run_dsp(open_fn, name, batch, device=None):
    """the same through lrh_wideband_dsp: both contexts in this process, one thread each, the library asking for its collectives
    (linrad_amd.multichan.install_pair_exchange)"""
    import threading
    from linrad_amd.multichan import install_pair_exchange
    d, frames, lim = twochan_case(name)
    g = np.load(os.path.join(HERE, "golden", f"{name}.npz"))
    fr = frames.reshape(-1, 4)
    rxs = []
    for ch in (0, 1):
        iq = np.ascontiguousarray(frames[ch::2]) if d["real"] else np.ascontiguousarray(fr[:, 2 * ch:2 * ch + 2]).ravel()
        cfg = lrh_config(d, iq, blanker_channels=2, timf1_channel_index=ch, max_batch=max(4, batch))
        rx = open_fn(cfg)
        rx.timf1_write(iq)
        rx.set_liminfo(lim)
        if ch == 1:
            rx.set_ch2_phasing(d["ch2_c1"], d["ch2_c2"])
        rx.set_correlation(True)
        rxs.append(rx)
    install_pair_exchange(rxs, device)
    err = []

    def work(rx):
        try:
            rx.wideband_dsp(d["nblk"], batch)
        except Exception as e:  # noqa: BLE001
            err.append(e)
    th = [threading.Thread(target=work, args=(rx,)) for rx in rxs]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
        assert not t.is_alive()
    assert not err, err
    out = [dict(corrsum=rx.export(abi.RING_FFT1_CORRSUM), slowcorr=rx.export(abi.RING_FFT1_SLOWCORR), tot=rx.export(abi.RING_FFT1_SLOWCORR_TOT),
                avgnum=rx.slowcorr_tot_avgnum(), sumsq=rx.export(abi.RING_FFT1_SUMSQ), timf2=rx.export(abi.RING_TIMF2_FLOAT), p=rx.p.as_dict()) for rx in rxs]
    for rx in rxs:
        rx.close()
    return d, g, out
