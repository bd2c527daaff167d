"""Two RF channels: the reference keeps both in one array (fft1win_dif_chan / dif_permute_chan fft1.c:2041, 660;
fft1_c's |X0|^2+|X1|^2 fft1.c:4132-4145; fft1back_two timf2.c:210; two-channel fft1back_fp_finish timf2.c:1067-1110),
this build shards them one per context.  Golden = the COMPILED REFERENCE run with ui.rx_rf_channels = 2
(tests/golden/make_golden_2ch.py); each context must reproduce its channel of the interleaved rings, and the two
cross-channel sums (fft1_sumsq, timf2_pwr) must equal the sums of the per-context rings (the exchange of
linrad_amd/multichan.py)."""
import os

import numpy as np
import pytest

from linrad_amd import abi
from refcases import lrh_config, twochan_case

HERE = os.path.dirname(os.path.abspath(__file__))


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def _run(open_fn, name, frames_mode, golden=None):
    d, frames, lim = twochan_case(name)
    g = golden if golden is not None else np.load(os.path.join(HERE, "golden", f"{name}.npz"))
    assert np.array_equal(g["frames"], frames)
    fr = frames.reshape(-1, 4)
    out = []
    for ch in (0, 1):
        # the channel's own stream: I/Q pairs out of frames {I0,Q0,I1,Q1}, or reals out of frames {a,b} (fft1_reherm_dit_two)
        iq = np.ascontiguousarray(frames[ch::2]) if d["real"] else np.ascontiguousarray(fr[:, 2 * ch:2 * ch + 2]).ravel()
        cfg = lrh_config(d, iq)
        if frames_mode:                       # the HIP path reads its channel out of the interleaved frames
            cfg.timf1_bytes *= 2
            cfg.timf1_frame_channels, cfg.timf1_channel_index = 2, ch
        rx = open_fn(cfg)
        rx.timf1_write(frames if frames_mode else iq)
        rx.set_liminfo(lim)
        if ch == 1:
            rx.set_ch2_phasing(d["ch2_c1"], d["ch2_c2"])
        for _ in range(d["nblk"]):            # the harness's call pattern: one block through all three stages
            rx.fft1_b(1), rx.fft1_c(1), rx.make_timf2(1)
        out.append(dict(fft1=rx.export(abi.RING_FFT1_FLOAT), sumsq=rx.export(abi.RING_FFT1_SUMSQ),
                        slowsum=rx.export(abi.RING_FFT1_SLOWSUM), timf2=rx.export(abi.RING_TIMF2_FLOAT),
                        pwr=rx.export(abi.RING_TIMF2_PWR), p=rx.p.as_dict(), M1=rx.N1 - rx.fft1_interleave_points))
    return d, g, out


def _check(d, g, out, tol):
    N1 = 1 << d["n1"]
    gf = g["fft1_float"].reshape(-1, N1, 2, 2)
    for ch in (0, 1):
        assert _rel(out[ch]["fft1"].reshape(-1, N1, 2), gf[:, :, ch, :]) < tol, ch
    # cross-channel power sum: what the all-reduce of multichan.cross_channel_power_sum forms
    assert _rel(out[0]["sumsq"] + out[1]["sumsq"], g["fft1_sumsq"]) < tol
    assert _rel(out[0]["slowsum"] + out[1]["slowsum"], g["fft1_slowsum"]) < 10 * tol
    assert out[0]["p"]["fft1_sumsq_pa"] == int(g["itrace"].reshape(-1, 16)[-1, 9])
    # timf2: reference sample layout {w0Re,w0Im,w1Re,w1Im,s0Re,s0Im,s1Re,s1Im}; finished samples only (the newest half
    # block of the reference ring holds a parked raw half)
    pa = int(g["itrace"].reshape(-1, 16)[-1, 0]) // 8
    assert pa == out[0]["p"]["timf2_pa"] // 4 and pa > 0
    gt = g["timf2_float"].reshape(-1, 2, 2, 2)            # [sample][weak/strong][channel][re/im]
    for ch in (0, 1):
        t = out[ch]["timf2"].reshape(-1, 2, 2)             # [sample][weak/strong][re/im]
        assert _rel(t[:pa], gt[:pa, :, ch, :]) < tol, ch
        assert np.abs(gt[:pa, 1, ch, :]).max() > 0          # the strong route is exercised
    assert _rel((out[0]["pwr"] + out[1]["pwr"])[:pa], g["timf2_pwr_float"][:pa]) < tol


@pytest.mark.parametrize("name", ["twochan_n10", "twochan_n9_sin3", "twochan_real_n9"])
def test_oracle_channels_match_two_channel_reference(name):
    from oracle_binding import open_oracle
    d, g, out = _run(open_oracle, name, frames_mode=False)
    _check(d, g, out, 2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["twochan_n10", "twochan_n9_sin3", "twochan_real_n9"])
def test_hip_contexts_match_two_channel_reference(name):
    from linrad_amd.lib import open_hip
    d, g, out = _run(open_hip, name, frames_mode=True)
    _check(d, g, out, 1e-5)


# ---- coupled blanker: decisions on the channel power sum, noise floor from both channels (blank1.c:1017, 1236-1300,
# 1510-1545, 1570); the two sum exchanges are done by hand here (tests/test_multichan_gloo.py does them with gloo)
def _run_coupled(open_fn, name, frames_mode, golden=None):
    d, frames, lim = twochan_case(name)
    g = golden if golden is not None else np.load(os.path.join(HERE, "golden", f"{name}.npz"))
    fr = frames.reshape(-1, 4)
    rxs = []
    for ch in (0, 1):
        iq = np.ascontiguousarray(fr[:, 2 * ch:2 * ch + 2]).ravel()
        cfg = lrh_config(d, iq, blanker_channels=2, timf1_channel_index=ch)
        if frames_mode:
            cfg.timf1_bytes *= 2
            cfg.timf1_frame_channels = 2
        rx = open_fn(cfg)
        rx.timf1_write(frames if frames_mode else iq)
        rx.set_liminfo(lim)
        if ch == 1:
            rx.set_ch2_phasing(d["ch2_c1"], d["ch2_c2"])
        rxs.append(rx)
    trace = []
    for _ in range(d["nblk"]):
        for rx in rxs:
            rx.fft1_b(1), rx.fft1_c(1), rx.make_timf2(1)
        n = [rx.blanker_begin() for rx in rxs]
        assert n[0] == n[1]
        if n[0]:
            tot = rxs[0].exchange_read(abi.StageAPI.X_PWR, n[0]) + rxs[1].exchange_read(abi.StageAPI.X_PWR, n[0])
            for rx in rxs:
                rx.exchange_write(abi.StageAPI.X_PWR, tot)
        for rx in rxs:
            rx.first_noise_blanker()
        if n[0]:
            st = rxs[0].exchange_read(abi.StageAPI.X_STAT, 2) + rxs[1].exchange_read(abi.StageAPI.X_STAT, 2)
            for rx in rxs:
                rx.exchange_write(abi.StageAPI.X_STAT, st)
                rx.blanker_finish()
        bs = [rx.blanker_state() for rx in rxs]
        trace.append([(b.timf2_noise_floor, b.stupid_bln_limit, b.timf2_cleared_points) for b in bs] + [rxs[0].p.timf2p_fit, rxs[0].p.timf2_pn2])
    out = [dict(timf2=rx.export(abi.RING_TIMF2_FLOAT), pwr=rx.export(abi.RING_TIMF2_PWR), bs=rx.blanker_state(), p=rx.p.as_dict()) for rx in rxs]
    return d, g, out, trace


def _check_coupled(d, g, out, trace, tol):
    it = g["bln_itrace"].reshape(-1, 16)
    tr = g["bln_trace"].reshape(-1, 16)
    for b, row in enumerate(trace):
        for ch in (0, 1):                                   # both contexts carry the same, the reference's, blanker state
            assert abs(row[ch][0] - it[b, 12]) <= 1 and abs(int(row[ch][1]) - it[b, 13]) <= 5, (b, ch)
        assert row[2] == it[b, 1] and 2 * row[3] == it[b, 2]      # timf2p_fit; timf2_pn2 counts 8 floats per sample there
    assert trace[-1][0] == trace[-1][1]
    exact = all(row[0][0] == it[b, 12] for b, row in enumerate(trace))
    fit = int(it[-1, 1])
    gt = g["bln_timf2_float"].reshape(-1, 2, 2, 2)
    gp = g["bln_timf2_pwr_float"]
    cleared_ref = gp[:fit] == 0
    for ch in (0, 1):
        own_cleared = (out[ch]["timf2"].reshape(-1, 2, 2)[:fit, 0, :] == 0).all(axis=1)
        if exact:
            assert np.array_equal(own_cleared, cleared_ref), ch
            assert _rel(out[ch]["timf2"].reshape(-1, 2, 2)[:fit], gt[:fit, :, ch, :]) < tol, ch
        else:
            inter, union = (own_cleared & cleared_ref).sum(), (own_cleared | cleared_ref).sum()
            assert inter / max(union, 1) > 0.97, ch
    assert cleared_ref.sum() > 50                            # the pulses are there
    for ch in (0, 1):
        assert abs(out[ch]["bs"].timf2_despiked_pwr[0] - tr[-1, 3]) <= 1e-4 * tr[-1, 3] + 0.5
        assert abs(out[ch]["bs"].timf2_despiked_pwr[1] - tr[-1, 6]) <= 1e-4 * tr[-1, 6] + 0.5


@pytest.mark.parametrize("name", ["twochan_n10", "twochan_n9_sin3"])
def test_oracle_coupled_blanker_matches_two_channel_reference(name):
    from oracle_binding import open_oracle
    d, g, out, trace = _run_coupled(open_oracle, name, frames_mode=False)
    _check_coupled(d, g, out, trace, 2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["twochan_n10", "twochan_n9_sin3"])
def test_hip_coupled_blanker_matches_two_channel_reference(name):
    from linrad_amd.lib import open_hip
    d, g, out, trace = _run_coupled(open_hip, name, frames_mode=True)
    _check_coupled(d, g, out, trace, 1e-5)


# ---- the whole two-channel chain: coupled blanker -> make_fft2 per channel -> cross products / sums / polarisation-
# independent waterfall line from both channels' bins (fft2.c:1622-1640, 1700-1815; the all-gather is done by hand here,
# tests/test_multichan_gloo.py does it with gloo) -> fft2_mix1_fixed per channel
def _chain_contexts(open_fn, name, frames_mode, golden_name=None, golden=None):
    """golden_name: the golden whose fft3 filter function is installed when `name` is a case of the random tests (refcases.TWOCHAN entry made on the fly);
    golden: the reference's dump itself (tests/test_oracle_vs_reference_random_cpu.py)"""
    d, frames, lim = twochan_case(name, chain=True)
    g = golden if golden is not None else np.load(os.path.join(HERE, "golden", f"{golden_name or name}_chain.npz"))
    fr = frames.reshape(-1, 4)
    rxs = []
    for ch in (0, 1):
        iq = np.ascontiguousarray(fr[:, 2 * ch:2 * ch + 2]).ravel()
        cfg = lrh_config(d, iq, blanker_channels=2, timf1_channel_index=ch)
        if frames_mode:
            cfg.timf1_bytes *= 2
            cfg.timf1_frame_channels = 2
        rx = open_fn(cfg)
        rx.timf1_write(frames if frames_mode else iq)
        rx.set_liminfo(lim)
        rx.set_mix1_selfreq(d["fq"])
        rx.set_bg_filterfunc(g["bg_filterfunc"])
        rx.set_pol(*d["pol"])
        if ch == 1:
            rx.set_ch2_phasing(d["ch2_c1"], d["ch2_c2"])
        rxs.append(rx)
    return d, g, rxs


def _run_chain(open_fn, name, frames_mode, batch=1, golden_name=None, golden=None):
    d, g, rxs = _chain_contexts(open_fn, name, frames_mode, golden_name, golden)
    wf_lines, nfft2 = [], 0
    X = abi.StageAPI
    for _ in range(d["nblk"]):
        for rx in rxs:
            rx.fft1_b(1), rx.fft1_c(1), rx.make_timf2(1)
        n = [rx.blanker_begin() for rx in rxs]
        if n[0]:
            tot = rxs[0].exchange_read(X.X_PWR, n[0]) + rxs[1].exchange_read(X.X_PWR, n[0])
            for rx in rxs:
                rx.exchange_write(X.X_PWR, tot)
        for rx in rxs:
            rx.first_noise_blanker()
        if n[0]:
            st = rxs[0].exchange_read(X.X_STAT, 2) + rxs[1].exchange_read(X.X_STAT, 2)
            for rx in rxs:
                rx.exchange_write(X.X_STAT, st)
                rx.blanker_finish()
        k = rxs[0].fft2_available()
        assert k == rxs[1].fft2_available()
        while k > 0:
            kb = min(k, batch)
            at = [rx.ptrs_copy() for rx in rxs]
            cnt = []
            for rx, a in zip(rxs, at):
                rx.make_fft2(kb)
                cnt.append(rx.fft2_xy_begin(a, kb))
            assert cnt[0] == cnt[1] == kb * 2 * rxs[0].N2
            own = [rx.exchange_read(X.X_BINS, cnt[0], ch * cnt[0]) for ch, rx in enumerate(rxs)]
            for ch, rx in enumerate(rxs):                      # all-gather: each context receives the other channel's slot
                rx.exchange_write(X.X_BINS, own[1 - ch], (1 - ch) * cnt[0])
                rx.fft2_xy_finish(at[ch], kb)
            wptr = at[0].wg_waterf_ptr
            for _ in range((at[0].wg_waterf_sum_counter + kb) // d["wf_avgnum"]):
                wf_lines.append([rx.export(abi.RING_WG_WATERF, wptr, rx.cfg.wf_xpixels) for rx in rxs])
                wptr = (wptr - rxs[0].cfg.wf_xpixels) % (rxs[0].cfg.wf_lines * rxs[0].cfg.wf_xpixels)
            for rx in rxs:
                rx.fft2_mix1_fixed(kb)
            # fft3, then fft3_mix2 with the polarisation sums A / B formed over both contexts (all-reduce by hand)
            k3 = rxs[0].fft3_available()
            while k3 > 0:
                k3b = min(k3, max(1, batch), rxs[0].cfg.max_fft3n // 2)
                cnt3 = []
                for rx in rxs:
                    rx.make_fft3_all(k3b)
                    cnt3.append(rx.mix2_pol_begin(k3b))
                assert cnt3[0] == cnt3[1] == 4 * k3b * (1 << d["mix2_n"])
                tot = rxs[0].exchange_read(X.X_POL, cnt3[0]) + rxs[1].exchange_read(X.X_POL, cnt3[0])
                for rx in rxs:
                    rx.exchange_write(X.X_POL, tot)
                    rx.fft3_mix2(k3b)
                k3 -= k3b
            nfft2 += kb
            k -= kb
    out = [dict(fft3=rx.export(abi.RING_FFT3), baseb=rx.export(abi.RING_BASEB_RAW), fft2=rx.export(abi.RING_FFT2_FLOAT), xyp=rx.export(abi.RING_FFT2_XYPOWER), xys=rx.export(abi.RING_FFT2_XYSUM),
                timf3=rx.export(abi.RING_TIMF3_FLOAT), p=rx.p.as_dict(), bs=rx.blanker_state(), pwr=rx.export(abi.RING_TIMF2_PWR), timf2=rx.export(abi.RING_TIMF2_FLOAT)) for rx in rxs]
    if golden_name is not None or golden is not None:
        for rx in rxs:
            rx.close()
    return d, g, out, wf_lines, nfft2


def _check_chain(d, g, out, wf_lines, nfft2, tol, fixture_checks=True):
    """fixture_checks: the demands on the FIXTURE itself (enough fft3 transforms, a polarisation pair that differs) -- off for the random cases"""
    N2 = 1 << d["n2"]
    fin = g["final"]
    if wf_lines is not None:                                 # (None: the rings only -- a run through lrh_wideband_dsp hands no lines out on the way)
        assert nfft2 == fin[10] and len(wf_lines) == fin[11] and nfft2 >= 4 and fin[11] >= 2
    gf = g["fft2_float"].reshape(-1, N2, 2, 2)
    gt = g["timf3_float"].reshape(-1, 2, 2)
    for ch in (0, 1):
        assert out[ch]["p"]["fft2_na"] == fin[7] and out[ch]["p"]["fft2_nx"] == fin[8] and 2 * out[ch]["p"]["timf3_pa"] == fin[9]
        assert _rel(out[ch]["fft2"].reshape(-1, N2, 2), gf[:, :, ch, :]) < tol, ch
        # timf3 is a weak band cut from the wide spectrum: relative bound, or the float32 floor of that spectrum (paritylib)
        nm, a, b = N2 >> d["mixred"], out[ch]["timf3"].reshape(-1, 2).astype(np.float64), gt[:, ch, :].astype(np.float64)
        wide = np.linalg.norm(gf[:, :, ch, :].astype(np.float64)) / np.sqrt(gf.shape[0])
        floor = 4 * 6e-8 * wide * np.sqrt(nm / N2) * np.sqrt(a.size / nm / 2) * np.sqrt(nm)
        assert np.abs(b).max() > 0 and (_rel(a, b) < tol or np.linalg.norm(a - b) <= floor), (ch, _rel(a, b), floor)
        # cross products: both contexts hold the reference's rings
        assert _rel(out[ch]["xyp"], g["fft2_xypower"]) < 2 * tol, ch
        assert _rel(out[ch]["xys"], g["fft2_xysum"]) < 2 * tol, ch
    assert np.abs(g["fft2_xypower"].reshape(-1, 4)[:, 2:]).max() > 0
    # fft3 per channel; the polarisation pair: context 0 carries baseb_raw (A), context 1 baseb_raw_orthog (B)
    N3 = 1 << d["fft3_n"]
    g3 = g["fft3"].reshape(-1, N3, 2, 2)
    assert g["fft3_ptrs"][0] >= (8 if fixture_checks else 4)
    for ch, key in ((0, "baseb_raw"), (1, "baseb_raw_orthog")):
        assert out[ch]["p"]["fft3_pa"] == g["fft3_ptrs"][1] // 2 and out[ch]["p"]["timf3_px"] == g["fft3_ptrs"][2] // 2
        assert out[ch]["p"]["baseb_pa"] == g["baseb_ptrs"][0] and out[ch]["p"]["fft3_px"] == g["baseb_ptrs"][1] // 2
        # float32 floor of a weak band cut from the wide spectrum (see timf3 above): rms e_s per timf3 sample; an
        # unnormalised N3-point transform turns it into e_s sqrt(N3) per bin, the mix2.size-point back transform of the
        # filtered bins into at most e_s sqrt(N3 Nm2) per sample, overlap-added twice
        nm, Nm2 = N2 >> d["mixred"], 1 << d["mix2_n"]
        wide = max(np.linalg.norm(gf[:, :, c_, :].astype(np.float64)) for c_ in (0, 1)) / np.sqrt(gf.shape[0])
        e_s = 4 * 6e-8 * wide * np.sqrt(nm / N2)
        a, b = out[ch]["fft3"].reshape(-1, N3, 2).astype(np.float64), g3[:, :, ch, :].astype(np.float64)
        assert _rel(a, b) < tol or np.linalg.norm(a - b) <= e_s * np.sqrt(N3) * np.sqrt(a.size / 2), (ch, _rel(a, b))
        a, b = out[ch]["baseb"].astype(np.float64), g[key].astype(np.float64)
        assert np.count_nonzero(b) >= 100
        assert _rel(a, b) < tol or np.linalg.norm(a - b) <= e_s * np.sqrt(2 * N3 * Nm2) * np.sqrt(np.count_nonzero(b) / 2), (key, _rel(a, b))
    assert not fixture_checks or _rel(g["baseb_raw_orthog"], g["baseb_raw"]) > 0.5
    if wf_lines is None:
        return
    gw = g["wf_lines"].reshape(len(wf_lines), -1).astype(np.int32)
    for ch in (0, 1):
        ow = np.array([ln[ch] for ln in wf_lines], np.int32)
        diff = np.abs(ow - gw)
        assert diff.max() <= 2 and np.mean(diff != 0) < 0.02, (ch, diff.max(), np.mean(diff != 0))
    assert np.array_equal(np.array([ln[0] for ln in wf_lines]), np.array([ln[1] for ln in wf_lines]))


@pytest.mark.parametrize("name", ["twochan_n10", "twochan_n9_sin3"])
def test_oracle_two_channel_chain_matches_reference(name):
    from oracle_binding import open_oracle
    _check_chain(*_run_chain(open_oracle, name, frames_mode=False), 2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name,batch", [("twochan_n10", 1), ("twochan_n9_sin3", 1), ("twochan_n10", 3)])
def test_hip_two_channel_chain_matches_reference(name, batch):
    from linrad_amd.lib import open_hip
    _check_chain(*_run_chain(open_hip, name, frames_mode=True, batch=batch), 1e-5)


@pytest.mark.gpu
def test_hip_pair_through_wideband_dsp_equals_the_stage_calls(monkeypatch):
    """Both channels' contexts on one GPU, one caller thread each, ONE lrh_wideband_dsp call per context with the collectives traded
    through device memory (linrad_amd.multichan.install_pair_exchange): inside that call the own channel's fft2 transforms go into the
    gather and into k_xypower where they lie in the fft2 ring (`own` of lrh_exchange_fn), no copy into the exchange slot.  Every ring
    must equal the stage-by-stage form with the exchanges made by hand, which the goldens of the compiled two-channel reference pin
    (test_hip_two_channel_chain_matches_reference)."""
    import threading
    import torch
    from linrad_amd.lib import open_hip
    from linrad_amd.multichan import install_pair_exchange
    monkeypatch.setenv("LRH_FUSE_FFT1", "0")                # the same kernels on both sides (k_fft1w rounds its last pass differently)
    d, g, stage, wf_lines, nfft2 = _run_chain(open_hip, "twochan_n10", frames_mode=True)
    _, _, rxs = _chain_contexts(open_hip, "twochan_n10", frames_mode=True)
    install_pair_exchange(rxs, torch.device("cuda:0"))
    err = []

    def work(rx):
        try:
            rx.wideband_dsp(d["nblk"], 1)
        except Exception as e:  # noqa: BLE001
            err.append(e)
    th = [threading.Thread(target=work, args=(rx,)) for rx in rxs]
    [t.start() for t in th]
    for t in th:
        t.join(180)
        assert not t.is_alive()
    assert not err, err
    assert nfft2 >= 4
    for ch, rx in enumerate(rxs):
        assert rx.p.as_dict() == stage[ch]["p"]
        for ring, key in ((abi.RING_FFT2_FLOAT, "fft2"), (abi.RING_FFT2_XYPOWER, "xyp"), (abi.RING_FFT2_XYSUM, "xys"), (abi.RING_TIMF3_FLOAT, "timf3"),
                          (abi.RING_FFT3, "fft3"), (abi.RING_BASEB_RAW, "baseb")):
            assert np.array_equal(rx.export(ring), stage[ch][key]), (ch, key)
        rx.close()
