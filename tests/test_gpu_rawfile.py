"""GPU: recorded-IQ ingest.  The device-side 18-bit expansion against the compiled reference's golden ring image, and a
whole recording (header + packed blocks, written with linrad_amd.rawfile) played through the HIP path and the oracle."""
import os

import numpy as np
import pytest

from paritylib import truth_gate

from linrad_amd import abi, rawfile
from linrad_amd.abi import default_config
from refcases import case_params, lrh_config, make_input, make_liminfo

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rawdat_18bit.npz"))


def _hip(cfg):
    from linrad_amd.lib import open_hip
    return open_hip(cfg)


def _oracle(cfg):
    from oracle_binding import open_oracle
    return open_oracle(cfg)


def _truth(cfg):
    from oracle_binding import open_truth
    return open_truth(cfg)


def test_device_expansion_matches_reference_golden():
    ring_bytes = 1 << int(G["ring_log2"])
    d = case_params("n10_n12_dword")
    cfg = lrh_config(d, make_input(d), timf1_bytes=ring_bytes)
    rx = _hip(cfg)
    rx.timf1_write_packed18(G["packed"], int(G["pa"]))
    assert np.array_equal(rx.export(abi.RING_TIMF1).view(np.uint8), G["ring"])
    # wrap: same bytes as the oracle
    nexp = G["packed"].size // 9 * 16
    rx2, ro = _hip(cfg), _oracle(cfg)
    for r in (rx2, ro):
        r.timf1_write_packed18(G["packed"], ring_bytes - nexp // 2)
    assert np.array_equal(rx2.export(abi.RING_TIMF1), ro.export(abi.RING_TIMF1))


def test_net_export_survives_a_growing_packed18_staging_buffer():
    """lrh_export_timf2_net keeps its own staging buffer; a later, larger lrh_timf1_write_packed18 (which re-allocates its
    staging) must not disturb it (round-1 advisor finding: a stray hipFree of the net buffer in the grow branch)."""
    d = case_params("n10_n12_dword")
    cfg = lrh_config(d, make_input(d), timf1_bytes=1 << int(G["ring_log2"]))
    rx, ro = _hip(cfg), _oracle(cfg)
    first = None
    for r in (rx, ro):
        r.timf1_write_packed18(G["packed"][:9 * 64], 0)              # small staging buffer first
        r.wideband_dsp(8, 4)
    a0 = rx.export_timf2_net(0, 1024, 2.0, 0.5)
    rx.timf1_write_packed18(G["packed"], 0)                          # grows the packed18 staging
    a1 = rx.export_timf2_net(0, 1024, 2.0, 0.5)                      # count <= net_cap: reuses the net staging
    a2 = rx.export_timf2_net(0, 2048, 2.0, 0.5)                      # and grows it
    assert np.array_equal(a0, a1) and np.array_equal(a0, a2[:2048])
    ref = ro.export_timf2_net(0, 1024, 2.0, 0.5)
    assert np.linalg.norm(a0 - ref) <= 1e-5 * max(np.linalg.norm(ref), 1e-30)


@pytest.mark.parametrize("dword", [1, 0])
def test_recording_plays_through_the_chain(tmp_path, dword):
    name = "n10_n12_dword" if dword else "n10_n12"
    d = case_params(name)
    d["nblk"] = 24
    iq, lim = make_input(d), make_liminfo(d)
    h = rawfile.RawHeader(rx_input_mode=rawfile.IQ_DATA | (rawfile.DWORD_INPUT if dword else 0), rx_ad_speed=2_000_000)
    path = tmp_path / "rec.raw"
    nblk_file = rawfile.write_raw(path, h, iq)
    assert nblk_file * rawfile.BLOCK_BYTES >= (d["nblk"] * 512 + 2048) * (8 if dword else 4)
    cfg = lrh_config(d, iq)
    cfg.sample_shift = 0
    out = []
    for fn in (_hip, _oracle, _truth):
        rx = fn(cfg)
        rd = rawfile.RawReader(path)
        assert rd.header.dword == bool(dword) and rd.header.rx_ad_speed == 2_000_000
        rd.feed(rx)
        rx.set_liminfo(lim)
        rx.set_mix1_selfreq(d["fq"])
        rx.wideband_dsp(d["nblk"], 4)
        out.append({k: rx.export(r) for r, k in ((abi.RING_TIMF1, "timf1"), (abi.RING_FFT1_FLOAT, "fft1"),
                                                 (abi.RING_FFT2_FLOAT, "fft2"), (abi.RING_TIMF3_FLOAT, "timf3"))})
    hh, oo, tt = out
    assert np.array_equal(hh["timf1"], oo["timf1"])                      # the ring holds the same samples, bit for bit
    if dword:       # what the file can carry of the original samples: the top 16 bits (see test_rawfile_cpu)
        n = nblk_file * rawfile.BLOCK_BYTES // 4
        assert np.array_equal(hh["timf1"].view(np.int32)[:n] >> 16, iq[:n] >> 16)
    else:
        n = nblk_file * rawfile.BLOCK_BYTES // 2
        assert np.array_equal(hh["timf1"][:n], iq[:n])
    rep = {}
    for k in ("fft1", "fft2", "timf3"):      # 1e-5, or no further from the float64 truth than the oracle's own float32 result (paritylib.truth_gate)
        truth_gate(rep, k, hh[k], oo[k], tt[k], factor=1.05)     # (the partner is the oracle: see tests/test_gpu_fullsize.py ORACLE_FACTOR)
    print(rep)


@pytest.mark.parametrize("fmt", ["pcm16", "pcm24"])
def test_wav_recording_plays_through_the_chain(tmp_path, fmt):
    """A .wav recording (16-bit as written by write_wav with a Perseus chunk; 24-bit PCM built by hand) through
    WavReader into the HIP chain and the oracle: same ring image, same spectra."""
    import struct
    from linrad_amd import wavfile
    dword = fmt == "pcm24"
    d = case_params("n10_n12_dword" if dword else "n10_n12")
    d["nblk"] = 24
    iq, lim = make_input(d), make_liminfo(d)
    path = tmp_path / "rec.wav"
    if dword:
        b = (iq.astype(np.int64) >> 8).astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3].tobytes()   # top 24 bits
        body = b"WAVEfmt " + struct.pack("<ihhiihh", 16, 1, 2, 2_000_000, 12_000_000, 6, 24) + b"data" + struct.pack("<i", len(b)) + b
        path.write_bytes(b"RIFF" + struct.pack("<i", len(body)) + body + b"\x7f" * 40)     # trailing non-sample bytes
    else:
        wavfile.write_wav(path, iq, 2_000_000, 2, proprietary=(b"rcvr", struct.pack("<IIq", 144_300_000, 4, 1_700_000_000)))
    cfg = lrh_config(d, iq)
    cfg.sample_shift = 0
    out = []
    for fn in (_hip, _oracle, _truth):
        rx = fn(cfg)
        rd = wavfile.WavReader(str(path))
        assert rd.header.dword == dword and rd.header.rx_ad_speed == 2_000_000 and rd.header.rx_ad_channels == 2
        if not dword:
            assert rd.header.passband_center == pytest.approx(144.3)
        nbytes = rd.feed(rx)
        rx.set_liminfo(lim)
        rx.set_mix1_selfreq(d["fq"])
        rx.wideband_dsp(d["nblk"], 4)
        out.append({k: rx.export(r) for r, k in ((abi.RING_TIMF1, "timf1"), (abi.RING_FFT1_FLOAT, "fft1"),
                                                 (abi.RING_FFT2_FLOAT, "fft2"), (abi.RING_TIMF3_FLOAT, "timf3"))})
    hh, oo, tt = out
    assert np.array_equal(hh["timf1"], oo["timf1"])
    if dword:
        n = nbytes // 4
        assert np.array_equal(hh["timf1"].view(np.int32)[:n], (iq[:n] >> 8) << 8)          # 24 bits survive, left-justified
    else:
        assert np.array_equal(hh["timf1"][:nbytes // 2], iq[:nbytes // 2])
    rep = {}
    for k in ("fft1", "fft2", "timf3"):      # 1e-5, or no further from the float64 truth than the oracle's own float32 result (paritylib.truth_gate)
        truth_gate(rep, k, hh[k], oo[k], tt[k], factor=1.05)     # (the partner is the oracle: see tests/test_gpu_fullsize.py ORACLE_FACTOR)
    print(rep)
