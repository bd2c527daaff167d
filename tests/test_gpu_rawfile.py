"""GPU: recorded-IQ ingest.  The device-side 18-bit expansion against the compiled reference's golden ring image, and a
whole recording (header + packed blocks, written with linrad_amd.rawfile) played through the HIP path and the oracle."""
import os

import numpy as np
import pytest

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
    for fn in (_hip, _oracle):
        rx = fn(cfg)
        rd = rawfile.RawReader(path)
        assert rd.header.dword == bool(dword) and rd.header.rx_ad_speed == 2_000_000
        rd.feed(rx)
        rx.set_liminfo(lim)
        rx.set_mix1_selfreq(d["fq"])
        rx.wideband_dsp(d["nblk"], 4)
        out.append({k: rx.export(r) for r, k in ((abi.RING_TIMF1, "timf1"), (abi.RING_FFT1_FLOAT, "fft1"),
                                                 (abi.RING_FFT2_FLOAT, "fft2"), (abi.RING_TIMF3_FLOAT, "timf3"))})
    hh, oo = out
    assert np.array_equal(hh["timf1"], oo["timf1"])                      # the ring holds the same samples, bit for bit
    if dword:       # what the file can carry of the original samples: the top 16 bits (see test_rawfile_cpu)
        n = nblk_file * rawfile.BLOCK_BYTES // 4
        assert np.array_equal(hh["timf1"].view(np.int32)[:n] >> 16, iq[:n] >> 16)
    else:
        n = nblk_file * rawfile.BLOCK_BYTES // 2
        assert np.array_equal(hh["timf1"][:n], iq[:n])
    for k in ("fft1", "fft2", "timf3"):
        a, b = hh[k].astype(np.float64), oo[k].astype(np.float64)
        assert np.linalg.norm(a - b) <= 2e-5 * np.linalg.norm(b), k
