"""Recorded-IQ ingest, CPU side: the 18-bit packing against the compiled reference's golden
vectors (tests/golden/rawdat_18bit.npz, generator tests/golden/make_rawdat_golden.py), the oracle's expansion into the
timf1 ring, and the .raw header reader / writer (restated from modesub.c; round trip and the legacy layout)."""
import io
import os
import struct

import numpy as np
import pytest

from linrad_amd import abi, rawfile
from linrad_amd.abi import default_config

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rawdat_18bit.npz"))


def test_compress_matches_reference():
    assert np.array_equal(rawfile.compress_rawdat(G["samples"]), G["packed"])


def test_expand_matches_reference():
    pa, ring = int(G["pa"]), G["ring"]
    e = rawfile.expand_rawdat(G["packed"]).view(np.uint8)
    assert np.array_equal(ring[pa:pa + e.size], e)
    assert not ring[:pa].any() and not ring[pa + e.size:].any()
    # left-justified 18-bit value, half-LSB bit set, garbage below dropped (csplit.c:25-31); the two bits kept in the
    # ninth byte are NOT the sample's own bits 15:14 (compress/expand are not inverses there), the top 16 bits are
    v = e.view(np.int32)
    assert np.array_equal(v >> 16, G["samples"] >> 16)
    assert np.all((v & 0x3fff) == 0x2000)


def test_oracle_expands_into_the_ring():
    from oracle_binding import open_oracle
    ring_bytes = 1 << int(G["ring_log2"])
    cfg = default_config(8, 8, timf1_bytes=ring_bytes, timf1_dword_input=1)
    rx = open_oracle(cfg)
    rx.timf1_write_packed18(G["packed"], int(G["pa"]))
    got = rx.export(abi.RING_TIMF1).view(np.uint8)
    assert np.array_equal(got, G["ring"])
    # wrap-around: the second half of the data lands at the start of the ring
    rx2 = open_oracle(cfg)
    nexp = G["packed"].size // 9 * 16
    rx2.timf1_write_packed18(G["packed"], ring_bytes - nexp // 2)
    got2 = rx2.export(abi.RING_TIMF1).view(np.uint8)
    e = rawfile.expand_rawdat(G["packed"]).view(np.uint8)
    assert np.array_equal(got2[ring_bytes - nexp // 2:], e[:nexp // 2]) and np.array_equal(got2[:nexp // 2], e[nexp // 2:])


def test_packed18_rejects_bad_calls():
    from oracle_binding import open_oracle
    rx16 = open_oracle(default_config(8, 8))
    with pytest.raises(abi.LrhError):
        rx16.timf1_write_packed18(G["packed"][:9], 0)                 # int16 receiver
    rx = open_oracle(default_config(8, 8, timf1_dword_input=1))
    with pytest.raises(abi.LrhError):
        rx.timf1_write_packed18(G["packed"][:10], 0)                  # not a multiple of 9
    with pytest.raises(abi.LrhError):
        rx.timf1_write_packed18(G["packed"][:9], 8)                   # offset not on a 16-byte group


def test_header_round_trip_and_legacy_layout(tmp_path):
    h = rawfile.RawHeader(rx_input_mode=rawfile.IQ_DATA | rawfile.DWORD_INPUT, rx_rf_channels=1, rx_ad_channels=2,
                          rx_ad_speed=10_000_000, diskread_time=4711.0, passband_center=144.1, passband_direction=-1)
    # embedded filtercorr calibration block: rdbuf[1] points, frequency domain (rdbuf[7] = 0), rdbuf[6] channels
    rd = (10, 1024, 0, h.rx_input_mode, 0, h.rx_ad_speed, 1, 0, 0, 0)
    h.save_init_flag = 1
    h.calibration = struct.pack("<10i", *rd) + bytes(1024 * 4 * 3) + struct.pack("<10i", *rd)
    p = tmp_path / "t.raw"
    n = rawfile.write_raw(p, h, G["samples"])
    assert n == 2
    r = rawfile.RawReader(p)
    g = r.header
    assert (g.rx_input_mode, g.rx_rf_channels, g.rx_ad_channels, g.rx_ad_speed) == (h.rx_input_mode, 1, 2, 10_000_000)
    assert (g.diskread_time, g.passband_center, g.passband_direction, g.save_init_flag) == (4711.0, 144.1, -1, 1)
    assert g.calibration == h.calibration and g.save_rw_bytes == 4608
    blocks = list(r.blocks())
    assert len(blocks) == 2 and np.array_equal(np.concatenate(blocks), G["packed"])
    # the original layout: first int = rx_input_mode >= 0, no time / passband fields (modesub.c:712-719)
    legacy = struct.pack("<iiiiB", rawfile.IQ_DATA, 1, 2, 96000, 0) + np.arange(8192, dtype=np.int16).tobytes()[:8192]
    g2 = rawfile.read_header(io.BytesIO(legacy))
    assert (g2.rx_input_mode, g2.rx_ad_speed, g2.remember, g2.passband_direction, g2.data_offset) == \
        (rawfile.IQ_DATA, 96000, rawfile.REMEMBER_NOTHING, 1, 17)
    # corrupted: calibration trailer does not repeat the leader
    bad = io.BytesIO()
    h.calibration = h.calibration[:-4] + struct.pack("<i", 99)
    rawfile.write_header(bad, h)
    bad.seek(0)
    with pytest.raises(rawfile.RawFileError):
        rawfile.read_header(bad)


def _filehdr_cases(prefix):
    import json
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "filehdr.npz"))
    return {k[:-6]: (g[k].tobytes(), json.loads(g[k[:-6] + "__ref"].tobytes().decode())) for k in g.files if k.startswith(prefix) and k.endswith("__file")}


@pytest.mark.parametrize("name", ["raw_plain", "raw_perseus_rev", "raw_sdr14_dword", "raw_twochan", "raw_real_mono", "raw_oldformat"])
def test_raw_header_reader_equals_the_compiled_references(name):
    """tests/golden/filehdr.npz: the bytes of small .raw files and what the COMPILED REFERENCE's open_savefile (modesub.c:606-733, run
    head-less by oracle/ref_files.c) left in Linrad's globals after reading them; linrad_amd.rawfile.read_header must read the same out
    of the same bytes -- every header field and the position of the first data block"""
    import io
    blob, ref = _filehdr_cases("raw_")[name]
    f = io.BytesIO(blob)
    h = rawfile.read_header(f)
    assert (h.rx_input_mode, h.rx_rf_channels, h.rx_ad_channels, h.rx_ad_speed, h.save_init_flag) == \
           (ref["rx_input_mode"], ref["rx_rf_channels"], ref["rx_ad_channels"], ref["rx_ad_speed"], ref["save_init_flag"])
    assert h.remember == ref["remember0"] and len(h.proprietary) == ref["remember1"]
    assert (h.diskread_time, h.passband_center, h.passband_direction) == (ref["diskread_time"], ref["passband_center"], ref["passband_direction"])
    assert h.data_offset == ref["file_pos"] == f.tell()
    if ref["remember0"] == rawfile.REMEMBER_PERSEUS:          # the Perseus chunk's own fields, as the reference's RCVR struct holds them
        import struct
        assert struct.unpack_from("<I", h.proprietary, 0)[0] == ref["perseus_center_hz"] and struct.unpack_from("<q", h.proprietary, 8)[0] == ref["perseus_time"]
