"""WAV ingest (linrad_amd/wavfile.py) against known-answer files built field by field from the format definitions the
reference reads (modesub.c:113-142, 1088-1347; rxin.c:1572-1640)."""
import io
import struct

import numpy as np
import pytest

from linrad_amd import wavfile as W
from linrad_amd.rawfile import (BYTE_INPUT, DIGITAL_IQ, DWORD_INPUT, FLOAT_INPUT, QWORD_INPUT, REMEMBER_NOTHING,
                                REMEMBER_PERSEUS, REMEMBER_SDR14)


def fmt_chunk(tag, channels, rate, bytes_per, extra=b""):
    body = struct.pack("<hhiihh", tag, channels, rate, rate * bytes_per * channels, bytes_per * channels, 8 * bytes_per) + extra
    return b"fmt " + struct.pack("<i", len(body)) + body


def riff(*chunks, data=b""):
    body = b"WAVE" + b"".join(chunks) + b"data" + struct.pack("<i", len(data)) + data
    return b"RIFF" + struct.pack("<i", len(body)) + body


def rcvr_payload(freq_hz, t0):
    return struct.pack("<IIqH4B16s", freq_hz, 3, t0, 1, 0, 1, 0, 0, b"\0" * 16)


def test_plain_pcm16_header_and_modes():
    for tag, bytes_per, mode in ((1, 1, BYTE_INPUT), (1, 2, 0), (1, 3, BYTE_INPUT + DWORD_INPUT),
                                 (1, 4, QWORD_INPUT + DWORD_INPUT), (3, 4, FLOAT_INPUT + DWORD_INPUT)):
        b = riff(fmt_chunk(tag, 2, 192000, bytes_per), data=b"\1\2\3\4")
        f = io.BytesIO(b)
        h = W.read_wav_header(f)
        assert (h.rx_ad_channels, h.rx_ad_speed, h.rx_input_mode, h.remember) == (2, 192000, mode, REMEMBER_NOTHING)
        assert h.data_offset == len(b) - 4 and f.read() == b"\1\2\3\4"
        assert not h.freq_from_file and h.passband_center == 0


def test_extensible_format_chunk_and_unknown_chunks_are_skipped():
    b = riff(fmt_chunk(1, 1, 48000, 2, extra=b"\x16\0" + b"\xaa" * 22), b"LIST" + struct.pack("<i", 5) + b"hello",
             b"fact" + struct.pack("<i", 4) + b"\0\0\0\0", data=b"xy")
    h = W.read_wav_header(io.BytesIO(b))
    assert (h.rx_ad_channels, h.rx_ad_speed, h.rx_input_mode) == (1, 48000, 0) and h.data_offset == len(b) - 2


def test_perseus_rcvr_chunk():
    p = rcvr_payload(14_200_000, 1_600_000_000)
    b = riff(fmt_chunk(1, 2, 2_000_000, 2), b"rcvr" + struct.pack("<i", len(p)) + p, data=b"")
    h = W.read_wav_header(io.BytesIO(b))
    assert h.remember == REMEMBER_PERSEUS and h.proprietary == p and h.freq_from_file
    assert h.passband_center == pytest.approx(14.2) and h.diskread_time == 1_600_000_000.0
    # a chunk skipped afterwards resets the remembered chunk type but not what was taken from it (skip_chunk label)
    b2 = riff(fmt_chunk(1, 2, 2_000_000, 2), b"rcvr" + struct.pack("<i", len(p)) + p, b"junk" + struct.pack("<i", 2) + b"ab")
    h2 = W.read_wav_header(io.BytesIO(b2))
    assert h2.remember == REMEMBER_NOTHING and h2.passband_center == pytest.approx(14.2)
    with pytest.raises(W.WavFileError) as e:                       # longer than the struct
        W.read_wav_header(io.BytesIO(riff(fmt_chunk(1, 2, 1, 2), b"rcvr" + struct.pack("<i", 65) + b"\0" * 65)))
    assert e.value.errnr == 13
    with pytest.raises(W.WavFileError):                            # two proprietary chunks
        W.read_wav_header(io.BytesIO(riff(fmt_chunk(1, 2, 1, 2), (b"rcvr" + struct.pack("<i", len(p)) + p) * 2)))


def test_sdr14_auxi_chunk_binary_and_sdr_console_xml():
    st = struct.pack("<8H", 2024, 5, 3, 17, 13, 45, 7, 250)
    p = st + st + struct.pack("<7I", 7_050_000, 66_666_667, 0, 190_000, 0, 0, 0)
    h = W.read_wav_header(io.BytesIO(riff(fmt_chunk(1, 2, 196078, 2), b"auxi" + struct.pack("<i", len(p)) + p)))
    assert h.remember == REMEMBER_SDR14 and h.proprietary == p
    assert h.diskread_time == 13 * 3600 + 45 * 60 + 7 and h.passband_center == pytest.approx(7.05)
    xml = '<?xml version="1.0"?><SDR-XML-Root Description="x" CurrentTimeUTC="05-01-2024 21:03:09" RadioCenterFreq="10489750000" SampleRate="1"/>  '
    px = xml.encode("utf-16-le")
    hx = W.read_wav_header(io.BytesIO(riff(fmt_chunk(1, 2, 1_000_000, 2), b"auxi" + struct.pack("<i", len(px)) + px)))
    assert hx.remember == REMEMBER_NOTHING and hx.diskread_time == 21 * 3600 + 3 * 60 + 9
    assert hx.passband_center == pytest.approx(0.000001 * float(np.float32(10489750000)))


def test_expert_sdr2_marks_digital_iq_and_eats_one_byte():
    b = riff(fmt_chunk(3, 2, 312500, 4), b"esdr" + struct.pack("<i", 3) + b"abc", data=b"\x99" + b"\0" * 8)
    f = io.BytesIO(b)
    h = W.read_wav_header(f)
    assert h.expert_sdr2 and h.rx_input_mode == FLOAT_INPUT + DWORD_INPUT + DIGITAL_IQ and f.read() == b"\0" * 8


@pytest.mark.parametrize("blob,errnr", [
    (b"RIFX" + b"\0" * 40, 1), (b"RIFF\0\0", 2), (b"RIFF\0\0\0\0WAVX", 2), (b"RIFF\0\0\0\0WAVEfmtx", 3),
    (b"RIFF\0\0\0\0WAVEfmt \x10\0\0\0" + struct.pack("<h", 2), 5),
    (b"RIFF\0\0\0\0WAVEfmt \x10\0\0\0" + struct.pack("<hh", 1, 3), 6),
    (b"RIFF\0\0\0\0WAVEfmt \x10\0\0\0" + struct.pack("<hhiih", 1, 2, 8000, 0, 10), 9),
    (b"RIFF\0\0\0\0WAVEfmt \x10\0\0\0" + struct.pack("<hhiih", 3, 2, 8000, 0, 4), 11),
    (b"RIFF\0\0\0\0WAVEfmt \x10\0\0\0" + struct.pack("<hhiihh", 1, 2, 8000, 0, 4, 16), 12),
])
def test_header_errors_carry_the_reference_numbers(blob, errnr):
    with pytest.raises(W.WavFileError) as e:
        W.read_wav_header(io.BytesIO(blob))
    assert e.value.errnr == errnr


def test_sample_conversions():
    assert W.convert_block(bytes([0, 127, 128, 255]), BYTE_INPUT).tolist() == [-32640, -128, 128, 32640]
    assert W.convert_block(struct.pack("<4h", 1, -2, 32767, -32768), 0).tolist() == [1, -2, 32767, -32768]
    got = W.convert_block(bytes([0x01, 0x02, 0x03, 0xff, 0xff, 0xff, 0x00, 0x00, 0x80]), BYTE_INPUT + DWORD_INPUT)
    assert got.tolist() == [0x03020100, -256, -2147483648] and got.dtype == np.int32
    assert W.convert_block(struct.pack("<2i", 123456789, -5), QWORD_INPUT + DWORD_INPUT).tolist() == [123456789, -5]
    z = np.array([0.0, 0.5, -0.5, 0.25, 1.0, -1.0, 1.5, np.nan], np.float32)
    want = [0, 1073741824, -1073741824, 536870912, -2147483648, -2147483648, -2147483648, -2147483648]
    assert W.convert_block(z.tobytes(), FLOAT_INPUT + DWORD_INPUT).tolist() == want   # 0x7fffffff is 2^31 as a float


def test_reader_blocks_tail_and_writer_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    s = rng.integers(-30000, 30000, 3 * 4096 + 100).astype(np.int16)         # three whole blocks and 200 bytes more
    path = tmp_path / "a.wav"
    p = rcvr_payload(50_100_000, 12345)
    assert W.write_wav(path, s, 96000, 2, proprietary=(b"rcvr", p)) == 3
    r = W.WavReader(str(path))
    assert r.header.remember == REMEMBER_PERSEUS and r.header.rx_ad_speed == 96000 and r.header.file_block_bytes == 8192
    blocks = list(r.blocks())
    assert len(blocks) == 3 and np.array_equal(np.concatenate(blocks), s[:3 * 4096])
    r.close()
    r = W.WavReader(str(path))
    tail = list(r.blocks(clear_tail=True))
    assert np.array_equal(tail[0], blocks[0]) and np.array_equal(tail[2][:4096 - 150], blocks[2][:4096 - 150])
    assert not tail[2][4096 - 150:].any()                                    # 500 - 200 bytes of the last whole block
    r.close()
    # 24-bit file: 6144 file bytes per ring block
    raw = rng.integers(0, 256, 2 * 6144 + 10, dtype=np.uint8).tobytes()
    path2 = tmp_path / "b.wav"
    path2.write_bytes(riff(fmt_chunk(1, 2, 48000, 3), data=raw))
    r2 = W.WavReader(str(path2))
    b2 = list(r2.blocks())
    assert len(b2) == 2 and b2[0].dtype == np.int32 and b2[0].size == 2048
    assert np.array_equal(b2[1], W.convert_block(raw[6144:12288], BYTE_INPUT + DWORD_INPUT))


@pytest.mark.parametrize("name", ["wav_pcm16", "wav_pcm16_rcvr", "wav_pcm16_auxi", "wav_pcm24_ext", "wav_pcm8_mono", "wav_float32", "wav_pcm32_list"])
def test_wav_header_reader_equals_the_compiled_references(name):
    """tests/golden/filehdr.npz: the bytes of small .wav files (PCM 8 / 16 / 24 / 32 bit, float, an extensible fmt chunk, rcvr and auxi
    chunks, an unknown chunk to skip) and what the COMPILED REFERENCE's init_wavread (modesub.c:1022-1347, run head-less by
    oracle/ref_files.c) made of them; linrad_amd.wavfile.read_wav_header must make the same of the same bytes"""
    import io
    import json
    import os
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "filehdr.npz"))
    blob, ref = g[name + "__file"].tobytes(), json.loads(g[name + "__ref"].tobytes().decode())
    f = io.BytesIO(blob)
    from linrad_amd import wavfile
    h = wavfile.read_wav_header(f)
    assert (h.rx_input_mode, h.rx_ad_channels, h.rx_ad_speed) == (ref["rx_input_mode"], ref["rx_ad_channels"], ref["rx_ad_speed"])
    assert h.remember == ref["remember0"] and len(h.proprietary) == ref["remember1"]
    assert h.diskread_time == ref["diskread_time"] and h.passband_center == ref["passband_center"] and bool(h.freq_from_file) == bool(ref["freq_from_file"])
    assert h.data_offset == ref["file_pos"] == f.tell()
