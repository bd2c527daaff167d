"""Recorded-IQ ingest: `.wav` files, host-side mirror of the reference's reader.

* header: `init_wavread` (modesub.c:1088-1347) -- RIFF/WAVE/"fmt " in this fixed order, PCM (tag 1) or IEEE float
  (tag 3), 1 or 2 channels, 8/16/24/32 bit; then chunks up to "data": Perseus `rcvr` (modesub.c:113-126, 1210-1229),
  SpectraVue / SDR-14 `auxi` (modesub.c:128-142, 1231-1316) in its binary form and in the UTF-16 XML form SDR Console
  writes, ExpertSDR2 `esdr` (modesub.c:1319-1326); any other chunk is skipped by its size (no padding byte, as there).
  Failures carry the reference's error number ("Error in .wav file header [n]").
* samples: the conversions of `rx_file_input` (rxin.c:1572-1640) into the timf1 ring formats the chain knows:
  8 bit -> int16 `(b << 8) - 32640`; 16 bit as is; 24 bit -> int32 left-justified; 32-bit int as is;
  float -> int32 `0x7fffffff * z` with the x86 conversion's out-of-range result.
* writer: the 44-byte header of `write_wav_header` (modesub.c:1768-1882) with an optional proprietary chunk, for tests
  and for recordings made from device rings.

Parity: restated from the reference source; the reader sits inside GUI code that cannot be linked head-less, so it is
pinned by known-answer files built field by field from the format definitions (tests/test_wavfile_cpu.py) --
"parity unpinned" in the sense of DESIGN.md.  The short last block of a file is dropped like RawReader does; the
reference additionally zeroes the 500 bytes before the end of the data (rxin.c:1676-1684), which `blocks(clear_tail=True)`
reproduces for the delivered blocks.
"""
import struct
from dataclasses import dataclass

import numpy as np

from .rawfile import (BLOCK_BYTES, BYTE_INPUT, DIGITAL_IQ, DWORD_INPUT, FLOAT_INPUT, QWORD_INPUT, REMEMBER_NOTHING,
                      REMEMBER_PERSEUS, REMEMBER_SDR14)

RCVR_MAX = 64            # sizeof(RCVR) on x86-64 (time_t 8 bytes, padded to 64): the size test of modesub.c:1221
AUXI_FIXED = 60          # sizeof(AUXI) - 8: two SYSTEMTIME + seven DWORD, modesub.c:1255
XML_VERSION = b'<?xml version="1.0"?>'            # modesub.c:1034
CONSOLE_TIME, CONSOLE_FREQ = b"CurrentTimeUTC=", b"RadioCenterFreq="     # modesub.c:1030-1032


class WavFileError(ValueError):
    def __init__(self, errnr, msg=""):
        super().__init__(f"Error in .wav file header [{errnr}]" + (f": {msg}" if msg else ""))
        self.errnr = errnr


@dataclass
class WavHeader:
    format_tag: int = 1
    rx_ad_channels: int = 2
    rx_ad_speed: int = 96000
    rx_input_mode: int = 0            # BYTE/DWORD/FLOAT/QWORD_INPUT (+DIGITAL_IQ for ExpertSDR2); IQ_DATA is the operator's choice
    remember: int = REMEMBER_NOTHING
    proprietary: bytes = b""
    diskread_time: float = 0.0
    passband_center: float = 0.0      # MHz
    freq_from_file: bool = False
    expert_sdr2: bool = False
    data_offset: int = 0

    @property
    def dword(self):
        return bool(self.rx_input_mode & DWORD_INPUT)

    @property
    def file_block_bytes(self):
        """file bytes behind one 8192-byte ring block (rxin.c:1578, 1601)"""
        if self.rx_input_mode == BYTE_INPUT:
            return BLOCK_BYTES // 2
        if (self.rx_input_mode & (BYTE_INPUT | DWORD_INPUT)) == (BYTE_INPUT | DWORD_INPUT):
            return 3 * (BLOCK_BYTES // 4)
        return BLOCK_BYTES


def read_wav_header(f):
    """init_wavread's header walk; `f` is left at the first sample."""
    def rd(n, errnr):
        b = f.read(n)
        if len(b) != n:
            raise WavFileError(errnr)
        return b

    h = WavHeader()
    if rd(4, 0) != b"RIFF":
        raise WavFileError(1)
    rd(4, 2)                                                   # file size: not needed
    if f.read(4) != b"WAVE":
        raise WavFileError(2)
    if f.read(4) != b"fmt ":
        raise WavFileError(3)
    chunk_size, = struct.unpack("<i", rd(4, 4))
    tag_b = f.read(2)
    if len(tag_b) != 2 or struct.unpack("<h", tag_b)[0] not in (1, 3):
        raise WavFileError(5, "Unknown wFormatTag.")
    h.format_tag, = struct.unpack("<h", tag_b)
    ch_b = f.read(2)
    if len(ch_b) != 2 or not 1 <= struct.unpack("<h", ch_b)[0] <= 2:
        raise WavFileError(6)
    h.rx_ad_channels, = struct.unpack("<h", ch_b)
    h.rx_ad_speed, = struct.unpack("<i", rd(4, 7))
    rd(4, 8)                                                   # average bytes per second
    align, = struct.unpack("<h", rd(2, 9))
    per_component = align // h.rx_ad_channels
    if per_component == 1:
        if h.format_tag != 1:
            raise WavFileError(10)
        h.rx_input_mode = BYTE_INPUT
    elif per_component == 2:
        if h.format_tag != 1:
            raise WavFileError(11)
        h.rx_input_mode = 0
    elif per_component == 3:
        if h.format_tag != 1:
            raise WavFileError(12)
        h.rx_input_mode = BYTE_INPUT + DWORD_INPUT
    elif per_component == 4:
        h.rx_input_mode = (QWORD_INPUT if h.format_tag == 1 else FLOAT_INPUT) + DWORD_INPUT
    else:
        raise WavFileError(9)
    chunk_size -= 14                                           # bits per sample and any extension: skipped unseen
    while True:
        # skip_chunk (modesub.c:1193-1207): whatever was skipped, the remembered proprietary chunk type is reset
        if chunk_size is not None:
            h.remember = REMEMBER_NOTHING
        else:
            chunk_size = 0
        if chunk_size < 0:
            raise WavFileError(11)                             # the byte-wise skip of modesub.c:1195-1200 never ends on it
        rd(chunk_size, 11)
        name = rd(4, 12)
        if name == b"rcvr":
            if h.remember != REMEMBER_NOTHING:
                raise WavFileError(13, "second proprietary chunk")          # lirerr(1522318)
            chunk_size, = struct.unpack("<i", rd(4, 13))
            if chunk_size > RCVR_MAX:
                raise WavFileError(13)
            h.remember, h.proprietary = REMEMBER_PERSEUS, rd(chunk_size, 13)
            p = h.proprietary.ljust(RCVR_MAX, b"\0")
            h.passband_center = 0.000001 * struct.unpack_from("<I", p, 0)[0]
            h.diskread_time = float(struct.unpack_from("<q", p, 8)[0])         # time_t timeStart
            h.freq_from_file = True
            chunk_size = None                                  # next_chunk: no skip, nothing reset
            continue
        if name == b"auxi":
            if h.remember != REMEMBER_NOTHING:
                raise WavFileError(14, "second proprietary chunk")          # lirerr(1522319)
            chunk_size, = struct.unpack("<i", rd(4, 14))
            raw = rd(chunk_size, 14)
            text = raw[0:1] + raw[2::2]                        # "Remove the zeroes": UTF-16LE text to bytes
            if text.startswith(XML_VERSION):
                text = text[:chunk_size // 2]
                i = text.find(CONSOLE_TIME, 0, max(0, len(text) - 1))
                if i < 0:
                    raise WavFileError(14)
                i = text.index(b" ", i + len(CONSOLE_TIME)) + 1
                hh, mm, ss = (int(v) for v in _scan_ints(text[i:], 3, b":"))
                h.diskread_time = 3600.0 * hh + 60 * mm + ss
                i = text.find(CONSOLE_FREQ, 0, max(0, len(text) - 1))
                if i < 0:
                    raise WavFileError(14)
                hz = _scan_ints(text[i + len(CONSOLE_FREQ) + 1:], 1, b"")[0]
                h.passband_center = 0.000001 * float(np.float32(hz))
            else:
                h.remember, h.proprietary = REMEMBER_SDR14, raw
                p = raw.ljust(AUXI_FIXED, b"\0")
                hour, minute, second = struct.unpack_from("<3H", p, 8)     # SYSTEMTIME StartTime: wHour, wMinute, wSecond
                h.diskread_time = hour * 3600.0 + minute * 60.0 + second
                h.passband_center = 0.000001 * struct.unpack_from("<I", p, 32)[0]
            h.freq_from_file = True
            chunk_size = None
            continue
        if name == b"esdr":
            h.expert_sdr2 = True
            chunk_size, = struct.unpack("<i", rd(4, 15))
            continue
        if name != b"data":
            chunk_size, = struct.unpack("<i", rd(4, 25))       # unknown: get the size and skip
            continue
        rd(4, 25)                                              # data size: not needed
        if h.expert_sdr2:
            rd(1, 25)                                          # SunSDR2 files carry one more byte
            h.rx_input_mode |= DIGITAL_IQ
        h.data_offset = f.tell()
        return h


def _scan_ints(b, count, sep):
    """sscanf("%d:%d:%d") / ("%ld"): leading blanks, optional sign, digits"""
    out, i = [], 0
    for n in range(count):
        while i < len(b) and b[i:i + 1] in b" \t\n\r":
            i += 1
        j = i + 1 if b[i:i + 1] in (b"-", b"+") else i
        while j < len(b) and b[j:j + 1].isdigit():
            j += 1
        if j == i or not b[j - 1:j].isdigit():
            raise WavFileError(14)
        out.append(int(b[i:j]))
        i = j
        if n < count - 1:
            if b[i:i + 1] != sep:
                raise WavFileError(14)
            i += 1
    return out


def convert_block(raw, rx_input_mode):
    """file bytes of one block -> timf1 ring samples (int16, or int32 for DWORD modes), rxin.c:1572-1640"""
    raw = np.frombuffer(bytes(raw), np.uint8)
    mode = rx_input_mode & (BYTE_INPUT | DWORD_INPUT | FLOAT_INPUT | QWORD_INPUT)
    if mode == BYTE_INPUT:                                     # (0,255) -> (-127.5,127.5), 16-bit scale
        return ((raw.astype(np.int32) << 8) - 32640).astype(np.int16)
    if mode == 0:
        return raw.view(np.int16).copy()
    if mode == BYTE_INPUT + DWORD_INPUT:                       # 24-bit PCM, left-justified with a zero low byte
        t = raw.reshape(-1, 3).astype(np.uint32)
        return ((t[:, 0] << 8) | (t[:, 1] << 16) | (t[:, 2] << 24)).astype(np.uint32).view(np.int32)
    if mode == QWORD_INPUT + DWORD_INPUT:
        return raw.view(np.int32).copy()
    if mode == FLOAT_INPUT + DWORD_INPUT:
        with np.errstate(invalid="ignore", over="ignore"):
            v = (np.float32(0x7fffffff) * raw.view(np.float32)).astype(np.float32)
            ok = np.isfinite(v) & (v >= -2147483648.0) & (v < 2147483648.0)
            out = np.where(ok, np.trunc(np.where(ok, v, 0)), -2147483648.0).astype(np.int64)
        return out.astype(np.int32)                            # cvttss2si: 0x80000000 when out of range
    raise WavFileError(9, "unsupported sample format")


class WavReader:
    """Iterates the blocks of a .wav recording and feeds a receiver's timf1 ring (rx_file_input, rxin.c:1560-1690)."""

    def __init__(self, path):
        self.f = open(path, "rb")
        self.header = read_wav_header(self.f)

    def close(self):
        self.f.close()

    def blocks(self, clear_tail=False):
        """ring-format sample arrays, 8192 ring bytes each; the short last block is dropped.  clear_tail: zero the last
        500 ring bytes in front of the end of the data, where WAV writers append non-sample bytes (rxin.c:1676-1684)."""
        n, mode = self.header.file_block_bytes, self.header.rx_input_mode
        prev = None
        while True:
            b = self.f.read(n)
            if len(b) != n:
                if prev is not None:
                    if clear_tail:
                        got = len(b) * BLOCK_BYTES // n          # ring bytes of the partial read (rxin.c:1585, 1611)
                        if got < 500:
                            prev = prev.copy()
                            prev.view(np.uint8)[BLOCK_BYTES - (500 - got):] = 0
                    yield prev
                return
            if prev is not None:
                yield prev
            prev = convert_block(b, mode)

    def feed(self, rx, byte_offset=0, max_blocks=None, clear_tail=False):
        """Write blocks into rx's timf1 ring from byte_offset on; returns the number of ring bytes written."""
        if bool(rx.cfg.timf1_dword_input) != self.header.dword:
            raise WavFileError(9, "receiver input format does not match the recording")
        mask, written = rx.cfg.timf1_bytes - 1, 0
        for i, blk in enumerate(self.blocks(clear_tail)):
            if max_blocks is not None and i >= max_blocks:
                break
            rx.timf1_write(blk, (byte_offset + written) & mask)
            written += BLOCK_BYTES
        return written


def write_wav(path, samples, rate, channels=2, proprietary=None):
    """16-bit PCM file the way write_wav_header lays it out (modesub.c:1768-1882); `proprietary` = (b"rcvr"|b"auxi",
    payload) is written between the format and the data chunk like the reference does when it replays a recording."""
    s = np.ascontiguousarray(samples, np.int16)
    extra = b""
    if proprietary is not None:
        extra = proprietary[0] + struct.pack("<i", len(proprietary[1])) + proprietary[1]
    filesize = 44 + len(extra) + s.nbytes
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<i", min(filesize - 8, 0x7fffffff)) + b"WAVEfmt " + struct.pack("<i", 16))
        f.write(struct.pack("<hhiihh", 1, channels, rate, rate * 2 * channels, 2 * channels, 16))
        f.write(extra)
        f.write(b"data" + struct.pack("<i", min(filesize - 44, 0x7fffffff)))     # the reference's size field: file size - 44
        f.write(s.tobytes())
    return s.nbytes // BLOCK_BYTES
