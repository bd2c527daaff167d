"""Recorded-IQ ingest: Linrad `.raw` files.

Host-side mirror of the reference's reader: `open_savefile` (modesub.c:606-733: header), `skip_calibration`
(modesub.c:246-299: embedded calibration blocks), `rx_file_input` (rxin.c:1560-1650: block reads) and, for writing test
recordings, the header writer of modesub.c:1515-1600 with `compress_rawdat_disk` (csplit.c:76-118).  Sample formats:
16-bit int16 I/Q as is; DWORD_INPUT recordings store 18 bits per component, 9 bytes per four components, which the
device expands (`lrh_timf1_write_packed18`) exactly like `expand_rawdat` (csplit.c:20-73).

The header layout is restated from the reference source; the 18-bit packing is pinned against the compiled
reference (tests/golden/rawdat_18bit.npz).  WAV input (`rcvr`/`auxi` chunks): linrad_amd/wavfile.py.
"""
import struct
from dataclasses import dataclass, field

import numpy as np

# ui.rx_input_mode bits, globdef.h:277-285
DWORD_INPUT, TWO_CHANNELS, IQ_DATA, BYTE_INPUT, NO_DUPLEX, DIGITAL_IQ, FLOAT_INPUT, QWORD_INPUT, MODEPARM_MAX = \
    1, 2, 4, 8, 16, 32, 64, 128, 256
# first int of the file when negative, modesub.c:88-91
REMEMBER_UNKNOWN, REMEMBER_NOTHING, REMEMBER_PERSEUS, REMEMBER_SDR14 = -1, -2, -3, -4
BLOCK_BYTES = 8192           # snd[RXAD].block_bytes for file input (modesub.c:357)


class RawFileError(ValueError):
    pass


@dataclass
class RawHeader:
    rx_input_mode: int = IQ_DATA
    rx_rf_channels: int = 1
    rx_ad_channels: int = 2
    rx_ad_speed: int = 96000
    remember: int = REMEMBER_UNKNOWN
    proprietary: bytes = b""
    diskread_time: float = 0.0
    passband_center: float = 0.0
    passband_direction: int = 1
    save_init_flag: int = 0
    calibration: bytes = field(default=b"", repr=False)     # raw bytes of the embedded calibration blocks, if any
    data_offset: int = 0

    @property
    def dword(self):
        return bool(self.rx_input_mode & DWORD_INPUT)

    @property
    def save_rw_bytes(self):
        """Bytes per block in the file (buf.c:597-599, modesub.c:365-372)."""
        return 18 * BLOCK_BYTES // 32 if self.dword else BLOCK_BYTES


def _rd(f, fmt):
    n = struct.calcsize(fmt)
    b = f.read(n)
    if len(b) != n:
        raise RawFileError("File corrupted.")           # errfile_2, modesub.c:665-667
    return struct.unpack("<" + fmt, b)


def skip_calibration(f, save_init_flag):
    """modesub.c:246-299: returns the raw bytes skipped."""
    start = f.tell()
    if save_init_flag & 1:
        rdbuf = _rd(f, "10i")
        mm = rdbuf[1] if rdbuf[7] == 0 else rdbuf[7]
        need = mm * 4 * (2 * rdbuf[6] + 1)
        if len(f.read(need)) != need or _rd(f, "10i") != rdbuf:
            raise RawFileError("ERROR. File corrupted")
    if save_init_flag & 2:
        rdbuf = _rd(f, "10i")
        need = rdbuf[3] * rdbuf[0] * 4 * 8
        if len(f.read(need)) != need or _rd(f, "10i") != rdbuf:
            raise RawFileError("ERROR. File corrupted")
    end = f.tell()
    f.seek(start)
    blob = f.read(end - start)
    return blob


def read_header(f):
    """open_savefile, modesub.c:661-733 (+ skip_calibration).  Leaves `f` at the first data block."""
    h = RawHeader()
    (first,) = _rd(f, "i")
    if first < 0:
        h.remember = first
        if first in (REMEMBER_PERSEUS, REMEMBER_SDR14):
            (n,) = _rd(f, "i")
            h.proprietary = f.read(n)
            if len(h.proprietary) != n:
                raise RawFileError("File corrupted.")
        elif first not in (REMEMBER_UNKNOWN, REMEMBER_NOTHING):
            raise RawFileError("This Linrad version is too old")
        h.diskread_time, h.passband_center = _rd(f, "dd")
        (h.passband_direction,) = _rd(f, "i")
        if abs(h.passband_direction) != 1:
            raise RawFileError("File corrupted.")
        (h.rx_input_mode,) = _rd(f, "i")
    else:                                               # original format: the first item is rx_input_mode
        h.remember = REMEMBER_NOTHING
        h.rx_input_mode = first
    if h.rx_input_mode >= MODEPARM_MAX:
        raise RawFileError("File corrupted.")
    h.rx_rf_channels, h.rx_ad_channels, h.rx_ad_speed = _rd(f, "iii")
    if h.rx_rf_channels == 2:
        h.rx_input_mode |= TWO_CHANNELS
    if not 1 <= h.rx_ad_channels <= 4 or h.rx_ad_channels not in (h.rx_rf_channels, 2 * h.rx_rf_channels):
        raise RawFileError("File corrupted.")
    (h.save_init_flag,) = _rd(f, "B")
    h.calibration = skip_calibration(f, h.save_init_flag)
    h.data_offset = f.tell()
    return h


def write_header(f, h):
    """modesub.c:1515-1600 (calibration blocks are copied verbatim when present)."""
    f.write(struct.pack("<i", h.remember))
    if h.remember in (REMEMBER_PERSEUS, REMEMBER_SDR14):
        f.write(struct.pack("<i", len(h.proprietary)) + h.proprietary)
    f.write(struct.pack("<ddi", h.diskread_time, h.passband_center, h.passband_direction))
    f.write(struct.pack("<i", h.rx_input_mode & (TWO_CHANNELS + DWORD_INPUT + IQ_DATA + DIGITAL_IQ)))
    f.write(struct.pack("<iiiB", h.rx_rf_channels, h.rx_ad_channels, h.rx_ad_speed, h.save_init_flag))
    f.write(h.calibration)


def compress_rawdat(samples):
    """compress_rawdat_disk, csplit.c:76-118: int32 components (multiple of 4) -> 9 bytes per 4, as uint8.

    As in the reference the bottom two bits of the fourth component come from byte 3 of the group (its `n=timf1_char[i+3]`,
    csplit.c:109), not from byte 13 -- a recording made by Linrad therefore carries sample 0's top byte bits there."""
    b = np.ascontiguousarray(samples, np.int32).view(np.uint8).reshape(-1, 16)
    out = np.empty((b.shape[0], 9), np.uint8)
    for k in range(4):
        out[:, 2 * k] = b[:, 4 * k + 2]
        out[:, 2 * k + 1] = b[:, 4 * k + 3]
    m = b[:, 1] >> 2
    m = (m | (b[:, 5] & 0xc0)) >> 2
    m = (m | (b[:, 9] & 0xc0)) >> 2
    m = m | (b[:, 3] & 0xc0)
    out[:, 8] = m
    return out.reshape(-1)


def expand_rawdat(packed):
    """expand_rawdat, csplit.c:20-73, on the host (numpy) -- reference for tests and for feeding the oracle."""
    r = np.ascontiguousarray(packed, np.uint8).reshape(-1, 9)
    out = np.zeros((r.shape[0], 16), np.uint8)
    m = r[:, 8].astype(np.uint8)
    for k in range(4):
        out[:, 4 * k + 1] = (m & 0xc0) | 0x20
        out[:, 4 * k + 2] = r[:, 2 * k]
        out[:, 4 * k + 3] = r[:, 2 * k + 1]
        m = (m << 2).astype(np.uint8)
    return out.reshape(-1).view(np.int32)


class RawReader:
    """Iterates the blocks of a recording and feeds a receiver's timf1 ring (rx_file_input, rxin.c:1560-1650)."""

    def __init__(self, path):
        self.f = open(path, "rb")
        self.header = read_header(self.f)

    def close(self):
        self.f.close()

    def blocks(self):
        n = self.header.save_rw_bytes
        while True:
            b = self.f.read(n)
            if len(b) != n:                             # end_savfile: a short last block is dropped
                return
            yield np.frombuffer(b, np.uint8)

    def feed(self, rx, byte_offset=0, max_blocks=None):
        """Write blocks into rx's timf1 ring from byte_offset on; returns the number of ring bytes written."""
        if bool(rx.cfg.timf1_dword_input) != self.header.dword:
            raise RawFileError("receiver input format does not match the recording")
        mask, written = rx.cfg.timf1_bytes - 1, 0
        for i, blk in enumerate(self.blocks()):
            if max_blocks is not None and i >= max_blocks:
                break
            if self.header.dword:
                rx.timf1_write_packed18(blk, (byte_offset + written) & mask)
            else:
                rx.timf1_write(blk.view(np.int16), (byte_offset + written) & mask)
            written += BLOCK_BYTES
        return written


def write_raw(path, header, samples):
    """Write a recording: `samples` = interleaved I/Q components (int16, or int32 left-justified for DWORD_INPUT);
    only whole 8192-byte blocks are written, like write_raw_file (rxin.c:626-660)."""
    with open(path, "wb") as f:
        write_header(f, header)
        if header.dword:
            s = np.ascontiguousarray(samples, np.int32)
            nblk = s.nbytes // BLOCK_BYTES
            f.write(compress_rawdat(s[:nblk * BLOCK_BYTES // 4]).tobytes())
        else:
            s = np.ascontiguousarray(samples, np.int16)
            nblk = s.nbytes // BLOCK_BYTES
            f.write(s[:nblk * BLOCK_BYTES // 2].tobytes())
    return nblk
