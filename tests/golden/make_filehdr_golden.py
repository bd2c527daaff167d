#!/usr/bin/env python3
"""Golden of the recorded-IQ file headers from the COMPILED REFERENCE's own readers (oracle/_ref/ref_files = modesub.c compiled where it
lies + a head-less caller): open_savefile (modesub.c:606-733) on .raw files, init_wavread (modesub.c:1022-1347) on .wav files.
Each entry of tests/golden/filehdr.npz = the bytes of a small file (header + two blocks of samples) and what the reference left in its
globals after reading it (JSON).  Data only.  tests/test_rawfile_cpu.py / test_wavfile_cpu.py hold linrad_amd/rawfile.py and wavfile.py to
these answers.  usage: python tests/golden/make_filehdr_golden.py"""
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from linrad_amd import rawfile, wavfile  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_files")


def wav_bytes(fmt_tag, channels, rate, bits, chunks=(), fmt_extra=b"", ndata=256, tail=b""):
    align = channels * bits // 8
    fmt = struct.pack("<hhiihh", fmt_tag, channels, rate, rate * align, align, bits) + fmt_extra
    body = b"WAVEfmt " + struct.pack("<i", len(fmt)) + fmt
    for name, payload in chunks:
        body += name + struct.pack("<i", len(payload)) + payload
    data = bytes((i * 7) & 255 for i in range(ndata * align))
    body += b"data" + struct.pack("<i", len(data)) + data + tail
    return b"RIFF" + struct.pack("<i", len(body)) + body


def cases():
    iq = (np.arange(2 * 8192, dtype=np.int32) * 37 % 20011 - 10000).astype(np.int16)
    perseus = struct.pack("<IIq", 144_300_000, 4, 1_700_000_000) + struct.pack("<H4B", 1, 0, 1, 0, 0) + bytes(32)      # RCVR behind chunkSize (modesub.c:113-127)
    systemtime = struct.pack("<8H", 2024, 5, 3, 15, 13, 7, 42, 500)                                      # wYear .. wMilliseconds
    auxi = systemtime + systemtime + struct.pack("<7I", 14_070_000, 66_666_667, 0, 190_000, 0, 0, 0)      # AUXI behind chunkSize (modesub.c:130-141)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        def raw(name, h, samples):
            p = os.path.join(td, name)
            rawfile.write_raw(p, h, samples)
            out["raw_" + name] = open(p, "rb").read()
        raw("plain", rawfile.RawHeader(rx_input_mode=rawfile.IQ_DATA, rx_rf_channels=1, rx_ad_channels=2, rx_ad_speed=96000, remember=rawfile.REMEMBER_NOTHING,
                                       diskread_time=86399.25, passband_center=14.07, passband_direction=1), iq)
        raw("perseus_rev", rawfile.RawHeader(rx_input_mode=rawfile.IQ_DATA, rx_rf_channels=1, rx_ad_channels=2, rx_ad_speed=2_000_000, remember=rawfile.REMEMBER_PERSEUS,
                                             proprietary=perseus, diskread_time=1234.5, passband_center=144.3, passband_direction=-1), iq)
        raw("sdr14_dword", rawfile.RawHeader(rx_input_mode=rawfile.IQ_DATA | rawfile.DWORD_INPUT, rx_rf_channels=1, rx_ad_channels=2, rx_ad_speed=190_000,
                                             remember=rawfile.REMEMBER_SDR14, proprietary=auxi, diskread_time=47587.0, passband_center=14.07, passband_direction=1),
            (iq.astype(np.int32) << 14))
        raw("twochan", rawfile.RawHeader(rx_input_mode=rawfile.IQ_DATA, rx_rf_channels=2, rx_ad_channels=4, rx_ad_speed=48000, remember=rawfile.REMEMBER_UNKNOWN,
                                         diskread_time=0.0, passband_center=0.0, passband_direction=1), iq)
        raw("real_mono", rawfile.RawHeader(rx_input_mode=0, rx_rf_channels=1, rx_ad_channels=1, rx_ad_speed=44100, remember=rawfile.REMEMBER_NOTHING,
                                           diskread_time=10.0, passband_center=0.0125, passband_direction=1), iq)
    # the original format: ui.rx_input_mode is the first int of the file (modesub.c:661-667, 714-722)
    out["raw_oldformat"] = struct.pack("<iiiiB", rawfile.IQ_DATA, 1, 2, 8000, 0) + iq[:4096].tobytes()
    out["wav_pcm16"] = wav_bytes(1, 2, 96000, 16)
    out["wav_pcm16_rcvr"] = wav_bytes(1, 2, 2_000_000, 16, chunks=[(b"rcvr", perseus[:40])])
    out["wav_pcm16_auxi"] = wav_bytes(1, 2, 190_000, 16, chunks=[(b"auxi", auxi)])
    out["wav_pcm24_ext"] = wav_bytes(1, 2, 192000, 24, fmt_extra=struct.pack("<hhi", 22, 24, 3) + bytes(16))      # WAVE_FORMAT_EXTENSIBLE-sized fmt chunk
    out["wav_pcm8_mono"] = wav_bytes(1, 1, 8000, 8)
    out["wav_float32"] = wav_bytes(3, 2, 48000, 32)
    out["wav_pcm32_list"] = wav_bytes(1, 2, 48000, 32, chunks=[(b"LIST", b"INFOISFT" + struct.pack("<i", 6) + b"linrad")])   # an unknown chunk is skipped
    return out


def main():
    if not os.path.exists(REF):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, blob in cases().items():
            p = os.path.join(td, name + (".raw" if name.startswith("raw_") else ".wav"))
            open(p, "wb").write(blob)
            r = subprocess.run([REF, "raw" if name.startswith("raw_") else "wav", p], capture_output=True, text=True)
            assert r.returncode == 0, (name, r.stderr)
            ans = json.loads(r.stdout.strip().splitlines()[-1])
            assert ans["rc"] == 0, (name, ans, r.stderr)
            out[name + "__file"] = np.frombuffer(blob, np.uint8)
            out[name + "__ref"] = np.frombuffer(json.dumps(ans, sort_keys=True).encode(), np.uint8)
            print(name, ans)
    path = os.path.join(os.environ.get("LRH_GOLDEN_OUT", HERE), "filehdr.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
