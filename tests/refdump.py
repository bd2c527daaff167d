"""Reader for the container files written by oracle/ref_harness.c (test infrastructure)."""
import struct
import numpy as np

_DT = {"f4": np.float32, "i4": np.int32, "u2": np.uint16, "i2": np.int16, "u4": np.uint32, "f8": np.float64}


def load_dump(path):
    out = {}
    with open(path, "rb") as f:
        while True:
            h = f.read(44)
            if len(h) < 44:
                break
            name = h[:32].split(b"\0")[0].decode()
            dt = h[32:36].split(b"\0")[0].decode()
            (cnt,) = struct.unpack("<Q", h[36:44])
            dtype = np.dtype(_DT[dt])
            out[name] = np.frombuffer(f.read(cnt * dtype.itemsize), dtype=dtype).copy()
    return out
