"""Loader for liblinrad_hip.so -- the HIP product.  There is no CPU fallback: a missing library is an error."""
import ctypes as C
import os

import numpy as np

from .abi import LrhSynth, StageAPI

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblinrad_hip.so")
_lib = None


def hip_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                f"g.build()' or make -C linrad_amd/csrc).  linrad_amd has no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _lib.lrh_last_error.restype = C.c_char_p
        _lib.lrh_last_error.argtypes = [C.c_void_p]
    return _lib


class HipReceiver(StageAPI):
    """The wideband chain on one MI355X behind the reference's stage names (see StageAPI)."""

    def __init__(self, cfg):
        lib = hip_lib()
        try:
            super().__init__(lib, "lrh", cfg)
        except Exception as e:  # noqa: BLE001
            raise RuntimeError(f"lrh_open failed ({e}); a HIP device and liblinrad_hip.so are required") from e
        vp = C.c_void_p
        lib.lrh_sync.argtypes, lib.lrh_sync.restype = [vp], C.c_int
        lib.lrh_timer_start.argtypes, lib.lrh_timer_start.restype = [vp], C.c_int
        lib.lrh_timer_stop.argtypes, lib.lrh_timer_stop.restype = [vp, C.POINTER(C.c_float)], C.c_int
        lib.lrh_profile_enable.argtypes, lib.lrh_profile_enable.restype = [vp, C.c_int], C.c_int
        lib.lrh_profile_get.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_long)]
        lib.lrh_profile_get.restype = C.c_int

    def _chk(self, rc, what):
        if rc != 0:
            msg = self.lib.lrh_last_error(self.ctx)
            raise RuntimeError(f"lrh_{what} rc={rc}: {msg.decode() if msg else ''}")

    def export_device(self, ring, dst_ptr, offset, count):
        self.lib.lrh_export_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t]
        self.lib.lrh_export_device.restype = C.c_int
        self._chk(self.lib.lrh_export_device(self.ctx, ring, C.c_void_p(dst_ptr), offset, count), "export_device")

    def export_device_async(self, ring, dst_ptr, offset, count):
        """Device-to-device copy ordered on the context's stream, no host wait (pair it with stream_handle())."""
        self.lib.lrh_export_device_async.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t]
        self.lib.lrh_export_device_async.restype = C.c_int
        self._chk(self.lib.lrh_export_device_async(self.ctx, ring, C.c_void_p(dst_ptr), offset, count), "export_device_async")

    def stream_handle(self):
        """hipStream_t of the context as an integer (torch.cuda.ExternalStream takes it)."""
        self.lib.lrh_stream.argtypes, self.lib.lrh_stream.restype = [C.c_void_p], C.c_void_p
        return int(self.lib.lrh_stream(self.ctx) or 0)

    def sync(self):
        self._chk(self.lib.lrh_sync(self.ctx), "sync")

    def flush(self):
        """issue what lrh_wideband_dsp still holds back from its last round, without waiting (lrh_flush): afterwards the context's stream
        carries everything the calls so far have produced -- what a consumer ordered only on that stream needs"""
        self.lib.lrh_flush.argtypes, self.lib.lrh_flush.restype = [C.c_void_p], C.c_int
        self._chk(self.lib.lrh_flush(self.ctx), "flush")

    def timer_start(self):
        self._chk(self.lib.lrh_timer_start(self.ctx), "timer_start")

    def timer_stop(self):
        ms = C.c_float()
        self._chk(self.lib.lrh_timer_stop(self.ctx, C.byref(ms)), "timer_stop")
        return ms.value

    def profile_enable(self, on=True):
        self._chk(self.lib.lrh_profile_enable(self.ctx, int(on)), "profile_enable")

    def profile_get(self, kernel):
        ms, n = C.c_double(), C.c_long()
        self._chk(self.lib.lrh_profile_get(self.ctx, kernel.encode(), C.byref(ms), C.byref(n)), "profile_get")
        return ms.value, n.value


def open_hip(cfg):
    return HipReceiver(cfg)


def synth_defaults(fft1_size, channel=0):
    lib = hip_lib()
    s = LrhSynth()
    lib.lrh_synth_defaults.argtypes, lib.lrh_synth_defaults.restype = [C.POINTER(LrhSynth), C.c_int, C.c_int], None
    lib.lrh_synth_defaults(C.byref(s), fft1_size, channel)
    return s


def synth_iq(s, first_sample, nsamples):
    """Deterministic synthetic int16 I/Q (host generator of the C ABI)."""
    lib = hip_lib()
    lib.lrh_synth_iq.argtypes = [C.POINTER(LrhSynth), C.c_int64, C.c_int64, C.POINTER(C.c_int16)]
    lib.lrh_synth_iq.restype = C.c_int
    out = np.empty(2 * nsamples, np.int16)
    rc = lib.lrh_synth_iq(C.byref(s), first_sample, nsamples, out.ctypes.data_as(C.POINTER(C.c_int16)))
    if rc != 0:
        raise RuntimeError(f"lrh_synth_iq rc={rc}")
    return out
