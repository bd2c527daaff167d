"""Loads oracle/liblinrad_oracle.so (CPU restatement, test infrastructure) behind the product's own StageAPI, and
oracle/liblinrad_oracle64.so -- the same source compiled with every float a double (-DLRO_F64) -- as the float64 TRUTH both float32
implementations (the compiled reference's goldens, the HIP path) are measured against where a relative tolerance alone cannot decide
(tests/paritylib.py: truth_gate)."""
import ctypes as C
import os
import subprocess

import numpy as np

from linrad_amd import abi
from linrad_amd.abi import StageAPI

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_lib = {}


def _load(so):
    if so not in _lib:
        path = os.path.join(ORACLE_DIR, so)
        src = os.path.join(ORACLE_DIR, "linrad_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
        _lib[so] = C.CDLL(path)
    return _lib[so]


def oracle_lib():
    return _load("liblinrad_oracle.so")


def open_oracle(cfg):
    return StageAPI(oracle_lib(), "lro", cfg)


class TruthAPI(StageAPI):
    """the float64 build behind the same calls: float32 arrays cross the ABI (converted at the boundary); export_f64 hands out a ring
    unrounded, and every waterfall line fetched through export() leaves its values BEFORE the truncation to short in wf_pre_lines"""

    def __init__(self, cfg):
        lib = _load("liblinrad_oracle64.so")
        super().__init__(lib, "lro", cfg)
        dp = C.POINTER(C.c_double)
        lib.lro_export_f64.argtypes, lib.lro_export_f64.restype = [C.c_void_p, C.c_int, dp, C.c_size_t, C.c_size_t], C.c_int
        lib.lro_export_wf_pre.argtypes, lib.lro_export_wf_pre.restype = [C.c_void_p, dp, C.c_size_t, C.c_size_t], C.c_int
        self.wf_pre_lines = []

    def export_f64(self, ring, offset=0, count=None):
        if count is None:
            count = self.ring_size(ring) - offset
        out = np.zeros(count, np.float64)
        self._chk(self.lib.lro_export_f64(self.ctx, ring, out.ctypes.data_as(C.POINTER(C.c_double)), offset, count), "export_f64")
        return out

    def export_wf_pre(self, offset=0, count=None):
        """waterfall values before the truncation to short, same places as RING_WG_WATERF"""
        if count is None:
            count = self.ring_size(abi.RING_WG_WATERF) - offset
        pre = np.zeros(count, np.float64)
        self._chk(self.lib.lro_export_wf_pre(self.ctx, pre.ctypes.data_as(C.POINTER(C.c_double)), offset, count), "export_wf_pre")
        return pre

    def export(self, ring, offset=0, count=None):
        if ring == abi.RING_WG_WATERF and count is not None:
            self.wf_pre_lines.append(self.export_wf_pre(offset, count))
        if ring in (abi.RING_TIMF1, abi.RING_WG_WATERF) or ring >= abi.RING_FFT2_XYPOWER:
            return super().export(ring, offset, count)
        return self.export_f64(ring, offset, count)          # run_case's ring dump: unrounded


def open_truth(cfg):
    return TruthAPI(cfg)
