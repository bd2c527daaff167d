"""Loads oracle/liblinrad_oracle.so (CPU restatement, test infrastructure) behind the product's own StageAPI."""
import ctypes as C
import os
import subprocess

from linrad_amd.abi import StageAPI

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_lib = None


def oracle_lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liblinrad_oracle.so")
        src = os.path.join(ORACLE_DIR, "linrad_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
        _lib = C.CDLL(so)
    return _lib


def open_oracle(cfg):
    return StageAPI(oracle_lib(), "lro", cfg)
