"""linrad_amd -- MI355X-native implementation of Linrad's wideband DSP hot path
(fft1 -> timf2 + blank1 -> fft2 -> mix1) behind a C ABI (include/linrad_hip.h).

  linrad_amd.abi   ctypes mirror of the C ABI (structs, StageAPI wrapper named after the reference stage functions)
  linrad_amd.lib   loader for the in-tree liblinrad_hip.so (HIP kernels, gfx950); no CPU fallback
  linrad_amd/csrc  hand-written HIP kernels + C-ABI host code
"""
from .abi import (LrhConfig, LrhPtrs, LrhBlankerState, LrhMix1State, LrhSynth, StageAPI, default_config)  # noqa: F401

__all__ = ["LrhConfig", "LrhPtrs", "LrhBlankerState", "LrhMix1State", "LrhSynth", "StageAPI", "default_config"]
