"""CPU: liblinrad_hip.so loads, exports every symbol include/linrad_hip.h declares, and refuses to run without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from linrad_amd import abi
from linrad_amd.lib import LIB_PATH, hip_lib, synth_defaults, synth_iq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "linrad_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lrh_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB_PATH), "build the HIP extension first (__graft_entry__.build())"
    lib = hip_lib()
    names = _declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.lrh_abi_version() == 7


def test_graft_entry_build_accepts_the_library_it_built():
    # the driver's build check: an incremental make, then the version the header declares against the one the library reports
    import __graft_entry__ as g
    g.build()


def test_struct_sizes_match_header():
    lib = hip_lib()
    cfg = abi.LrhConfig()
    lib.lrh_config_defaults.argtypes = [C.POINTER(abi.LrhConfig), C.c_int, C.c_int]
    assert lib.lrh_config_defaults(C.byref(cfg), 14, 16) == 0
    assert cfg.struct_size == C.sizeof(abi.LrhConfig)          # same layout on both sides of the ABI
    assert (cfg.fft1_n, cfg.fft2_n, cfg.fft1_sinpow, cfg.bckfft_att_n, cfg.mix1_bandwidth_reduction_n) == (14, 16, 2, 6, 6)


def test_every_public_struct_has_the_same_size_on_both_sides():
    """the ctypes restatements in linrad_amd/abi.py against sizeof in the compiled library (and in the oracle, built from the same header)"""
    from oracle_binding import oracle_lib
    order = [abi.LrhConfig, abi.LrhPtrs, abi.LrhBlankerState, abi.LrhBlankerTables, abi.LrhMix1State, abi.LrhSellim, abi.LrhSpur, abi.LrhAfc, abi.LrhSynth]
    for lib, fn in ((hip_lib(), "lrh_sizeof"), (oracle_lib(), "lro_sizeof")):
        f = getattr(lib, fn)
        f.argtypes, f.restype = [C.c_int], C.c_size_t
        assert [f(i) for i in range(len(order))] == [C.sizeof(t) for t in order], fn
        assert f(len(order)) == 0


def test_open_rejects_bad_config_and_missing_gpu():
    import torch
    lib = hip_lib()
    lib.lrh_open.argtypes = [C.POINTER(abi.LrhConfig), C.POINTER(C.c_void_p)]
    ctx = C.c_void_p()
    bad = abi.default_config(10, 12)
    bad.struct_size = 4
    assert lib.lrh_open(C.byref(bad), C.byref(ctx)) == abi.LRH_EINVAL
    bad = abi.default_config(10, 12, timf2pow_size=1000)       # not a power of two
    assert lib.lrh_open(C.byref(bad), C.byref(ctx)) == abi.LRH_EINVAL
    for kw in (dict(fft1_n=15), dict(timf1_real_input=1, sample_shift=1), dict(timf1_real_input=1, timf1_frame_channels=4, timf1_channel_index=1),
               dict(timf1_frame_channels=2, timf1_channel_index=2), dict(max_batch=0), dict(fft3_n=5, mix2_n=3)):
        bad = abi.default_config(kw.pop("fft1_n", 10), 12, **kw)   # sizes and modes the header says are not built / not valid
        assert lib.lrh_open(C.byref(bad), C.byref(ctx)) == abi.LRH_EINVAL, kw
    if not torch.cuda.is_available():
        ok = abi.default_config(10, 12, max_batch=4)
        rc = lib.lrh_open(C.byref(ok), C.byref(ctx))
        assert rc == abi.LRH_EDEVICE and not ctx.value          # no CPU fallback: the product path needs the GPU


def test_synth_is_position_addressable_and_seeded():
    s = synth_defaults(16384, 0)
    a = synth_iq(s, 0, 20000)
    b = synth_iq(s, 7777, 5000)
    assert np.array_equal(a[2 * 7777:2 * 12777], b)
    s1 = synth_defaults(16384, 1)
    assert not np.array_equal(synth_iq(s1, 0, 1000), a[:2000])
    x = a[0::2].astype(np.float64) + 1j * a[1::2]
    spec = np.abs(np.fft.fft(x[:16384] * np.hanning(16384)))
    k = int(np.argmax(spec))
    assert k in (16384 - 6000, 16384 - 6001, 16384 - 5999)      # strongest carrier of the SURVEY 8(d) signal: bin -6000
    assert np.abs(a).max() == 32767                             # impulses clip


def test_a_cpp_exception_inside_the_library_comes_back_as_an_error_code():
    """every extern "C" entry point is a function-try-block (round 5's GPU suite showed std::terminate reachable from lrh_make_timf2): the
    self-test raises inside the library and the C caller sees LRH_EINTERNAL"""
    import ctypes as C
    from linrad_amd.abi import LRH_EINTERNAL
    from linrad_amd.lib import hip_lib
    lib = hip_lib()
    lib.lrh_selftest_exception.argtypes, lib.lrh_selftest_exception.restype = [C.c_int], C.c_int
    for kind in (0, 1, 2):
        assert lib.lrh_selftest_exception(kind) == LRH_EINTERNAL == -6
    assert lib.lrh_selftest_exception(3) == 0
