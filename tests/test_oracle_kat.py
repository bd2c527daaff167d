"""Known-answer tests the reference itself implies for FFT back ends:
   {1,2,3,4,0,0,0,0} forward-then-inverse and unit-impulse round trips (oclprogs.c:693-757, 770-800),
   complex exponential -> single bin (cuda.c:61-70)."""
import ctypes as C

import numpy as np

from oracle_binding import oracle_lib


def _fft(x, n, forward=True):
    lib = oracle_lib()
    buf = np.ascontiguousarray(np.stack([x.real, x.imag], axis=1).ravel(), np.float32)
    f = lib.lro_fft_forward if forward else lib.lro_fft_backward
    f.argtypes, f.restype = [C.c_int, C.POINTER(C.c_float)], None
    f(n, buf.ctypes.data_as(C.POINTER(C.c_float)))
    return buf[0::2] + 1j * buf[1::2]


def test_roundtrip_1234():
    x = np.array([1, 2, 3, 4, 0, 0, 0, 0], np.complex64)
    y = _fft(_fft(x, 3, True), 3, False) / 8
    assert np.max(np.abs(y - x)) < 1e-5          # tolerance of oclprogs.c:700-757


def test_impulse_roundtrip_2pow18():
    n = 18
    x = np.zeros(1 << n, np.complex64)
    x[5] = 1
    y = _fft(_fft(x, n, True), n, False) / (1 << n)
    assert np.max(np.abs(y - x)) < 1e-5


def test_exponential_single_bin():
    N = 256
    for i in (1, 3, 15):
        x = np.exp(2j * np.pi * i * np.arange(N) / N).astype(np.complex64)
        X = _fft(x, 8, True)
        assert abs(X[i] - N) < 1e-2 and np.max(np.abs(np.delete(X, i))) < 1e-2


def test_forward_matches_numpy():
    rng = np.random.default_rng(0)
    for n in (6, 10, 13):
        x = (rng.normal(size=1 << n) + 1j * rng.normal(size=1 << n)).astype(np.complex64)
        ref = np.fft.fft(x.astype(np.complex128))
        assert np.linalg.norm(_fft(x, n) - ref) / np.linalg.norm(ref) < 5e-7
        refb = np.fft.ifft(x.astype(np.complex128)) * (1 << n)
        assert np.linalg.norm(_fft(x, n, False) - refb) / np.linalg.norm(refb) < 5e-7


def test_timf2_network_payload_formula():
    """NET_RXOUT_TIMF2 payload of the oracle = rxin.c:949-956 on its own timf2 ring"""
    import numpy as np
    from linrad_amd import abi
    from oracle_binding import open_oracle
    from paritylib import load_golden, run_case
    g = load_golden("n8_n10")
    rx = run_case(open_oracle, "n8_n10", golden=g)["api"]
    net = rx.export_timf2_net(4 * 100, 1000, 2.0, 0.5).reshape(-1, 2)
    t = rx.export(abi.RING_TIMF2_FLOAT).reshape(-1, 4)[100:1100]
    assert np.array_equal(net, np.float32(2.0) * (t[:, :2] + np.float32(0.5) * t[:, 2:]))
