"""CPU: the oracle (oracle/linrad_oracle.c) against the committed golden vectors made by the compiled reference."""
import numpy as np
import pytest

from oracle_binding import open_oracle
from paritylib import compare_with_golden, load_golden, relerr, run_case
from refcases import CASES


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_golden(name):
    g = load_golden(name)
    out = run_case(open_oracle, name, golden=g)
    # fft2_size 131072: seventeen float32 butterfly stages on either side, in different orders (reference radix-2 DIF, oracle its own
    # decomposition); the narrow band cut out next to a carrier 46 dB up carries that rounding noise (measured 2.8e-6)
    rep = compare_with_golden(out, g, tol=5e-6 if name == "n15_n17_big1" else 2e-6)
    # the oracle shares the reference's fft1 arithmetic: the spectrum ring is bit-exact (up to the last bit of the
    # gain constant at N1 = 8192, where gcc -ffast-math folds pow() differently in the two translation units)
    a, b = out["_cmp"]["fft1_float"]
    # real input: the reference runs a split-radix real-to-Hermitian transform (fft0.c:33), the oracle the plain complex one
    assert np.array_equal(a, b) or (name == "n13_n15_big2" and relerr(a, b) < 2e-7) or ("_real" in name and relerr(a, b) < 3e-7)
    print(name, rep)


@pytest.mark.parametrize("name", ["n8_n10", "n9_n11_sin3"])
def test_oracle_timf2_without_blanker(name):
    g = load_golden(name)
    out = run_case(open_oracle, name, golden=g, stupid=0)
    st = int(g["__stride"]) if "__stride" in g else 1
    assert relerr(out["timf2_float"][::st], g["timf2_float_noblank"]) < 2e-6
    assert relerr(out["timf2_pwr_float"][::st], g["timf2_pwr_float_noblank"]) < 2e-6


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_tables_match_reference(name):
    g = load_golden(name)
    out = run_case(open_oracle, name, golden=g)
    api = out["api"]
    for t in ("fft1_window", "fft2_window", "mix1_fqwin", "fft1_filtercorr", "wg_waterf_yfac"):
        got = api.get_table(t, g[t].size)
        assert np.allclose(got, g[t][:got.size], rtol=1.5e-7, atol=0), t


def test_oracle_first_transform_identity():
    """SURVEY 8a identity: fft1_float[k] = conj(FFT(x*w))[(k - N/2) mod N] (checked against float64 numpy)."""
    g = load_golden("n8_n10")
    N1, I1 = 256, 128
    iq = g["iq"].astype(np.float64)
    x = iq[0::2] + 1j * iq[1::2]
    ring = 1
    while ring < x.size:
        ring <<= 1
    xr = np.zeros(ring, complex)
    xr[:x.size] = x
    w1 = g["fft1_window"]
    w = np.concatenate([w1[0::2], w1[1::2]]).astype(np.float64)
    seg = xr[(np.arange(N1) - I1) % ring]
    ref = np.conj(np.fft.fftshift(np.fft.fft(seg * w)))
    raw = g["fft1_first_raw"]
    assert relerr(raw[0::2] + 1j * raw[1::2], ref) < 5e-7
