"""CPU: the oracle's correlation spectrum (fft1_corrsum, fft1_slowcorr, fft1_slowcorr_tot; fft1.c:4146-4150, 4189-4193, 4584-4603) against
the compiled two-channel reference run with genparm[FFT1_CORRELATION_SPECTRUM] = 1."""
import pytest

import corrlib


@pytest.mark.parametrize("name,batch", [("twochan_n10", 1), ("twochan_n9_sin3", 1), ("twochan_real_n9", 1), ("twochan_n10", 3)])
def test_oracle_correlation_spectrum_matches_reference(name, batch):
    from oracle_binding import open_oracle
    d, g, out = corrlib.run(open_oracle, name, batch)
    print(name, corrlib.compare(d, g, out, 2e-6))


@pytest.mark.parametrize("name,batch", [("twochan_n10", 1), ("twochan_n10", 4)])
def test_oracle_correlation_spectrum_through_wideband_dsp(name, batch):
    """lro_wideband_dsp with two coupled contexts in one process (one thread each, linrad_amd.multichan.install_pair_exchange): the
    all-gather of LRH_X_SPEC is asked for by the library between fft1_c and make_timf2."""
    from oracle_binding import open_oracle
    d, g, out = corrlib.run_dsp(open_oracle, name, batch)
    print(name, corrlib.compare(d, g, out, 2e-6))
