"""CPU: the oracle's restatement of the selective limiter (fft1_update_liminfo + selfreq_liminfo, sellim.c:738-1157, 38-157)
against the compiled reference's liminfo after every update (goldens tests/golden/sellim_*.npz)."""
import pytest

import sellimlib
from refcases import SELLIM


@pytest.mark.parametrize("name", list(SELLIM))
def test_oracle_selective_limiter_matches_reference(name):
    from oracle_binding import open_oracle
    g = sellimlib.load(name)
    rep = sellimlib.compare(sellimlib.run(open_oracle, name, g), g, tol=2e-6)
    print(name, rep)
    assert rep["cleared_equal"]


def test_oracle_wideband_dsp_makes_the_limiter_calls_itself():
    """lro_wideband_limiter: both limiters inside lro_wideband_dsp after every round == the caller's calls after every round"""
    import numpy as np
    from oracle_binding import open_oracle
    name = "sellim2_n10_n12"
    g = sellimlib.load(name)
    a = sellimlib.run_dsp(open_oracle, name, g, in_call=True)
    b = sellimlib.run_dsp(open_oracle, name, g, in_call=False)
    assert np.count_nonzero(a["lim"]) > 50 and a["amp"] != 1.0            # the limiter has been at work (more than half the band strong here: factor 0)
    for k in ("lim", "timf2", "pwr", "timf3"):
        assert np.array_equal(a[k], b[k]), k
    assert a["amp"] == b["amp"] and a["p"] == b["p"]
