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
