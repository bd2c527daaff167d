"""CPU: the oracle's restatement of the spur tracking / subtraction (eliminate_spurs + refine_pll_parameters + spur_phase_parameters +
shift_spur_table; spur.c:36-494, 634-680, 1427-1652, spursub.c:942-1246) against the compiled reference, which acquires the spur
itself (store_new_spur / spur_phase_lock) and tracks it for the rest of the run (goldens tests/golden/spur_*.npz)."""
import pytest

import spurlib
from refcases import SPUR


@pytest.mark.parametrize("name", list(SPUR))
def test_oracle_spur_tracking_matches_reference(name):
    from oracle_binding import open_oracle
    g = spurlib.load(name)
    rep = spurlib.compare(spurlib.run(open_oracle, name, g), g, tol=2e-6)
    print(name, rep)


@pytest.mark.parametrize("name", list(SPUR))
def test_oracle_spur_acquisition_matches_reference(name):
    """store_new_spur + spur_phase_lock (spursub.c:619, 1247) restated: the oracle finds and locks the carrier on its own resident
    spectra at the transform where the reference did; loop state after the lock and the whole tracked run equal the reference's"""
    from oracle_binding import open_oracle
    g = spurlib.load(name)
    out = spurlib.run(open_oracle, name, g, acquire=True)
    print(name, spurlib.compare_acquisition(out, g), spurlib.compare(out, g, tol=2e-6))


def test_line_shape_table_matches_the_reference_table():
    """linrad_amd.spurs.spur_spectra restates init_spur_spectra (spursub.c:824-940); the golden carries the compiled reference's table"""
    import numpy as np
    from linrad_amd.spurs import spur_spectra
    g = spurlib.load("spur_n10_n12")
    assert np.max(np.abs(spur_spectra(2) - g["spur_spectra"])) < 2e-5      # the reference forms it with a float32 transform of 256 points
