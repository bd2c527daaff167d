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
