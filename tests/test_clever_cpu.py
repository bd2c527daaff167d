"""The oracle's restatement of the linear ("clever") blanker against the compiled reference's goldens (CPU)."""
import pytest

import cleverlib
from oracle_binding import open_oracle
from refcases import CLEVER


@pytest.mark.parametrize("name", list(CLEVER))
def test_oracle_clever_blanker_matches_reference(name):
    g = cleverlib.load(name)
    out = cleverlib.run(open_oracle, name, g)
    rep = cleverlib.compare(out, g, 2e-6)
    print(name, rep)
    out["api"].close()
