"""GPU: the linear ("clever") blanker on the device (k_clever_prep / k_clever ahead of the stupid blanker inside
lrh_first_noise_blanker) against the compiled reference, which built the pulse tables itself (goldens tests/golden/clever_*.npz)."""
import numpy as np
import pytest

import cleverlib
from refcases import CLEVER

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(CLEVER))
def test_hip_clever_blanker_matches_reference(name):
    from linrad_amd.lib import open_hip
    g = cleverlib.load(name)
    out = cleverlib.run(open_hip, name, g)
    rep = cleverlib.compare(out, g, 1e-5)
    print(name, rep)
    out["api"].close()


def test_clever_tables_errors_and_off():
    from linrad_amd import abi
    from linrad_amd.abi import LrhError
    from linrad_amd.lib import open_hip
    from refcases import clever_case, lrh_config
    name = "clever_n10_n12"
    g = cleverlib.load(name)
    d, cl, iq, lim, des = clever_case(name)
    rx = open_hip(lrh_config(d, iq))                          # blnfit_range / pulsewidth of the uncalibrated default
    with pytest.raises((LrhError, RuntimeError), match=f"rc={abi.LRH_EINVAL}"):
        cleverlib.install_tables(rx, g, d["noise_floor"])
    rx.set_blanker_tables()                                  # off: accepted
    rx.close()
