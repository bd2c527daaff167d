"""GPU: the linear blanker on two coupled channels (k_clever's two-channel fit: get_pulse_pol, transform_timf2_pol,
subtract_twochan_pulse, blank1.c:232-609) against the compiled two-channel reference's goldens and against the oracle."""
import numpy as np
import pytest

import clever2lib
from refcases import CLEVER2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("frames_mode", [False, True])
@pytest.mark.parametrize("name", list(CLEVER2))
def test_hip_two_channel_clever_blanker_matches_reference(name, frames_mode):
    from linrad_amd.lib import open_hip
    g = clever2lib.load(name)
    res = clever2lib.run(open_hip, name, g, frames_mode=frames_mode)
    rep = clever2lib.compare(res, g, 1e-5)
    print(name, rep)
    for rx in res["rxs"]:
        assert rx.blanker_state().clever_serial_calls > 0            # two channels: the one-wave replay
        rx.close()


def test_hip_two_channel_clever_blanker_follows_the_oracle_call_by_call():
    from linrad_amd.lib import open_hip
    from oracle_binding import open_oracle
    name = "clever2_n10"
    g = clever2lib.load(name)
    h = clever2lib.run(open_hip, name, g, frames_mode=False)
    o = clever2lib.run(open_oracle, name, g, frames_mode=False)
    assert np.array_equal(h["rows"][:, :, [0, 1, 2, 3, 4, 8, 9, 10]], o["rows"][:, :, [0, 1, 2, 3, 4, 8, 9, 10]])     # pointers, cleared, fitted and rejected per call
    n1 = h["rxs"][0].N1
    keep = np.ones(h["out"][0]["timf2"].size, bool)             # sin^2 overlap: the pending half beyond timf2_pa
    keep[(h["out"][0]["p"]["timf2_pa"] + np.arange(4 * (n1 // 2))) % keep.size] = False
    for ch in (0, 1):
        a, b = h["out"][ch]["timf2"].astype(np.float64) * keep, o["out"][ch]["timf2"].astype(np.float64) * keep
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-5
