"""The oracle's restatement of the linear blanker on two coupled channels (get_pulse_pol, transform_timf2_pol, subtract_twochan_pulse,
blank1.c:232-609) against the compiled two-channel reference's goldens (CPU)."""
import pytest

import clever2lib
from oracle_binding import open_oracle
from refcases import CLEVER2


@pytest.mark.parametrize("name", list(CLEVER2))
def test_oracle_two_channel_clever_blanker_matches_reference(name):
    g = clever2lib.load(name)
    res = clever2lib.run(open_oracle, name, g, frames_mode=False)
    rep = clever2lib.compare(res, g, 2e-6)
    print(name, rep)
    for rx in res["rxs"]:
        rx.close()
