"""CPU, build container only: every committed golden is reproducible bit for bit from the committed recipe -- the generator scripts
under tests/golden/ run the compiled reference (oracle/_ref/ref_harness, built by oracle/Makefile from the reference's own sources)
on the seeded inputs of tests/refcases.py, and what they write must equal the committed .npz array by array, bookkeeping included."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not present (GPU box)")

SCRIPTS = ["make_golden.py", "make_golden_2ch.py", "make_golden_clever.py", "make_golden_clever2.py", "make_golden_sellim.py",
           "make_golden_spur.py", "make_rawdat_golden.py", "make_filehdr_golden.py"]


@pytest.mark.parametrize("script", SCRIPTS)
def test_generator_reproduces_the_committed_goldens(script, tmp_path):
    env = dict(os.environ, LRH_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(GOLD, script)], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    made = sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    assert made, "the generator wrote nothing"
    for f in made:
        new, old = np.load(tmp_path / f), np.load(os.path.join(GOLD, f))
        assert sorted(new.files) == sorted(old.files), (f, sorted(set(new.files) ^ set(old.files)))
        for k in new.files:
            assert new[k].dtype == old[k].dtype and np.array_equal(new[k], old[k]), (f, k)


def test_every_committed_golden_has_a_generator(tmp_path):
    """no orphan fixtures: each .npz under tests/golden/ is written by one of the scripts above (names from tests/refcases.py)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from refcases import CASES, CLEVER, CLEVER2, SELLIM, SPUR, SPUR_CLICKS, TWOCHAN
    names = set(CASES) | set(CLEVER) | set(CLEVER2) | set(SELLIM) | set(SPUR) | set(SPUR_CLICKS) | set(TWOCHAN) | {"rawdat_18bit", "filehdr"}
    names |= {n + "_chain" for n, t in TWOCHAN.items() if "chain" in t}
    have = {f[:-4] for f in os.listdir(GOLD) if f.endswith(".npz")}
    assert have == names, sorted(have ^ names)
