"""CPU (build container only): the Linrad-side change set of SURVEY 8f-1.  integration/linrad_hip.patch must apply to the
reference snapshot, every touched object and integration/hipshim.c must compile with the reference's own flags, every hook the
patch calls must be defined by hipshim.c and every library call hipshim.c makes must be exported by liblinrad_hip.so."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


def test_patch_applies_and_touched_objects_compile():
    r = subprocess.run(["bash", os.path.join(ROOT, "integration", "check_patch.sh"), REF], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "patch ok" in r.stdout
    for f in ("fft1var.c", "buf.c", "wcw.c", "fft1.c", "timf2.c", "blank1.c", "fft2.c", "mix1.c", "rxin.c", "hipshim.c"):
        assert f"compiled {f}" in r.stdout


def test_committed_patch_is_what_the_generator_writes(tmp_path):
    """the patch is generated (integration/make_patch.py, anchored insertions); the committed file must be current"""
    committed = open(os.path.join(ROOT, "integration", "linrad_hip.patch"), encoding="latin-1").read()
    env = dict(os.environ)
    work = tmp_path / "integration"
    work.mkdir()
    src = open(os.path.join(ROOT, "integration", "make_patch.py")).read()
    (work / "make_patch.py").write_text(src)
    subprocess.check_call([sys.executable, str(work / "make_patch.py"), REF], env=env)
    assert (work / "linrad_hip.patch").read_text(encoding="latin-1") == committed
    # zero-context diff: apart from the five one-line replacements (version count, four fft1_version rows) nothing of the reference's text is stored
    removed = [l for l in committed.splitlines() if l.startswith("-") and not l.startswith("---")]
    assert len(removed) == 5
