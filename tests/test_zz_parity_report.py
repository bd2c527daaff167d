"""GPU, LAST in collection order and in a child process: the published parity report (scripts/parity_report.py -> gpurun_out/parity_r06.json).
It re-measures what tests/test_gpu_parity.py / _fullsize.py / the feature tests already assert, so it must never be able to hide them:
a crash of the child fails this one test and nothing else."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_report_runs_to_the_end_in_a_child_process():
    env = dict(os.environ, LRH_CRASH_TRACE="1", PYTHONFAULTHANDLER="1")
    env.pop("LRH_CRASH_TRACE_FD", None)                     # (a descriptor number of THIS process)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "parity_report.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stdout[-4000:]
    rep = json.load(open(os.path.join(ROOT, "gpurun_out", "parity_r06.json")))
    assert not rep["errors"] and rep["summary"]["cases"] >= 19 and rep["summary"]["max_rel_err_of_the_rings_within_tolerance"] <= 1e-5
