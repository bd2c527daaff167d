"""CPU: linrad_amd/csrc/lrh_phase.h -- do_mix1's running phase (`t1 += t2` per output sample in float, mix1.c:141-195) advanced by many samples in
closed form -- against the plain float loop, bit for bit: 400 000 random cases (phases inside and far outside +-pi, increments down to 1e-6 ulp-fractions,
starts at zero, runs through zero, exact ties, runs of up to 100 000 steps).  The library's host code builds mix1's phase tables with it."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_phase_advance_equals_the_float_loop(tmp_path):
    exe = str(tmp_path / "phase_advance_test")
    subprocess.check_call(["g++", "-O2", "-I", os.path.join(ROOT, "linrad_amd", "csrc"), os.path.join(ROOT, "tests", "csrc", "phase_advance_test.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout[-2000:]
