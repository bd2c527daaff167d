"""CPU, build container only: the Linrad-side glue (integration/hipshim.c + integration/linrad_hip.patch) EXECUTES.

oracle/build_shim_harness.sh links the head-less reference driver with the reference objects as the patch leaves them and with
hipshim.c compiled over the oracle's C ABI (oracle/shim_alias.h: lrh_* -> lro_*, the device replaced by liblinrad_oracle.so).  With
fft1 version 21 selected the driver then calls the reference's own entry points -- fft1_b, fft1_c, make_timf2, first_noise_blanker,
compute_timf2_powersum, make_fft2, fft2_mix1_fixed / _afc, and with the second fft off (Linrad's default, uivar.c:371-392)
fft1_c, fft1_mix1_fixed / _afc; fft1_update_liminfo for the limiter case -- every one of which hands over to the glue at the
hunk the patch added.  What is compared with the UNPATCHED reference's goldens is listed in tests/shimlib.py.  The same cases run
against the HIP library on the GPU box (tests/test_gpu_shim.py, oracle/_ref/shim_harness_hip)."""
import os
import subprocess

import pytest

import shimlib
from shimlib import ROOT

REF = "/root/reference"
HARNESS = os.path.join(ROOT, "oracle", "_ref", "shim_harness")
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


@pytest.fixture(scope="module")
def harness():
    # LRH_SHIM_SANITIZE=1: the same cases through the AddressSanitizer / UBSan build of the glue and the driver (SAN=1 oracle/build_shim_harness.sh;
    # run with ASAN_OPTIONS=detect_leaks=0 -- the reference objects are not instrumented and the harness does not free Linrad's arenas)
    san = os.environ.get("LRH_SHIM_SANITIZE") == "1"
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "shim"] + (["SAN=1"] if san else []), capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, **({"SAN": "1"} if san else {})))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return HARNESS + "_asan" if san else HARNESS


@pytest.mark.parametrize("name", shimlib.GOLDEN_CASES)
def test_patched_reference_through_the_glue_matches_the_unpatched_goldens(harness, tmp_path, name):
    print(name, shimlib.check_golden_case(harness, tmp_path, name, truth_factor=1.05))   # (here the float32 ORACLE sits behind the glue: 1.02 on timf3 of n10_n12_afc)


@pytest.mark.parametrize("name", ["n10_n12", "n10_mix1only"])
def test_stage_functions_called_from_linrads_stage_threads_in_lock_step(harness, tmp_path, name):
    """shim_threads=1: fft1_b on a worker thread, fft1_c / make_timf2 / first_noise_blanker on THREAD_TIMF2's stand-in, make_fft2 on
    second_fft's, mix1 on the narrowband thread's (oracle/ref_harness.c on_stage); same goldens"""
    print(name, shimlib.check_golden_case(harness, tmp_path, name, extra=["shim_threads=1"], truth_factor=1.05))


def test_network_output_hooks_fill_the_host_rings_the_senders_read(harness, tmp_path):
    """ui.network_flag = NET_RXOUT_FFT1 | TIMF2 | FFT2 is no longer refused: hip_net_fft1 / hip_net_timf2 / hip_net_fft2 (hunks wcw.c:1025, 1039,
    rxin.c:946, 1026) fetch the spans the senders read"""
    print(shimlib.check_golden_case(harness, tmp_path, "n10_n12", extra=["shim_net=1"]))


@pytest.mark.parametrize("extra", [(), ("shim_batch=4",)])
def test_selective_limiter_hooks_run_on_the_device_resident_sums(harness, tmp_path, extra):
    shimlib.check_sellim_case(harness, tmp_path, extra=extra)


def test_linear_blanker_tables_reach_the_device_through_the_glue(harness, tmp_path):
    shimlib.check_clever_case(harness, tmp_path)


def test_glue_refuses_what_version_21_does_not_serve(harness, tmp_path):
    shimlib.check_refusal(harness, tmp_path)


@pytest.mark.parametrize("name,workers", [("n10_n12", 3)])
def test_free_running_stage_threads_with_fft1b_workers(harness, tmp_path, name, workers):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    shimlib.check_free_running(harness, os.path.join(ROOT, "oracle", "_ref", "ref_harness"), tmp_path, name, workers)


@pytest.mark.parametrize("name", ["spur_n10_n12", "spur_n10_n12_drift", "spur_n10_fft1", "spur_n15_fft1"])
def test_spur_removal_is_served_through_the_acquisition_hooks(harness, tmp_path, name):
    print(name, shimlib.check_spur_case(harness, tmp_path, name))


@pytest.mark.parametrize("name", ["spur_n10_n12_clicks", "spur_n10_n12_clicks_strong"])
def test_linrads_own_spur_control_plane_over_the_glue(harness, tmp_path, name):
    """init_spur_elimination (spursub.c:181-343) as a whole: clicks, the search spectrum, acquisition, initial_remove_spur, the weaker of a
    close pair dropped (the last of the list by count only; another through remove_spur), swap_spurs"""
    print(name, shimlib.check_spur_clicks_case(harness, tmp_path, name))


@pytest.mark.parametrize("name", ["twochan_n10", "twochan_n9_sin3", "twochan_real_n9"])
def test_two_rf_channels_are_served_as_two_contexts(harness, tmp_path, name):
    print(name, shimlib.check_twochan_case(harness, tmp_path, name))
    if "real" not in name:                                    # (the chain golden exists for the I/Q cases)
        print(name, shimlib.check_twochan_chain(harness, tmp_path, name))
        # ... and from Linrad's stage threads (wideband, timf2, second fft, narrowband) in lock step: each hook copies its own pointers only
        print(name, shimlib.check_twochan_chain(harness, tmp_path, name, extra=["shim_threads=1"]))
