"""GPU: the Linrad-side glue against the real library.  oracle/_ref/shim_harness_hip (oracle/build_shim_harness.sh, built in the
container where the reference tree is, shipped like ref_harness) = the reference's objects as integration/linrad_hip.patch leaves
them + the head-less driver + integration/hipshim.c AS SHIPPED, linked to linrad_amd/liblinrad_hip.so: what a patched xlinrad64
executes on its wideband side -- fft1_b case 21 (fft1.c:3519-3553 hunk) -> fft1_c -> make_timf2 (timf2.c:31) -> first_noise_blanker
(blank1.c:684) -> compute_timf2_powersum -> make_fft2 (fft2.c:52) -> fft2_mix1_fixed / _afc (mix1.c:934, 863), the second-fft-off chain
fft1_c -> fft1_mix1_fixed / _afc (mix1.c:995, 1044), fft1_update_liminfo (sellim.c:738), the linear blanker's table hand-over, the
producer hook (rxin.c:1425) with lrh_host_register + lrh_timf1_write_async, the pinned read-backs of hipshim.c.  Host-visible
products against the goldens of the UNPATCHED compiled reference: pointer traces exact, rings <= 1e-5 (tests/shimlib.py).

Three call patterns: the single-CPU order (wcw.c:1094-1118); every stage on the thread Linrad calls it from with more than one CPU
(wcw.c:401-441 THREAD_TIMF2 beside wideband_dsp, second_fft, narrowband_dsp, an fft1_b worker) handed over in lock step; and the
same threads running free with three fft1_b workers (own handles / streams), blanker off so that the result does not depend on how
the calls happen to group, against the unpatched compiled reference run on the spot (oracle/_ref/ref_harness travels too)."""
import os
import subprocess

import numpy as np
import pytest

import shimlib
from paritylib import load_golden, relerr
from refcases import case_params, harness_args
from refdump import load_dump
from shimlib import ROOT

pytestmark = pytest.mark.gpu
HARNESS = os.path.join(ROOT, "oracle", "_ref", "shim_harness_hip")
REFH = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


@pytest.fixture(scope="module")
def harness():
    if os.path.isdir("/root/reference"):
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "shim"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    if not os.path.exists(HARNESS):
        pytest.skip("oracle/_ref/shim_harness_hip was not built (needs the reference tree: build() in the container)")
    r = subprocess.run(["nm", "-D", "--undefined-only", HARNESS], capture_output=True, text=True)
    assert " lrh_open" in r.stdout and " lro_" not in r.stdout, "the harness must bind the HIP library, not the oracle"
    return HARNESS


@pytest.mark.parametrize("name", shimlib.GOLDEN_CASES)
def test_patched_reference_over_the_hip_library_single_cpu_order(harness, tmp_path, name):
    print(name, shimlib.check_golden_case(harness, tmp_path, name))


@pytest.mark.parametrize("name", ["n10_n12", "n10_mix1only", "n10_n12_afc", "n9_n11_sin3"])
def test_patched_reference_over_the_hip_library_from_linrads_stage_threads(harness, tmp_path, name):
    print(name, shimlib.check_golden_case(harness, tmp_path, name, extra=["shim_threads=1"]))


@pytest.mark.parametrize("name", ["n10_n12", "n10_n12_fft3", "n9_n11_sin3"])
def test_glue_opens_the_fft2_ring_sparse_when_nobody_reads_it(harness, tmp_path, name):
    """hip_open with hip_sparse_rings = -1 (what a patched xlinrad64 runs with): AFC off, no spurs, no NET_RXOUT_FFT2 -> cfg.fft2_float_sparse,
    the transform kernels of the headline keep the power sums and waterfall lines themselves and store only the band fft2_mix1_fixed cuts
    out; everything Linrad sees on the host equals the unpatched reference's as before"""
    print(name, shimlib.check_golden_case(harness, tmp_path, name, extra=["shim_sparse=1"]))


def test_network_output_hooks_on_the_device(harness, tmp_path):
    """NET_RXOUT_FFT1 / TIMF2 / FFT2 (wcw.c:1024-1043, rxin.c:944-966, 1026-1035): the hooks in front of the senders' reads fetch the spans
    from the device rings; the FFT1 payload is the transform before fft1_c's correction (lrh_export_fft1_net)"""
    print(shimlib.check_golden_case(harness, tmp_path, "n10_n12", extra=["shim_net=1"]))


@pytest.mark.parametrize("extra", [(), ("shim_threads=1",), ("shim_batch=4",)])
def test_selective_limiter_hook_on_the_device(harness, tmp_path, extra):
    shimlib.check_sellim_case(harness, tmp_path, extra=extra)


@pytest.mark.parametrize("extra", [(), ("shim_threads=1",)])
def test_linear_blanker_tables_through_the_glue_on_the_device(harness, tmp_path, extra):
    shimlib.check_clever_case(harness, tmp_path, extra=extra)


@pytest.mark.parametrize("name", ["spur_n10_n12", "spur_n10_n12_drift", "spur_n10_fft1", "spur_n15_fft1"])
def test_spur_removal_through_the_acquisition_hooks_on_the_device(harness, tmp_path, name):
    """store_new_spur / spur_phase_lock (spursub.c:619, 1247; hooked) -> lrh_spur_acquire on the resident fft2 spectra; eliminate_spurs inside
    lrh_make_fft2; the loop state the glue brings back after every transform against the unpatched reference's"""
    print(name, shimlib.check_spur_case(harness, tmp_path, name))


@pytest.mark.parametrize("name", ["spur_n10_n12_clicks", "spur_n10_n12_clicks_strong"])
def test_linrads_own_spur_control_plane_over_the_glue_on_the_device(harness, tmp_path, name):
    """init_spur_elimination (spursub.c:181-343) as a whole over liblinrad_hip.so: the operator's clicks -> peak search in the search spectrum the
    device keeps -> lrh_spur_acquire (with initial_remove_spur) -> the weaker of a close pair dropped (the last of the list by counting
    no_of_spurs down only: hip_spur_resync; the first through remove_spur) -> swap_spurs; three spurs' loop state after every transform"""
    print(name, shimlib.check_spur_clicks_case(harness, tmp_path, name))


@pytest.mark.parametrize("name", ["twochan_n10", "twochan_n9_sin3", "twochan_real_n9"])
def test_two_rf_channels_as_two_contexts_on_one_gpu(harness, tmp_path, name):
    """ui.rx_rf_channels = 2 (fft1.c:3686-3900, 3874-4080; blank1.c:1236-1300; fft2.c:1622-1815): one context per channel behind the same
    hooks, the cross-channel sums and gathers made by the glue, the correlation spectrum; against the goldens of the compiled two-channel
    reference.  Two REAL channels (fft1_reherm_dit_two, fft1_re.c:133-231): the producer hook de-interleaves Linrad's {a_k, b_k} frames."""
    print(name, shimlib.check_twochan_case(harness, tmp_path, name))
    if "real" not in name:                                    # (the chain golden exists for the I/Q cases)
        print(name, shimlib.check_twochan_chain(harness, tmp_path, name))
        # ... and from Linrad's stage threads (wideband, timf2, second fft, narrowband) in lock step: each hook copies its own pointers only
        print(name, shimlib.check_twochan_chain(harness, tmp_path, name, extra=["shim_threads=1"]))


def test_glue_refuses_what_version_21_does_not_serve_on_the_device(harness, tmp_path):
    shimlib.check_refusal(harness, tmp_path)


@pytest.mark.parametrize("name,workers", [("n10_n12", 3), ("n9_n11_sin3", 6)])
def test_free_running_stage_threads_with_fft1b_workers(harness, tmp_path, name, workers):
    if not os.path.exists(REFH):
        pytest.skip("oracle/_ref/ref_harness not shipped")
    shimlib.check_free_running(harness, REFH, tmp_path, name, workers)
