"""GPU: spur tracking / subtraction on the device (k_spur inside lrh_make_fft2) against the compiled reference, which acquires the
spur itself and tracks it (goldens tests/golden/spur_*.npz); the acquisition result is handed over like the control plane would."""
import numpy as np
import pytest

import spurlib
from refcases import SPUR

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(SPUR))
def test_hip_spur_tracking_matches_reference(name):
    from linrad_amd.lib import open_hip
    g = spurlib.load(name)
    rep = spurlib.compare(spurlib.run(open_hip, name, g), g, tol=1e-5)
    print(name, rep)


def test_hip_spur_batched_calls_equal_single_calls():
    """several transforms per lrh_make_fft2 call: k_spur walks them in order inside one launch; same loop state, same rings"""
    from linrad_amd.lib import open_hip
    name = "spur_n10_n12_drift"
    g = spurlib.load(name)
    a = spurlib.run(open_hip, name, g, batch=1)
    b = spurlib.run(open_hip, name, g, batch=3)
    assert np.array_equal(a["fft2"], b["fft2"]) and np.array_equal(a["timf3"], b["timf3"]) and np.array_equal(a["ps2"], b["ps2"])
    last_a, last_b = a["api"].spur_get()[0], b["api"].spur_get()[0]
    assert (last_a.spur_location, last_a.spur_freq, last_a.spur_d0pha, last_a.spur_ampl) == (last_b.spur_location, last_b.spur_freq, last_b.spur_d0pha, last_b.spur_ampl)


def test_spur_api_errors():
    from linrad_amd import abi
    from linrad_amd.lib import open_hip
    from linrad_amd.abi import LrhError
    from linrad_amd.workload import chain_config
    rx = open_hip(chain_config(fft1_n=10, fft2_n=12, batch=4))
    with pytest.raises((LrhError, RuntimeError), match=f"rc={abi.LRH_EINVAL}"):
        rx.spur_config(4, rx.cfg.max_fft2n, np.zeros(2048, np.float32))      # 4 * speknum > max_fft2n
    rx.spur_config(0, 0, np.zeros(2048, np.float32))                        # off
    assert rx.spur_get() == []


@pytest.mark.parametrize("name", list(SPUR))
def test_hip_spur_acquisition_matches_reference(name):
    """lrh_spur_acquire: store_new_spur + spur_phase_lock on the device-resident fft2 spectra at the transform where the reference
    acquired; the loop state after the lock equals the reference's and the run it then tracks equals the golden like the handed-over one"""
    from linrad_amd.lib import open_hip
    g = spurlib.load(name)
    out = spurlib.run(open_hip, name, g, acquire=True)
    print(name, spurlib.compare_acquisition(out, g), spurlib.compare(out, g, tol=1e-5))


def test_hip_spur_acquisition_refuses_noise():
    """seven bins of plain noise: no lock, nothing tracked"""
    from linrad_amd.lib import open_hip
    name = "spur_n10_n12"
    g = spurlib.load(name)
    out = spurlib.run(open_hip, name, g, acquire=False)
    rx = out["api"]
    assert not rx.spur_acquire(700)
    assert len(rx.spur_get()) == 1
