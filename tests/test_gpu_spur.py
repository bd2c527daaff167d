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


def test_bench_shape_spur_tracking_equals_rounds_of_256_blocks(monkeypatch):
    """The call bench.py --spurs 8 times (4096 fft1 blocks = 1024 transforms of 65536 points per round, eight carriers acquired on the device
    after the first round, tracked and taken out inside lrh_make_fft2 from then on) against the same run in rounds of 256 blocks on the
    serial schedule: the loop is a recursion over the transforms and owes nothing to how many share a launch -- same loop state after
    the run, same power sums, waterfall and narrowband output, bit for bit."""
    import os
    from linrad_amd import abi
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    from linrad_amd.spurs import spur_spectra
    from linrad_amd.workload import chain_config, strong_liminfo
    N1, N2, nspur = 16384, 65536, 8
    s = synth_defaults(N1, 0)
    res = []
    for batch, pipeline in ((4096, None), (256, "0")):
        if pipeline is not None:
            monkeypatch.setenv("LRH_PIPELINE", pipeline)
        cfg = chain_config(14, 16, batch=4096, fft3_n=12, mix2_n=8, rounds=2)
        cfg.fft1_float_sparse, cfg.fft2_float_sparse = 1, 0      # acquisition reads whole transforms
        cfg.stupid_bln_mode = 0
        rx = open_hip(cfg)
        monkeypatch.delenv("LRH_PIPELINE", raising=False)
        rx.timf1_write(synth_iq(s, 0, cfg.timf1_bytes // 4))
        rx.set_liminfo(strong_liminfo(s, 14))
        rx.set_mix1_selfreq(0.31 * N2 + 0.3)
        rx.wideband_dsp(4096, batch)
        rx.sync()
        rx.spur_config(nspur, 16, spur_spectra(2))
        order = np.argsort(-np.asarray(s.carrier_amp[:s.ncarriers]))
        locked = sum(int(rx.spur_acquire(int(round(N2 / 2 + s.carrier_bin[i] * N2 / s.fft_size)) - 3)) for i in order[:nspur])
        rx.wideband_dsp(2 * 4096, batch)
        st = [(q.spur_location, q.spur_flag, q.spur_freq, q.spur_d0pha, q.spur_d1pha, q.spur_d2pha, q.spur_ampl, q.spur_noise, q.spur_avgd2) for q in rx.spur_get(16)]
        res.append(dict(locked=locked, st=st, p=rx.p.as_dict(), ps2=rx.export(abi.RING_FFT2_POWERSUM), wf=rx.export(abi.RING_WG_WATERF),
                        timf3=rx.export(abi.RING_TIMF3_FLOAT), baseb=rx.export(abi.RING_BASEB_RAW)))
        rx.close()
    a, b = res
    print("locked", a["locked"], b["locked"], "spurs", len(a["st"]))
    assert a["locked"] == b["locked"] >= 6 and len(a["st"]) == len(b["st"]) == a["locked"]
    assert a["p"] == b["p"]
    assert a["st"] == b["st"]
    for k in ("ps2", "wf", "timf3", "baseb"):
        assert np.array_equal(a[k], b[k]), (k, int(np.count_nonzero(a[k] != b[k])))
