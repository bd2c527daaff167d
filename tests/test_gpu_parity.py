"""GPU: the HIP path (through the C ABI of liblinrad_hip.so) against the reference's golden vectors and the oracle."""
import numpy as np
import pytest

from paritylib import RINGS, compare_with_golden, load_golden, relerr, run_case, truth_gate, truth_of
from refcases import CASES

pytestmark = pytest.mark.gpu


def _open_hip(cfg):
    from linrad_amd.lib import open_hip
    return open_hip(cfg)


def _open_oracle(cfg):
    from oracle_binding import open_oracle
    return open_oracle(cfg)


@pytest.mark.parametrize("name", list(CASES))
def test_hip_matches_reference_golden(name):
    """Same call pattern as the compiled reference, block by block; tolerance 1e-5 relative RMS (north_star)."""
    g = load_golden(name)
    out = run_case(_open_hip, name, golden=g)
    floor_same = np.array_equal(out["itrace"][:, 4], g["itrace"].reshape(-1, 16)[:, 12])
    rep = compare_with_golden(out, g, tol=1e-5, check_blanker_exact=floor_same, floor_slack=0 if floor_same else 1,
                              mask_pending_timf2=out["api"].fft1_interleave_points == out["api"].N1 // 2, truth=lambda: truth_of(name, g))
    print(name, rep)


@pytest.mark.parametrize("name", ["n8_n10", "n9_n11_sin3", "n11_n9_nowin2", "n13_n15_big2"])
def test_hip_timf2_without_blanker(name):
    g = load_golden(name)
    out = run_case(_open_hip, name, golden=g, stupid=0)
    st = int(g["__stride"]) if "__stride" in g else 1
    a, b = out["timf2_float"].copy(), g["timf2_float_noblank"].copy()
    api = out["api"]
    keep = np.ones(a.size, bool)
    if api.fft1_interleave_points == api.N1 // 2:      # pending raw second half beyond timf2_pa: see paritylib
        keep[(api.p.timf2_pa + np.arange(4 * (api.N1 // 2))) % a.size] = False
    assert relerr((a * keep)[::st], b * keep[::st]) < 1e-5
    assert relerr(out["timf2_pwr_float"][::st], g["timf2_pwr_float_noblank"]) < 1e-5


@pytest.mark.parametrize("name,batch", [("n8_n10", 4), ("n10_n12", 3), ("n11_n9_nowin2", 4)])
def test_hip_batched_matches_oracle(name, batch):
    """Batched launches (several fft1 blocks per call, blanker once per batch) against the oracle run the same way."""
    g = load_golden(name)
    a = run_case(_open_hip, name, golden=g, batch=batch)
    b = run_case(_open_oracle, name, golden=g, batch=batch)
    assert np.array_equal(a["itrace"][:, [0, 1, 2, 3, 6, 7, 8, 9, 10]], b["itrace"][:, [0, 1, 2, 3, 6, 7, 8, 9, 10]])
    assert np.abs(a["itrace"][:, 4] - b["itrace"][:, 4]).max() <= 1
    for _, key in RINGS:
        if key == "timf3_float":
            continue
        x, y = a[key], b[key]
        if key == "timf2_float" and a["api"].fft1_interleave_points == a["api"].N1 // 2:
            idx = (a["api"].p.timf2_pa + np.arange(4 * (a["api"].N1 // 2))) % x.size
            x, y = x.copy(), y.copy()
            x[idx] = 0
            y[idx] = 0
        assert relerr(x, y) < 1e-5, key
    if np.array_equal(a["itrace"][:, 4], b["itrace"][:, 4]):
        assert np.array_equal(a["timf2_pwr_float"] == 0, b["timf2_pwr_float"] == 0)
    assert np.abs(a["wf_lines"].astype(int) - b["wf_lines"].astype(int)).max() <= 2


def test_hip_long_call_matches_oracle():
    """One call spanning many averaging periods (32 fft1 blocks, avg1num 3: ten slow-average updates per call, more than
    a full cycle of the rolling refresh): k_slowsum then starts its search one refresh cycle before the end."""
    g = load_golden("n8_n10")
    kw = dict(max_batch=32, max_fft1n=64, fft1_sumsq_bufsize=32 * 256)
    a = run_case(_open_hip, "n8_n10", golden=g, batch=32, **kw)
    b = run_case(_open_oracle, "n8_n10", golden=g, batch=32, **kw)
    assert np.array_equal(a["itrace"][:, [0, 1, 2, 3, 6, 7, 8, 9, 10]], b["itrace"][:, [0, 1, 2, 3, 6, 7, 8, 9, 10]])
    for key in ("fft1_sumsq", "fft1_slowsum", "fft1_float", "fft2_float"):
        assert relerr(a[key], b[key]) < 1e-5, key


def test_hip_fft3_mix2_matches_oracle():
    """fft3 ring against the reference golden (in test_hip_matches_reference_golden) and the mix2 filter/decimate output
    baseb_raw against the oracle restatement (reference parity of the same part: golden n10_n12_fft3 in test_gpu_parity / test_oracle_golden)."""
    g = load_golden("n10_n12_fft3")
    a = run_case(_open_hip, "n10_n12_fft3", golden=g)
    b = run_case(_open_oracle, "n10_n12_fft3", golden=g)
    assert (a["api"].p.fft3_pa, a["api"].p.fft3_px, a["api"].p.baseb_pa, a["api"].p.timf3_px) == \
           (b["api"].p.fft3_pa, b["api"].p.fft3_px, b["api"].p.baseb_pa, b["api"].p.timf3_px)
    assert np.count_nonzero(b["baseb_raw"]) > 500
    t, rep = truth_of("n10_n12_fft3", g), {}
    for k in ("fft3", "baseb_raw"):          # 1e-5, or no further from the float64 truth than the oracle's own float32 result
        truth_gate(rep, k, a[k], b[k], t[k], factor=1.05)    # (the partner is the oracle: tests/test_gpu_fullsize.py ORACLE_FACTOR)
    print(rep)


@pytest.mark.parametrize("name", ["n8_n10", "n10_n12", "n9_n11_dir", "n9_n11_iqcal", "n10_n12_dword", "n9_n11_real", "n15_n17_big1"])
def test_hip_fft1_net_payload_matches_reference(name):
    """NET_RXOUT_FFT1 sends fft1_float as fft1_b leaves it (wcw.c:1024-1043), before fft1_c's filter correction: the golden keeps
    that block of the compiled reference (fft1_first_raw); the HIP path recomputes it on request (lrh_export_fft1_net)"""
    g = load_golden(name)
    out = run_case(_open_hip, name, golden=g)
    api = out["api"]
    got = api.export_fft1_net(0, 1)
    assert relerr(got, g["fft1_first_raw"]) < 1e-5
    two = api.export_fft1_net(0, 2)
    assert np.array_equal(two[:got.size], got)


def test_hip_tables_are_bit_equal_to_the_compiled_references():
    """host tables (make_window fft0.c:812-921, make_filcorrstart fft1.c:4653-4724, make_wg_yfac wide_graph.c:955-1001, mix1 / fft3 windows):
    every float identical to what the compiled reference built (round 5: the edge taper took sin() of a float, which is sinf in C++)"""
    for name in CASES:
        g = load_golden(name)
        out = run_case(_open_hip, name, golden=g)
        api = out["api"]
        for t in ("fft1_window", "fft2_window", "mix1_fqwin", "fft1_filtercorr", "wg_waterf_yfac", "fft1_inverted_window", "fft3_window"):
            if t not in g or (t == "fft2_window" and out["cfg"].fft2_sinpow == 0):
                continue                      # unused without a window; the reference leaves it unallocated/zero
            if t == "fft3_window" and not out["cfg"].fft3_n:
                continue
            got = api.get_table(t, g[t].size)
            ref = g[t][:got.size]
            bad = got != ref
            assert not bad.any(), (name, t, np.nonzero(bad)[0][:5], got[bad][:5], ref[bad][:5])


def test_c_host_driver_runs():
    """examples/lrh_stream.c: the plain-C producer + wideband loop over the C ABI (built by __graft_entry__.build())."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "lrh_stream")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "linrad_amd", "csrc"), "example"])   # gcc only; no-op when fresh
    out = subprocess.run([exe, "12", "14", "0.5"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Msamples/s" in out.stdout and "blanker: noise floor" in out.stdout


def test_async_producer_write_is_ordered_before_fft1():
    """lrh_timf1_write_async + page-locked host memory: the copy runs on its own stream, later stage calls wait for it on the
    device, and a second copy into the span an enqueued fft1 launch reads goes behind that launch."""
    from linrad_amd import abi
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    from linrad_amd.workload import chain_config
    cfg = chain_config(fft1_n=12, fft2_n=10, batch=16)
    s = synth_defaults(1 << cfg.fft1_n, 0)
    n = cfg.timf1_bytes // 4
    iq_a, iq_b = synth_iq(s, 0, n), synth_iq(s, 12345, n)
    ref = []
    for iq in (iq_a, iq_b):
        rx = open_hip(cfg)
        rx.timf1_write(iq)
        rx.fft1_b(16)
        ref.append(rx.export(abi.RING_FFT1_FLOAT))
    rx = open_hip(cfg)
    host_a, host_b = np.ascontiguousarray(iq_a), np.ascontiguousarray(iq_b)
    rx.host_register(host_a)
    rx.timf1_write_async(host_a)
    rx.fft1_b(16)                                   # waits for the copy on the device
    p_after = rx.ptrs_copy()
    rx.timf1_write_async(host_b)                    # overwrites what the launch above reads: must queue behind it
    out_a = rx.export(abi.RING_FFT1_FLOAT)
    rx.p = type(rx.p)()                             # start over on the new ring content
    rx._f("ptrs_init")(rx.ctx, __import__("ctypes").byref(rx.p))
    rx.fft1_b(16)
    out_b = rx.export(abi.RING_FFT1_FLOAT)
    rx.timf1_write_wait()
    rx.host_unregister(host_a)
    assert p_after.fft1_pa != 0
    assert np.array_equal(out_a, ref[0]) and np.array_equal(out_b, ref[1]) and not np.array_equal(ref[0], ref[1])


def test_timf2_network_payload():
    """NET_RXOUT_TIMF2 payload (what MAP65 / slave Linrads receive, rxin.c:944-966): gain * (weak + strong_scale * strong) per
    sample, formed on the device from the planar rings; against the oracle's restatement and the formula on the exported ring,
    across the ring wrap."""
    from linrad_amd import abi
    g = load_golden("n8_n10")
    res = []
    for fn in (_open_hip, _open_oracle):
        out = run_case(fn, "n8_n10", golden=g)
        rx = out["api"]
        size = 4 * rx.cfg.timf2pow_size
        pt = (size - 4 * 700) % size                               # 700 samples before the wrap, 1500 in all
        net = rx.export_timf2_net(pt, 1500, 0.5, 0.25).reshape(-1, 2)
        t = rx.export(abi.RING_TIMF2_FLOAT).reshape(-1, 4)
        idx = (pt // 4 + np.arange(1500)) % rx.cfg.timf2pow_size
        want = np.float32(0.5) * (t[idx, :2] + np.float32(0.25) * t[idx, 2:])
        assert np.abs(want).max() > 10 and np.abs(net - want).max() <= 1e-6 * np.abs(want).max()
        res.append(net)
    assert relerr(res[0], res[1]) < 1e-5


def test_error_behaviour_and_ragged_calls():
    """Return codes of calls out of order or out of range (LRH_EINVAL / LRH_ESTATE, the caller maps them to lirerr), and a
    ragged call pattern (block count not a multiple of the batch, single blocks in between) against the oracle."""
    from linrad_amd import abi
    from linrad_amd.abi import LrhError
    from linrad_amd.lib import open_hip, synth_defaults, synth_iq
    from linrad_amd.workload import chain_config, strong_liminfo
    from oracle_binding import open_oracle
    cfg = chain_config(fft1_n=11, fft2_n=10, batch=8)
    s = synth_defaults(1 << cfg.fft1_n, 0)
    iq = synth_iq(s, 0, cfg.timf1_bytes // 4)
    outs = []
    from oracle_binding import open_truth
    for fn in (open_hip, open_oracle, open_truth):
        rx = fn(cfg)
        rx.timf1_write(iq)
        rx.set_liminfo(strong_liminfo(s, cfg.fft1_n))
        rx.set_mix1_selfreq(576.3)                                 # on the 100-LSB carrier of the synthetic signal (fs/16)
        for nb in (37, 1, 8, 5):                                   # 51 blocks in ragged calls
            rx.wideband_dsp(nb, 8)
        outs.append((rx.export(abi.RING_FFT1_SUMSQ), rx.export(abi.RING_TIMF2_PWR), rx.export(abi.RING_FFT2_FLOAT), rx.export(abi.RING_TIMF3_FLOAT), rx.p.as_dict()))
    ints = [k for k, v in outs[0][4].items() if isinstance(v, int)]
    assert {k: outs[0][4][k] for k in ints} == {k: outs[1][4][k] for k in ints}
    rep = {}
    for i, (a, b, t) in enumerate(zip(outs[0][:4], outs[1][:4], outs[2][:4])):
        # north-star tolerance; above it (the despiked power ring keeps float32 residue of the pulses it was cleaned of) both sides against the truth
        if i == 1:
            assert np.array_equal(a == 0, b == 0)
            same = (t == 0) == (b == 0)                           # (where the float64 build clears like the oracle)
            a, b, t = a[same], b[same], t[same]
        truth_gate(rep, str(i), a, b, t, factor=1.05)
    print(rep)
    rx = open_hip(cfg)
    for call, code in ((lambda: rx.fft1_b(9), abi.LRH_EINVAL),                       # batch > max_batch
                       (lambda: rx.make_fft2(cfg.max_fft2n + 1), abi.LRH_EINVAL),
                       (lambda: rx.fft2_xy_begin(rx.ptrs_copy(), 1), abi.LRH_ESTATE),  # not one of two coupled channels
                       (lambda: rx.blanker_begin(), abi.LRH_ESTATE),
                       (lambda: rx.make_fft3_all(1), abi.LRH_ESTATE),                  # fft3 not configured
                       (lambda: rx.set_combine_weights(1.0), abi.LRH_ESTATE),
                       (lambda: rx.export(abi.RING_FFT2_XYSUM), abi.LRH_ESTATE),
                       (lambda: rx.export(abi.RING_FFT1_SLOWSUM, 0, (1 << cfg.fft1_n) + 1), abi.LRH_EINVAL)):
        with pytest.raises((LrhError, RuntimeError), match=f"rc={code}"):
            call()
    assert rx.export(abi.RING_FFT1_SLOWSUM, 0, 0).size == 0
    creal = chain_config(fft1_n=11, fft2_n=10, batch=8)
    creal.timf1_real_input = 1
    rr = open_hip(creal)
    with pytest.raises((LrhError, RuntimeError), match=f"rc={abi.LRH_ESTATE}"):
        rr.set_foldcorr(np.zeros(2 << creal.fft1_n, np.float32))    # no I/Q image with real samples
