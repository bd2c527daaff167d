"""CPU, build container only: the Linrad-side glue (integration/hipshim.c + integration/linrad_hip.patch) EXECUTES.

oracle/build_shim_harness.sh links the head-less reference driver with the reference objects as the patch leaves them and with
hipshim.c compiled over the oracle's C ABI (oracle/shim_alias.h: lrh_* -> lro_*, the device replaced by liblinrad_oracle.so).  With
fft1 version 21 selected the driver then calls the reference's own entry points -- fft1_b, fft1_c, make_timf2, first_noise_blanker,
compute_timf2_powersum, make_fft2, fft2_mix1_fixed / _afc, and with the second fft off (Linrad's default, uivar.c:371-392)
fft1_c, fft1_mix1_fixed / _afc; fft1_update_liminfo for the limiter case -- every one of which hands over to the glue at the
hunk the patch added.  What is compared with the UNPATCHED reference's goldens is what Linrad sees on the host afterwards: the
pointer globals after every block, the blanker scalars, fft1_sumsq / fft1_slowsum, waterfall lines, fft2_powersum_float, timf3
(and, computed from it by the reference's own host code, fft3 and baseb_raw), the AFC tables, liminfo; plus the device rings
fetched at the end."""
import os
import subprocess
import types

import numpy as np
import pytest

from paritylib import compare_with_golden, golden_itrace, load_golden, relerr
from refcases import case_params, clever_case, harness_args, lrh_config, sellim_case
from refdump import load_dump

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HARNESS = os.path.join(ROOT, "oracle", "_ref", "shim_harness")
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


@pytest.fixture(scope="module")
def harness():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "shim"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return HARNESS


def run_harness(harness, args, tmp_path, files):
    paths = {}
    for k, arr in files.items():
        paths[k] = str(tmp_path / f"{k}.bin")
        arr.tofile(paths[k])
    fo = str(tmp_path / "out.bin")
    r = subprocess.run([harness] + args(paths, fo), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "hip_open: 0" in r.stderr
    return load_dump(fo)


@pytest.mark.parametrize("name", ["n10_n12", "n10_mix1only", "n10_n12_afc", "n10_afc_mix1only", "n10_n12_fft3", "n9_n11_sin3"])
def test_patched_reference_through_the_glue_matches_the_unpatched_goldens(harness, tmp_path, name):
    d, g = case_params(name), load_golden(name)
    dump = run_harness(harness, lambda p, fo: harness_args(d, p["in"], p["lim"], fo), tmp_path, {"in": g["iq"], "lim": g["liminfo"]})
    cfg = lrh_config(d, g["iq"], max_batch=1)
    out = {k: dump[k] for k in ("fft1_float", "fft1_sumsq", "fft1_slowsum", "timf2_float", "timf2_pwr_float", "fft2_float",
                                "fft2_power_float", "fft2_powersum_float", "timf3_float")}
    out["itrace"] = golden_itrace(dump)
    out["wf_lines"] = dump["wf_lines"].reshape(-1, cfg.wf_xpixels)
    out["mixtrace"] = dump["mixtrace"].reshape(-1, 8).astype(np.float64)[:int(dump["final"][10])]
    out["cfg"] = cfg
    out["api"] = types.SimpleNamespace(get_table=lambda t, n: g[t][:n])
    if d["fft3_n"]:      # the reference's own make_fft3_all / fft3_mix2 ran on the host from the timf3 blocks the glue brought back
        out.update(fft3=dump["fft3"], fft3_ptrs=dump["fft3_ptrs"][1:], baseb_raw=dump["baseb_raw"], baseb_ptrs=dump["baseb_ptrs"])
    if d["blockpower_block"]:
        out.update(timf2_blockpower=dump["timf2_blockpower"], blockpower_ptrs=dump["blockpower_ptrs"])
    if d["afc"]:
        out["afc_tables"] = np.stack([dump["afc_fq_mid"], dump["afc_fq_slope"], dump["afc_fq_curv"], dump["afc_fq_start"]])
    # do_mix1 parks the raw second half of its newest block beyond timf3_pa until the next block adds to it (mix1.c:188-194);
    # consumers read up to timf3_pa, and that is what comes back to the host: the parked half block is left out of the comparison
    pa, blk = int(dump["final"][9]), int(dump["mixtrace"].reshape(-1, 8)[0, 6]) if dump["mixtrace"].size >= 8 else 0
    g = dict(g)
    if blk > 0:
        idx = (pa + np.arange(blk)) % out["timf3_float"].size
        out["timf3_float"] = out["timf3_float"].copy()
        g["timf3_float"] = g["timf3_float"].copy()
        out["timf3_float"][idx] = 0
        g["timf3_float"][idx] = 0
    # fft1_c accumulates the running averaging period in place at fft1_sumsq_pa (fft1.c:4126-4169); the graphs read completed
    # periods, and completed periods are what the glue brings back: an unfinished one at the end of the run is left out
    it = dump["itrace"].reshape(-1, 16)
    if it[-1, 10] != 0:
        n1 = 1 << d["n1"]
        out["fft1_sumsq"] = out["fft1_sumsq"].copy()
        g["fft1_sumsq"] = g["fft1_sumsq"].copy()
        out["fft1_sumsq"][it[-1, 9]:it[-1, 9] + n1] = 0
        g["fft1_sumsq"][it[-1, 9]:it[-1, 9] + n1] = 0
    assert np.array_equal(dump["final"], g["final"]), "final ring pointers differ"
    rep = compare_with_golden(out, g, tol=1e-5)
    # scalars the GUI reads, as the glue keeps them (blank1.c:1550-1601): noise floor and limit are in itrace; the despiked power here
    t, gt = dump["trace"].reshape(-1, 16), g["trace"].reshape(-1, 16)
    for col in (0, 1, 2, 3, 5):          # noise floor, limit, stupid_blanker_rate, despiked_pwr[0], fft1_lowlevel_fraction
        assert np.allclose(t[:, col], gt[:, col], rtol=2e-5, atol=1e-6), (col, np.abs(t[:, col] - gt[:, col]).max())
    print(name, {k: v for k, v in rep.items() if k not in ("abs_err", "abs_floor", "wf_boundary")})


def test_selective_limiter_hooks_run_on_the_device_resident_sums(harness, tmp_path):
    """fft1_update_liminfo (sellim.c:738, patched) -> hip_fft1_update_liminfo: the table after every update, as the glue publishes it
    into Linrad's liminfo[], has the reference's routing pattern; make_timf2 routes with it"""
    name = "sellim_n10_n12"
    d, sl, iq = sellim_case(name)
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz")))
    dump = run_harness(harness, lambda p, fo: [a for a in harness_args(d, p["in"], "none", fo) if not a.startswith("liminfo=")] + ["sellim=1"] +
                       [f"{k}={v}" for k, v in sl.items()], tmp_path, {"in": iq})
    n1 = 1 << d["n1"]
    got, ref = dump["liminfo_trace"].reshape(-1, n1), g["liminfo_trace"].reshape(-1, n1)
    assert got.shape == ref.shape and np.array_equal(dump["liminfo_trace_blk"], g["liminfo_trace_blk"])
    assert np.array_equal(np.sign(got), np.sign(ref)), int(np.sum(np.sign(got) != np.sign(ref)))
    pos = ref > 0
    assert np.max(np.abs(got[pos] - ref[pos]) / ref[pos]) <= 2e-6
    it, gi = dump["itrace"].reshape(-1, 16), g["itrace"].reshape(-1, 16)
    for col in (0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 15):     # pointers, fft1_lowlevel_points, noise floor, limit, fft1_liminfo_cnt
        assert np.array_equal(it[:, col], gi[:, col]), col
    assert relerr(dump["fft1_slowsum"], g["fft1_slowsum"]) <= 1e-5
    assert np.array_equal(dump["timf2_pwr_float"] == 0, g["timf2_pwr_float"] == 0)


def test_linear_blanker_tables_reach_the_device_through_the_glue(harness, tmp_path):
    """init_blanker's tables (buf.c:1771-2057, built by the reference itself in the harness) are handed over by hip_first_noise_blanker
    when hg.clever_bln_mode is set: resume pointer, fitted-pulse counter, thresholds equal the unpatched reference's call by call"""
    name = "clever_n10_n12"
    d, cl, iq, lim, des = clever_case(name)
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz")))
    dump = run_harness(harness, lambda p, fo: harness_args(d, p["in"], p["lim"], fo) + ["clever=1", f"desired={p['des']}", f"clever_factor={cl['clever_factor']}"],
                       tmp_path, {"in": iq, "lim": lim, "des": des})
    it, gi = dump["itrace"].reshape(-1, 16), g["itrace"].reshape(-1, 16)
    for col in (0, 1, 2, 3, 6, 8, 12, 13):                 # timf2_pa, timf2p_fit, timf2_pn2, timf2_px, blanker_points, fft2_na, noise floor, limit
        assert np.array_equal(it[:, col], gi[:, col]), col
    t, gt = dump["trace"].reshape(-1, 16), g["trace"].reshape(-1, 16)
    upd = it[:, 7] == 0                                     # calls that ended with a threshold update: the glue reads the scalars back there
    assert np.array_equal(t[upd, 7], gt[upd, 7]) and np.any(gt[upd, 7] > 0), "clever_blanker_rate"
    n1 = 1 << d["n1"]
    keep = np.ones(dump["timf2_float"].size, bool)
    keep[(int(dump["final"][3]) + np.arange(4 * (n1 // 2))) % keep.size] = False
    assert relerr(dump["timf2_float"] * keep, g["timf2_float"] * keep) <= 1e-5
    assert np.array_equal((dump["timf2_pwr_float"] == 0) & keep[::4], (g["timf2_pwr_float"] == 0) & keep[::4])


def test_glue_refuses_what_version_21_does_not_serve(harness, tmp_path):
    """hip_open answers non-zero (-> lirerr(1463), wcw.c hunk) instead of letting host code run on rings that stay empty"""
    name = "n10_n12"
    d, g = case_params(name), load_golden(name)
    paths = {}
    for k, arr in (("in", g["iq"]), ("lim", g["liminfo"])):
        paths[k] = str(tmp_path / f"{k}.bin")
        arr.tofile(paths[k])
    r = subprocess.run([harness] + harness_args(d, paths["in"], paths["lim"], str(tmp_path / "o.bin")) + ["shim_refuse=1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and '"hip_open": 106' in r.stdout, r.stdout + r.stderr
