"""Selective-limiter parity machinery shared by the oracle (CPU) and HIP (GPU) tests: drive a SELLIM case block by block in
the harness's order -- fft1_b, fft1_c, make_timf2, first_noise_blanker, fft2 / mix1 while data is released, then
fft1_update_liminfo when fft1_c has completed an averaging period (wcw.c:1124-1128) -- and compare with the reference golden."""
import os

import numpy as np

from linrad_amd import abi
from linrad_amd.abi import default_sellim
from refcases import lrh_config, sellim_case

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))


def sellim_params(cfg, g):
    sp, sf = g["sellim_params"], g["sellim_fparams"]
    return default_sellim(cfg, sellim_maxlevel=int(sp[1]), spek_avgnum=int(sp[2]), liminfo_group_points=int(sp[3]),
                          fft1_first_point=int(sp[4]), fft1_last_point=int(sp[5]), fft1_first_inband=int(sp[6]),
                          fft1_last_inband=int(sp[7]), baseband_bw_fftxpts=int(sp[8]), sellim_par2=int(sp[9]), sellim_par3=int(sp[10]),
                          sellim_par4=int(sp[11]), sellim_par5=int(sp[12]), sellim_par6=int(sp[13]), sellim_par7=int(sp[14]),
                          sellim_par8=int(sp[15]), fft1_blocktime=float(sf[0]), blanker_ston_fft1=float(sf[1]), exact_stats=1,
                          **({"blanker_ston_fft2": float(g["sellim2_fparams"][0]), "fft2_blocktime": float(g["sellim2_fparams"][1])} if "sellim2_fparams" in g else {}),
                          **({"sellim_par1": int(g["sellim2_par1"][0])} if "sellim2_par1" in g else {}))


def median_decision_margin(api, cfg, par):
    """hg.sellim_par1 = 0 (sellim.c:171-268): a bin goes strong when one of its fft2 sub-bins has powersum * wg_waterf_yfac above
    blanker_ston_fft2 * the median of all of them.  Returns (min over the bins of |value / threshold - 1|, the median) for the call about to
    be made: a golden whose decisions sit within float32 noise of that threshold pins rounding, not the algorithm (the start-up
    transient, where the sums are at the noise floor of an empty fft2 input and two float32 implementations differ by 4e-4, is where
    it happens) -- tests/golden/make_golden_sellim.py refuses such a case"""
    n1, n2 = 1 << cfg.fft1_n, 1 << cfg.fft2_n
    ps = api.export(abi.RING_FFT2_POWERSUM)[:n2].astype(np.float32)
    f = (ps.reshape(n1, n2 // n1) * api.get_table("wg_waterf_yfac", n1)[:, None].astype(np.float32)).reshape(-1)
    t1 = np.float32(par.blanker_ston_fft2) * np.sort(f)[n2 // 2 - 1]
    med = float(np.sort(f)[n2 // 2 - 1])
    return (float(np.min(np.abs(f / t1 - 1))) if t1 > 0 else 1.0), med


def run(open_fn, name, g, margins=None):
    d, _, iq = sellim_case(name)
    assert np.array_equal(iq, g["iq"])
    cfg = lrh_config(d, iq)
    api = open_fn(cfg)
    api.timf1_write(iq)
    api.set_mix1_selfreq(d["fq"])
    par = sellim_params(cfg, g)
    trace, blks, low, trace2, blks2, amp = [], [], [], [], [], []
    cnt = cnt2 = 0
    both = "liminfo_trace2" in g
    for b in range(d["nblk"]):
        api.fft1_b(1), api.fft1_c(1), api.make_timf2(1)
        api.first_noise_blanker()
        for _ in range(api.fft2_available()):
            api.make_fft2(1)
            api.fft2_mix1_fixed(1)
        low.append(api.p.fft1_lowlevel_points)
        if api.p.fft1_liminfo_cnt != cnt:
            api.fft1_update_liminfo(par)
            cnt = api.p.fft1_liminfo_cnt
            trace.append(api.get_liminfo())
            blks.append(b)
            amp.append(api.liminfo_amplitude_factor())
        if both and api.p.fft2_liminfo_cnt != cnt2:              # wcw.c:1129-1133
            if margins is not None:
                margins.append(median_decision_margin(api, cfg, par))
            api.fft2_update_liminfo(par)
            cnt2 = api.p.fft2_liminfo_cnt
            trace2.append(api.get_liminfo())
            blks2.append(b)
            amp.append(api.liminfo_amplitude_factor())
    return dict(api=api, cfg=cfg, d=d, trace=np.array(trace), blks=np.array(blks), low=np.array(low), trace2=np.array(trace2), blks2=np.array(blks2), amp=np.array(amp, np.float32),
                timf2=api.export(abi.RING_TIMF2_FLOAT), pwr=api.export(abi.RING_TIMF2_PWR), timf3=api.export(abi.RING_TIMF3_FLOAT),
                slowsum=api.export(abi.RING_FFT1_SLOWSUM))


def compare(out, g, tol, value_tol=2e-6):
    """routing pattern (weak / strong-unity / strong-attenuated per bin) exact after every update; attenuation values to
    value_tol; the weak-bin counts make_timf2 reports; the rings the routing shapes to tol"""
    n1 = out["api"].N1
    ref = g["liminfo_trace"].reshape(-1, n1)
    assert np.array_equal(out["blks"], g["liminfo_trace_blk"][:len(out["blks"])]) and len(out["blks"]) == ref.shape[0]
    got = out["trace"]
    rep = {"updates": int(ref.shape[0]), "pattern_mismatch_bins": int(np.sum(np.sign(got) != np.sign(ref)))}
    assert rep["pattern_mismatch_bins"] == 0, rep
    pos = ref > 0
    rep["attenuated_bins_last"] = int(pos[-1].sum())
    rep["value_err"] = float(np.max(np.abs(got[pos] - ref[pos]) / ref[pos])) if pos.any() else 0.0
    assert rep["value_err"] <= value_tol, rep
    it = g["itrace"].reshape(-1, 16)
    assert np.array_equal(out["low"], it[:, 11]), "fft1_lowlevel_points trace differs"
    if "amp_factor_trace" in g:                             # liminfo_amplitude_factor after every update (selfreq_liminfo, sellim.c:119-155)
        ra = g["amp_factor_trace"][:len(out["amp"])]
        rep["amp_factor_err"] = float(np.max(np.abs(out["amp"] - ra))) if len(ra) else 0.0
        assert len(ra) == len(out["amp"]) and rep["amp_factor_err"] <= 1e-6, rep
    if "liminfo_trace2" in g:                               # fft2_update_liminfo's tables
        ref2 = g["liminfo_trace2"].reshape(-1, n1)
        assert np.array_equal(out["blks2"], g["liminfo_trace2_blk"][:len(out["blks2"])]) and len(out["blks2"]) == ref2.shape[0]
        rep["updates2"] = int(ref2.shape[0])
        rep["pattern_mismatch_bins2"] = int(np.sum(np.sign(out["trace2"]) != np.sign(ref2)))
        assert rep["pattern_mismatch_bins2"] == 0, rep
        pos2 = ref2 > 0
        rep["value_err2"] = float(np.max(np.abs(out["trace2"][pos2] - ref2[pos2]) / ref2[pos2])) if pos2.any() else 0.0
        assert rep["value_err2"] <= value_tol, rep

    def rel(a, b):
        a, b = a.astype(np.float64), b.astype(np.float64)
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    keep = np.ones(out["timf2"].size, bool)                   # sin^2 overlap: pending half beyond timf2_pa (paritylib)
    keep[(out["api"].p.timf2_pa + np.arange(4 * (n1 // 2))) % keep.size] = False
    rep["timf2"] = rel(out["timf2"] * keep, g["timf2_float"] * keep)
    rep["slowsum"] = rel(out["slowsum"], g["fft1_slowsum"])
    rep["cleared_equal"] = bool(np.array_equal(out["pwr"] == 0, g["timf2_pwr_float"] == 0))
    assert rep["timf2"] <= tol and rep["slowsum"] <= tol, rep
    return rep


def run_dsp(open_fn, name, g, batch=4, in_call=True, fft2_too=True, rounds=None, env=None):
    """the same case through lrh_wideband_dsp in rounds of `batch` blocks: with the limiter calls inside the call
    (lrh_wideband_limiter) or made by the caller after every round, the way wcw.c:1124-1133 makes them after every pass"""
    d, _, iq = sellim_case(name)
    cfg = lrh_config(d, iq, max_batch=batch)
    api = open_fn(cfg)
    api.timf1_write(iq)
    api.set_mix1_selfreq(d["fq"])
    par = sellim_params(cfg, g)
    nr = rounds or d["nblk"] // batch
    if in_call:
        api.wideband_limiter(par, fft2_too)
        api.wideband_dsp(nr * batch, batch)
    else:
        c1 = c2 = 0
        for _ in range(nr):
            api.wideband_dsp(batch, batch)
            if api.p.fft1_liminfo_cnt != c1:
                api.fft1_update_liminfo(par)
                c1 = api.p.fft1_liminfo_cnt
            if fft2_too and api.p.fft2_liminfo_cnt != c2:
                api.fft2_update_liminfo(par)
                c2 = api.p.fft2_liminfo_cnt
    out = dict(lim=api.get_liminfo(), amp=api.liminfo_amplitude_factor(), timf2=api.export(abi.RING_TIMF2_FLOAT), pwr=api.export(abi.RING_TIMF2_PWR),
               timf3=api.export(abi.RING_TIMF3_FLOAT), p=api.p.as_dict(), low=api.p.fft1_lowlevel_points)
    api.close()
    return out
