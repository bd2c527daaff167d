"""Parity cases shared by tests/golden/make_golden.py (which runs the compiled reference here) and the tests.

A case = sizes + call pattern + seeded input.  The same dict builds the reference-harness command line and the
LrhConfig for the oracle / HIP path, so all three run the same job.
"""
import numpy as np

from linrad_amd.abi import default_config


def level_gain(n1, att_n, sigma=64.0, target_pwr=400.0):
    """FIRST_FFT_GAIN from the level plan of SURVEY.md 8(d): weak-noise power in timf2_pwr ~ target."""
    N1 = 1 << n1
    A = 1.0 / np.sqrt(3.0 / 8.0)
    g_per_gain = A * N1 * 2.0 ** (-att_n) / (150.0 * N1 ** 0.6)
    # power = 2*sigma^2*g^2
    g = np.sqrt(target_pwr / (2 * sigma * sigma))
    return max(1, int(round(g / g_per_gain)))


CASES = {
    # small, fully stored golden: every ring wraps at least once
    "n8_n10": dict(n1=8, n2=10, mixred=4, nblk=96, avg1num=3, avg2num=3, att_n=2, bln_interval=3, bln_avgnum=8,
                   fq=333.37, wf_avgnum=2, wf_mode=1, seed=11, timf2pow_log2=13, sumsq_blocks=8,
                   strong=[(-40.25, 9000.0), (100.0, 5000.0), (17.0, 900.0)], weak=[(77.5, 60.0), (-90.0, 30.0)],
                   pulse_period=997, lim_halfwidth=3),
    # mid size, stored as projections + strided samples
    "n10_n12": dict(n1=10, n2=12, mixred=6, nblk=80, avg1num=5, avg2num=4, att_n=4, bln_interval=4, bln_avgnum=16,
                    fq=2200.3, wf_avgnum=2, wf_mode=2, seed=12, timf2pow_log2=15, sumsq_blocks=8,
                    strong=[(-300.25, 9000.0), (37.0, 1000.0), (200.5, 300.0)], weak=[(-100.0, 50.0), (411.3, 25.0)],
                    pulse_period=1999, lim_halfwidth=3, blockpower_block=4 * 96),
    # N2 < N1 (BASELINE.json config 2 shape, scaled down), no-window fft2 exercises interleave 0 in mix1
    "n11_n9_nowin2": dict(n1=11, n2=9, mixred=4, nblk=40, avg1num=4, avg2num=2, att_n=5, bln_interval=4, bln_avgnum=16,
                          fq=100.2, wf_avgnum=3, wf_mode=-2, seed=13, timf2pow_log2=14, sumsq_blocks=4,
                          sinpow2=0, strong=[(512.5, 7000.0), (-200.0, 700.0)], weak=[(300.0, 40.0)], pulse_period=3001,
                          lim_halfwidth=4),
    # other fft1 window (sin^3): centre-part x inverted-window branch of timf2, interleave from formula
    "n9_n11_sin3": dict(n1=9, n2=11, mixred=5, nblk=64, avg1num=2, avg2num=2, att_n=3, bln_interval=4, bln_avgnum=16,
                        fq=900.0, wf_avgnum=1, wf_mode=1, seed=14, timf2pow_log2=14, sumsq_blocks=4,
                        sinpow1=3, strong=[(60.0, 8000.0), (-111.0, 500.0)], weak=[(150.5, 45.0)], pulse_period=0,
                        lim_halfwidth=3),
    # fft2 larger than one workgroup's LDS: the four-step path (N2 = 32768 = 256 x 128), reference-consistent N2 = 4*N1
    "n13_n15_big2": dict(n1=13, n2=15, mixred=6, nblk=24, avg1num=4, avg2num=2, att_n=6, bln_interval=2, bln_avgnum=4,
                         fq=9000.4, wf_avgnum=1, wf_mode=4, seed=15, timf2pow_log2=18, sumsq_blocks=8,
                         strong=[(2048.0, 8000.0), (-1000.5, 600.0)], weak=[(500.0, 40.0)], pulse_period=7919,
                         lim_halfwidth=3, golden_stride=11),
    # fft1_size 32768, the reference's maximum with the second fft on (buf.c:335): four-step fft1 / timf2 on the device; fft2 131072 = 4 x fft1
    "n15_n17_big1": dict(n1=15, n2=17, mixred=7, nblk=20, avg1num=4, avg2num=2, att_n=7, bln_interval=2, bln_avgnum=4,
                         fq=73536.3, wf_avgnum=1, wf_mode=8, seed=28, timf2pow_log2=20, sumsq_blocks=8, max_fft2n=4,
                         strong=[(8192.0, 8000.0), (-4000.5, 600.0)], weak=[(2000.0, 40.0), (-9000.25, 30.0)], pulse_period=31013,
                         lim_halfwidth=3, golden_stride=41),
    # fft3 behind mix1: make_fft3_all's transform part and fft3_mix2's filter / decimate part (mixer_mode 1), both run by
    # the compiled reference (the harness makes fft3_mix2 return at its thread-command check, mix2.c:749)
    "n10_n12_fft3": dict(n1=10, n2=12, mixred=5, nblk=120, avg1num=5, avg2num=4, att_n=4, bln_interval=4, bln_avgnum=16,
                         fq=2200.3, wf_avgnum=2, wf_mode=1, seed=17, timf2pow_log2=15, sumsq_blocks=8,
                         strong=[(-300.25, 9000.0), (37.0, 1000.0)], weak=[(38.6, 60.0), (411.3, 25.0)],
                         pulse_period=1999, lim_halfwidth=3, fft3_n=8, fft3_sinpow=2, mix2_n=6, max_fft3n=8, mix2=1),
    # ... and with THIRD_FFT_SINPOW = 3: fft3 windowed sin^3, mix2's transforms joined through the crossover functions of
    # prepare_mixer(&mix2, ..) (mix2.c:177-216; buf.c:55-111) instead of the sin^2 overlap-add
    "n10_n12_fft3_sin3": dict(n1=10, n2=12, mixred=5, nblk=120, avg1num=5, avg2num=4, att_n=4, bln_interval=4, bln_avgnum=16,
                              fq=2200.3, wf_avgnum=2, wf_mode=1, seed=17, timf2pow_log2=15, sumsq_blocks=8,
                              strong=[(-300.25, 9000.0), (37.0, 1000.0)], weak=[(38.6, 60.0), (411.3, 25.0)],
                              pulse_period=1999, lim_halfwidth=3, fft3_n=8, fft3_sinpow=3, mix2_n=6, max_fft3n=8, mix2=1),
    # ... and with bg.mixer_mode = 2: baseb_raw from the FIR decimator on timf3 (mix2.c:217-246) instead of the filter on fft3's bins
    "n10_n12_fft3_fir": dict(n1=10, n2=12, mixred=5, nblk=120, avg1num=5, avg2num=4, att_n=4, bln_interval=4, bln_avgnum=16,
                             fq=2200.3, wf_avgnum=2, wf_mode=1, seed=17, timf2pow_log2=15, sumsq_blocks=8,
                             strong=[(-300.25, 9000.0), (37.0, 1000.0)], weak=[(38.6, 60.0), (411.3, 25.0)],
                             pulse_period=1999, lim_halfwidth=3, fft3_n=8, fft3_sinpow=2, mix2_n=6, max_fft3n=8, mix2=1, mixer_mode=2),
    # int32 input (DWORD_INPUT: 18/24-bit hardware, expanded .raw recordings) with an I/Q sample skew (fft1.c:470-635)
    "n10_n12_dword": dict(n1=10, n2=12, mixred=6, nblk=48, avg1num=5, avg2num=4, att_n=8, gain=15, bln_interval=4, bln_avgnum=16,
                          fq=2200.3, wf_avgnum=2, wf_mode=1, seed=18, timf2pow_log2=15, sumsq_blocks=8, sigma=256.0,
                          strong=[(-300.25, 6000.0), (37.0, 2000.0)], weak=[(-100.0, 200.0), (411.3, 100.0)],
                          pulse_period=1999, pulse_amp=100000.0, lim_halfwidth=3, dword=1, sample_shift=-1),
    "n9_n11_shift": dict(n1=9, n2=11, mixred=5, nblk=40, avg1num=2, avg2num=2, att_n=3, bln_interval=4, bln_avgnum=16,
                         fq=900.0, wf_avgnum=1, wf_mode=1, seed=19, timf2pow_log2=14, sumsq_blocks=4,
                         strong=[], weak=[(150.5, 45.0), (60.0, 120.0)], pulse_period=1499, lim_halfwidth=3, sample_shift=2),
    # fft2 window sin^3: fft2 interleave from the formula, mix1 joins its blocks with crossover windows (mix1.c:196-262)
    "n10_n12_xover": dict(n1=10, n2=12, mixred=5, nblk=64, avg1num=5, avg2num=4, att_n=4, bln_interval=4, bln_avgnum=16,
                          fq=2200.3, wf_avgnum=2, wf_mode=1, seed=21, timf2pow_log2=15, sumsq_blocks=8, sinpow2=3,
                          strong=[(-300.25, 9000.0)], weak=[(38.6, 80.0), (37.0, 60.0), (411.3, 25.0)],
                          pulse_period=1999, lim_halfwidth=3),
    # mirrored passband (fft1_direction = -1, fft1.c:3660-3679)
    "n9_n11_dir": dict(n1=9, n2=11, mixred=5, nblk=40, avg1num=2, avg2num=2, att_n=3, bln_interval=4, bln_avgnum=16,
                       fq=900.0, wf_avgnum=1, wf_mode=1, seed=22, timf2pow_log2=14, sumsq_blocks=4, direction=-1,
                       strong=[(-60.0, 2500.0)], weak=[(-150.5, 45.0), (33.0, 80.0)], pulse_period=1499, lim_halfwidth=4),
    # I/Q mirror-image calibration (CALIQ, fft1.c:3598-3658) with a synthetic fft1_foldcorr, both passband directions
    "n9_n11_iqcal": dict(n1=9, n2=11, mixred=5, nblk=40, avg1num=2, avg2num=2, att_n=3, bln_interval=4, bln_avgnum=16,
                         fq=900.0, wf_avgnum=1, wf_mode=1, seed=23, timf2pow_log2=14, sumsq_blocks=4, foldcorr_seed=5,
                         strong=[(60.0, 8000.0)], weak=[(150.5, 45.0), (-33.0, 80.0)], pulse_period=1499, lim_halfwidth=3,
                         lim_mirror=1),
    "n8_n10_iqcal_rev": dict(n1=8, n2=10, mixred=4, nblk=48, avg1num=3, avg2num=3, att_n=2, bln_interval=3, bln_avgnum=8,
                             fq=333.37, wf_avgnum=2, wf_mode=1, seed=24, timf2pow_log2=13, sumsq_blocks=8, foldcorr_seed=6,
                             direction=-1, strong=[(40.25, 9000.0)], weak=[(77.5, 60.0), (-90.0, 30.0)], pulse_period=997,
                             lim_halfwidth=3, lim_mirror=1),
    # AFC variants of mix1 (fft2_mix1_afc / fft1_mix1_afc with do_mix1_afc's table bookkeeping): the frequency wanders
    # +-1.5 bins and jumps by 3 bins for a while, which exercises the limited-curvature branch (mix1.c:706-747)
    "n10_n12_afc": dict(n1=10, n2=12, mixred=6, nblk=48, avg1num=5, avg2num=4, att_n=4, bln_interval=4, bln_avgnum=16,
                        fq=2200.3, wf_avgnum=2, wf_mode=1, seed=25, timf2pow_log2=15, sumsq_blocks=8, afc=1,
                        strong=[(-300.25, 9000.0)], weak=[(-100.0, 50.0), (152.0, 120.0)], pulse_period=1999, lim_halfwidth=3),
    "n10_afc_mix1only": dict(n1=10, n2=10, mixred=4, nblk=64, avg1num=3, avg2num=2, att_n=4, bln_interval=4, bln_avgnum=16,
                             fq=700.3, wf_avgnum=1, wf_mode=1, seed=26, timf2pow_log2=13, sumsq_blocks=8, second_fft=0,
                             afc=1, strong=[(-100.0, 3000.0)], weak=[(188.3, 400.0)], pulse_period=0, lim_halfwidth=3),
    # real samples (ui.rx_input_mode without IQ_DATA): fft1 version 2 = fft1_reherm_dit_one (fft1_re.c:32), 2*N1 reals per
    # transform, bins 0..fs/2; carriers are given in bins of that half spectrum.  Second case: int32 samples, mirrored passband
    "n9_n11_real": dict(n1=9, n2=11, mixred=5, nblk=40, avg1num=2, avg2num=2, att_n=3, bln_interval=4, bln_avgnum=16,
                        fq=1300.0, wf_avgnum=1, wf_mode=1, seed=31, timf2pow_log2=14, sumsq_blocks=4, real=1, gain=6,
                        strong=[(100.3, 8000.0)], weak=[(325.2, 60.0), (33.0, 80.0)], pulse_period=1499, lim_halfwidth=3),
    "n8_n10_real_dword_rev": dict(n1=8, n2=10, mixred=4, nblk=48, avg1num=3, avg2num=3, att_n=6, gain=68, bln_interval=3, bln_avgnum=8,
                                  fq=333.37, wf_avgnum=2, wf_mode=1, seed=32, timf2pow_log2=13, sumsq_blocks=8, real=1, dword=1,
                                  direction=-1, sigma=256.0, strong=[(60.25, 6000.0)], weak=[(171.5, 200.0), (90.0, 100.0)],
                                  pulse_period=997, pulse_amp=100000.0, lim_halfwidth=3),
    # second fft disabled (the reference's own default, uivar.c:371): fft1 -> fft1_c -> fft1_mix1_fixed
    "n10_mix1only": dict(n1=10, n2=10, mixred=4, nblk=48, avg1num=3, avg2num=2, att_n=4, bln_interval=4, bln_avgnum=16,
                         fq=700.3, wf_avgnum=1, wf_mode=1, seed=16, timf2pow_log2=13, sumsq_blocks=8, second_fft=0,
                         strong=[(-100.0, 3000.0)], weak=[(188.3, 400.0)], pulse_period=0, lim_halfwidth=3),
    # fft1_size 65536, the reference's maximum, second fft off (fft0.c:1162-1169): four-step fft1 at 256 x 256, mix1 on the fft1 spectra
    "n16_mix1only_big": dict(n1=16, n2=10, mixred=7, nblk=10, avg1num=3, avg2num=2, att_n=4, bln_interval=4, bln_avgnum=16,
                             fq=45000.3, wf_avgnum=1, wf_mode=1, seed=29, timf2pow_log2=20, sumsq_blocks=4, second_fft=0,
                             strong=[(-9000.0, 3000.0), (12345.5, 900.0)], weak=[(12400.25, 150.0)], pulse_period=0, lim_halfwidth=3,
                             golden_stride=61),
}


def case_params(name):
    d = dict(sinpow1=2, sinpow2=2, gain=None, stupid=1, max_fft1n=8, max_fft2n=4, wf_first=0, wf_pixels=0,
             pulsewidth=0, blnfit_range=48, noise_floor=200, sigma=64.0, pulse_amp=20000.0, pulse_len=3, golden_stride=1,
             second_fft=1, blockpower_block=0, blockpower_size=1024, fft3_n=0, fft3_sinpow=2, mix2_n=0, max_fft3n=8,
             dword=0, sample_shift=0, direction=1, foldcorr_seed=0, lim_mirror=0, afc=0, afc_bw=20.0, mix2=0, real=0, mixer_mode=1)
    d.update(CASES[name])
    if d["gain"] is None:
        # DWORD input is left-justified (x 2^14) and make_filcorrstart divides by 4096*12 (fft1.c:4656-4663)
        d["gain"] = level_gain(d["n1"], d["att_n"], d["sigma"] * (16384.0 / 49152.0 if d["dword"] else 1.0))
        if d["real"]:            # 2*N1 reals of variance sigma^2 per transform carry the power of N1 complex samples of sigma per component
            d["gain"] = max(1, d["gain"])
    return d


def interleave(n, sinpow):
    if sinpow == 0:
        return 0
    ratio = 0.625 if sinpow == 9 else 0.8 if sinpow == 8 else 2 * np.arcsin(0.5 ** (1.0 / sinpow)) / np.pi
    return int(1 + np.float32(ratio) * (1 << n)) & 0xfffe


def make_input(d):
    """Seeded synthetic IQ: noise + carriers + impulses (numpy; only used for the small parity cases)."""
    N1 = 1 << d["n1"]
    M1 = N1 - interleave(d["n1"], d["sinpow1"])
    n = M1 * d["nblk"] + 2 * N1
    rng = np.random.default_rng(d["seed"])
    if d["real"]:
        # 2n real samples at twice the rate; a carrier at bin k of the N1-bin half spectrum is cos(2 pi k t / (2 N1))
        t = np.arange(2 * n)
        x = rng.normal(0, d["sigma"], 2 * n)
        for k, a in d["strong"] + d["weak"]:
            x += a * np.cos(2 * np.pi * k * t / (2 * N1) + rng.uniform(0, 6.28))
        if d["pulse_period"]:
            for s in range(d["pulse_period"] // 2, 2 * n - 4, 2 * d["pulse_period"]):     # two-sample impulses: flat spectrum
                x[s:s + 2] += d["pulse_amp"] * np.cos(rng.uniform(0, 6.28)) * np.array([1.0, -0.5])
        lim = 131071 if d["dword"] else 32767
        out = np.clip(np.round(x), -lim, lim).astype(np.int32 if d["dword"] else np.int16)
        return ((out << 14) | 0x2000) if d["dword"] else out
    t = np.arange(n)
    x = rng.normal(0, d["sigma"], n) + 1j * rng.normal(0, d["sigma"], n)
    for k, a in d["strong"] + d["weak"]:
        x += a * np.exp(2j * np.pi * k * t / N1 + 1j * rng.uniform(0, 6.28))
    if d["pulse_period"]:
        for s in range(d["pulse_period"] // 2, n - d["pulse_len"], d["pulse_period"]):
            x[s:s + d["pulse_len"]] += d["pulse_amp"] * np.exp(1j * rng.uniform(0, 6.28))
    lim = 131071 if d["dword"] else 32767            # 18-bit content, or int16
    iq = np.empty(2 * n, np.int32 if d["dword"] else np.int16)
    iq[0::2] = np.clip(np.round(x.real), -lim, lim)
    iq[1::2] = np.clip(np.round(x.imag), -lim, lim)
    if d["dword"]:                                   # left-justified in 32 bits with the half-LSB bit, like expand_rawdat (csplit.c:20-73)
        iq = (iq << 14) | 0x2000
    return iq


def make_foldcorr(d):
    """Synthetic I/Q calibration table: a few per cent, smooth in frequency (what caliq.c fits), N1 complex as floats."""
    N1 = 1 << d["n1"]
    rng = np.random.default_rng(d["foldcorr_seed"])
    k = np.arange(N1) / N1
    z = 0.03 * np.exp(2j * np.pi * (rng.uniform() + 1.5 * k)) * (1 + 0.5 * np.cos(2 * np.pi * (k + rng.uniform())))
    out = np.empty(2 * N1, np.float32)
    out[0::2], out[1::2] = z.real, z.imag
    return out


def make_liminfo(d):
    """strong/weak routing table: bins within lim_halfwidth of a strong carrier are marked strong (input to the path)."""
    N1 = 1 << d["n1"]
    lim = np.zeros(N1, np.float32)
    for k, _ in d["strong"] if d["real"] else []:       # real input: bins of the half spectrum 0..fs/2, mirrored passband counts down
        c = int(round(N1 - k if d["direction"] < 0 else k))
        lim[max(0, c - d["lim_halfwidth"]):c + d["lim_halfwidth"] + 1] = 1.0
    for k, _ in [] if d["real"] else d["strong"]:
        for kk in ([k, -k] if (d["lim_mirror"] or d["direction"] < 0) else [k]):   # mirrored passband / residual image
            c = int(round(N1 // 2 + kk))
            lim[max(0, c - d["lim_halfwidth"]):c + d["lim_halfwidth"] + 1] = 1.0
    return lim


def timf1_bytes_for(d, iq):
    b = 1
    while b < iq.nbytes:
        b <<= 1
    return b


def lrh_config(d, iq, **kw):
    N1, N2 = 1 << d["n1"], 1 << d["n2"]
    wfpix = d["wf_pixels"] or min(N2, 1024)
    c = default_config(
        d["n1"], d["n2"], fft1_sinpow=d["sinpow1"], fft2_sinpow=d["sinpow2"], fft1_gain=d["gain"],
        fft_avg1num=d["avg1num"], fft_avg2num=d["avg2num"], timf1_bytes=timf1_bytes_for(d, iq),
        max_fft1n=d["max_fft1n"], max_fft2n=d["max_fft2n"], fft1_sumsq_bufsize=d["sumsq_blocks"] * N1,
        bckfft_att_n=d["att_n"], timf2pow_size=1 << d["timf2pow_log2"], stupid_bln_mode=d["stupid"],
        blnfit_range=d["blnfit_range"], blanker_pulsewidth=d["pulsewidth"],
        timf2_noise_floor_avgnum=d["bln_avgnum"], blanker_info_update_interval=d["bln_interval"],
        blanker_min_points=N2 // 3, timf2_noise_floor=d["noise_floor"],
        waterfall_avgnum=d["wf_avgnum"], wf_first_xpoint=d["wf_first"], wf_xpixels=wfpix, wf_mode=d["wf_mode"],
        wf_lines=8, mix1_bandwidth_reduction_n=d["mixred"],
        timf3_size=16 * 2 * max(8, 1 << ((d["n2"] if d["second_fft"] else d["n1"]) - d["mixred"])),
        fftx_points_per_hz=1.0, mix1_lowest_fq=0.0, mix1_highest_fq=float(N2 if d["second_fft"] else N1), max_batch=4,
        second_fft_enable=d["second_fft"], timf2_blockpower_block=d["blockpower_block"],
        timf2_blockpower_size=d["blockpower_size"], fft3_n=d["fft3_n"], fft3_sinpow=d["fft3_sinpow"], mix2_n=d["mix2_n"],
        max_fft3n=d["max_fft3n"], baseband_size=4096, timf1_dword_input=d["dword"], sample_shift=d["sample_shift"],
        fft1_direction=d["direction"], timf1_real_input=d["real"])
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def harness_args(d, infile, limfile, outfile):
    keys = ["n1", "n2", "sinpow1", "sinpow2", "mixred", "att_n", "gain", "avg1num", "avg2num", "nblk", "max_fft1n",
            "max_fft2n", "sumsq_blocks", "stupid", "bln_interval", "bln_avgnum", "pulsewidth", "blnfit_range",
            "noise_floor", "fq", "wf_avgnum", "wf_first", "wf_pixels", "wf_mode", "timf2pow_log2", "second_fft",
            "blockpower_block", "blockpower_size", "fft3_n", "fft3_sinpow", "mix2_n", "max_fft3n", "dword", "sample_shift", "direction", "afc", "afc_bw", "mix2", "real", "mixer_mode"]
    a = [f"{k}={d[k]}" for k in keys]
    a += [f"in={infile}", f"liminfo={limfile}", f"out={outfile}"]
    return a


# ---- two RF channels in one array (the reference's own layout); the build shards them one per context
TWOCHAN = {
    "twochan_n10": dict(base="n10_n12", nblk=40, seed2=112, sky_phase=0.7, ch2_c1=float(np.float32(np.cos(0.5))),
                        ch2_c2=float(np.float32(np.sin(0.5))),
                        chain=dict(nblk=72, fft3_n=6, mix2_n=4, max_fft3n=8, mix2=1, pol=(0.8, 0.36, -0.48))),
    # two real channels per frame {a_k, b_k} (fft1_reherm_dit_two, fft1_re.c:133): channel 1 = the carriers and pulses at
    # 0.7 of channel 0's amplitude, independent noise
    "twochan_real_n9": dict(base="n9_n11_real", nblk=40, seed2=115, sky_phase=0.0, ch2_c1=1.0, ch2_c2=0.0),
    "twochan_n9_sin3": dict(base="n9_n11_sin3", nblk=32, seed2=114, sky_phase=-1.1, ch2_c1=1.0, ch2_c2=0.0,
                            chain=dict(nblk=56, fft3_n=6, mix2_n=4, max_fft3n=8, mix2=1, pol=(0.6, -0.64, 0.48))),
}


def twochan_case(name, chain=False):
    """params, frame-interleaved input {I0,Q0,I1,Q1} and liminfo of a two-channel case: channel 1 = the same carriers
    and pulses turned by sky_phase, independent noise.  chain: the run goes on through the two-channel
    blanker, make_fft2 and fft2_mix1_fixed at the base case's frequency (harness chain2=1)."""
    t = TWOCHAN[name]
    d = case_params(t["base"])
    d.update(nblk=t["nblk"], ch2_c1=t["ch2_c1"], ch2_c2=t["ch2_c2"], fq=d["fq"] if chain else -1.0, second_fft=1)
    if chain:        # longer run, fft3 + fft3_mix2 behind mix1 with the polarisation transform pg.c1..c3 = pol
        d.update(blockpower_block=0, **t["chain"])
    if d["real"]:
        x0 = make_input(d).astype(np.float64)
        rng0 = np.random.default_rng(d["seed"])
        noise0 = rng0.normal(0, d["sigma"], x0.size)                                    # the first draws of make_input
        rng = np.random.default_rng(t["seed2"])
        x1 = 0.7 * (x0 - noise0) + rng.normal(0, d["sigma"], x0.size)
        frames = np.empty(2 * x0.size, np.int16)
        frames[0::2], frames[1::2] = x0, np.clip(np.round(x1), -32767, 32767)
        return d, frames, make_liminfo(d)
    x0 = make_input(d).astype(np.float64)
    z0 = x0[0::2] + 1j * x0[1::2]
    rng0 = np.random.default_rng(d["seed"])
    n = z0.size
    noise0 = rng0.normal(0, d["sigma"], n) + 1j * rng0.normal(0, d["sigma"], n)      # the first draws of make_input
    rng = np.random.default_rng(t["seed2"])
    z1 = (z0 - noise0) * np.exp(1j * t["sky_phase"]) + rng.normal(0, d["sigma"], n) + 1j * rng.normal(0, d["sigma"], n)
    frames = np.empty(4 * n, np.int16)
    frames[0::4], frames[1::4] = x0[0::2], x0[1::2]
    frames[2::4] = np.clip(np.round(z1.real), -32767, 32767)
    frames[3::4] = np.clip(np.round(z1.imag), -32767, 32767)
    return d, frames, make_liminfo(d)


# ---- linear blanker on two coupled channels (harness channels=2 blanker2=1 clever=1): get_pulse_pol / transform_timf2_pol /
# subtract_twochan_pulse (blank1.c:232-609).  Channel 1 = channel 0's carriers and pulses times gain * e^{j sky_phase} (one
# polarisation for everything), independent noise.
CLEVER2 = {
    "clever2_n10": dict(base="clever_n10_n12", nblk=96, seed2=212, sky_phase=0.9, gain=0.75),
    "clever2_n9_only": dict(base="clever_n9_n11_only", nblk=96, seed2=213, sky_phase=-2.0, gain=1.3),
}


def clever2_case(name):
    """params, parameters of the clever case, frames {I0,Q0,I1,Q1}, liminfo, calibration target"""
    t = CLEVER2[name]
    d, cl, iq, lim, des = clever_case(t["base"])
    d.update(nblk=t["nblk"], ch2_c1=1.0, ch2_c2=0.0, fq=-1.0, second_fft=1)
    z0 = iq[0::2].astype(np.float64) + 1j * iq[1::2]
    n = z0.size
    rng0 = np.random.default_rng(d["seed"])
    noise0 = rng0.normal(0, d["sigma"], n) + 1j * rng0.normal(0, d["sigma"], n)      # the first draws of make_input
    rng = np.random.default_rng(t["seed2"])
    z1 = (z0 - noise0) * t["gain"] * np.exp(1j * t["sky_phase"]) + rng.normal(0, d["sigma"], n) + 1j * rng.normal(0, d["sigma"], n)
    frames = np.empty(4 * n, np.int16)
    frames[0::4], frames[1::4] = iq[0::2], iq[1::2]
    frames[2::4] = np.clip(np.round(z1.real), -32767, 32767)
    frames[3::4] = np.clip(np.round(z1.imag), -32767, 32767)
    return d, cl, frames, lim, des


# ---- selective limiter on (harness sellim=1): fft1_update_liminfo runs after every averaging period and make_timf2 routes with its table
SELLIM = {
    # strong carrier above the SELLIM_MAXLEVEL limit (pass 1: attenuation with tapered skirts), medium carriers picked up by the
    # noise-floor pass once spek_avgnum spectra have been seen, a keyed carrier that comes and goes (hold-off and slow release)
    "sellim_n10_n12": dict(base="n10_n12", nblk=200, maxlevel=12000, lim_groups=16, blocktime=0.0008, ston_fft1=4.0, bw_fftxpts=40,
                           keyed=(150.0, 7000.0, 60, 110), blockpower_block=0),
    # other switches: par2 (running group minima), par3 (edge groups skipped), par4, par8, selected passband protected (par7)
    "sellim_n9_n11_pars": dict(base="n9_n11_shift", nblk=160, maxlevel=4000, lim_groups=32, blocktime=0.002, ston_fft1=3.0, bw_fftxpts=24,
                               par2=1, par3=1, par4=1, par7=1, par8=1, sample_shift=0, keyed=(-77.0, 5000.0, 40, 90),
                               strong=[(101.0, 6000.0)], weak=[(150.5, 45.0), (60.0, 400.0)]),
}


# both limiters (harness sellim=1 sellim2=1): fft2_update_liminfo (sellim.c:159, par1 = 2) after every waterfall line on top of fft1_update_liminfo
SELLIM["sellim2_n10_n12"] = dict(base="n10_n12", nblk=200, maxlevel=12000, lim_groups=16, blocktime=0.0008, ston_fft1=4.0, bw_fftxpts=40,
                                 keyed=(150.0, 7000.0, 60, 110), blockpower_block=0, sellim2=1, ston_fft2=30.0)
SELLIM["sellim2_n9_n11_pars"] = dict(base="n9_n11_shift", nblk=160, maxlevel=4000, lim_groups=32, blocktime=0.002, ston_fft1=3.0, bw_fftxpts=24,
                                     par2=1, par3=1, par4=1, par7=1, par8=1, sample_shift=0, keyed=(-77.0, 5000.0, 40, 90),
                                     strong=[(101.0, 6000.0)], weak=[(150.5, 45.0), (60.0, 400.0)], sellim2=1, ston_fft2=8.0, wf_avgnum=3)


# the second limiter's older variants (hg.sellim_par1 = 0: median of all fft2 bins, sellim.c:170-281; 1: noise floor per weak-signal
# region, sellim.c:283-533); the third case puts more strong carriers into the band than there is room for regions (sellim.c:406-469)
SELLIM["sellim2v0_n10_n12"] = dict(SELLIM["sellim2_n10_n12"], par1=0, ston_fft2=42.0)
SELLIM["sellim2v1_n10_n12"] = dict(SELLIM["sellim2_n10_n12"], par1=1, ston_fft2=20.0)
SELLIM["sellim2v1_n9_n11_many"] = dict(base="n9_n11_shift", nblk=160, maxlevel=1500, lim_groups=16, blocktime=0.002, ston_fft1=3.0, bw_fftxpts=24,
                                       sample_shift=0, keyed=(-77.0, 2500.0, 40, 90), sellim2=1, par1=1, ston_fft2=12.0, wf_avgnum=3,
                                       strong=[(-200.0 + 37.0 * i, 2400.0 + 20.0 * i) for i in range(11)], weak=[(150.5, 45.0), (60.0, 400.0)])


# in-band end points inside the band (what an amplitude calibration sets, fft1.c:4631-4637) with the running group minima (par2) and the edge groups
# skipped (par3): the noise-floor pass starts its first group at fft1_first_inband and cuts the last one at fft1_last_inband + 1 (sellim.c:925-983)
SELLIM["sellim2_n9_n11_inband"] = dict(SELLIM["sellim2_n9_n11_pars"], first_inband=37, last_inband=470)


def sellim_case(name):
    """params + input of a selective-limiter case: the base case's signal plus a carrier keyed on for blocks [on, off)"""
    t = dict(SELLIM[name])
    d = case_params(t.pop("base"))
    keyed = t.pop("keyed")
    sl = {k: t.pop(k) for k in list(t) if k in ("maxlevel", "lim_groups", "blocktime", "ston_fft1", "bw_fftxpts", "par1", "par2", "par3", "par4", "par5", "par6", "par7", "par8",
                                                 "sellim2", "ston_fft2", "first_inband", "last_inband")}
    d.update(t)
    iq = make_input(d).astype(np.float64)
    N1 = 1 << d["n1"]
    M1 = N1 - interleave(d["n1"], d["sinpow1"])
    k, a, on, off = keyed
    n = iq.size // 2
    tt = np.arange(n)
    gate = ((tt >= on * M1) & (tt < off * M1)).astype(np.float64)
    z = a * gate * np.exp(2j * np.pi * k * tt / N1)
    iq[0::2] += z.real
    iq[1::2] += z.imag
    return d, sl, np.clip(np.round(iq), -32767, 32767).astype(np.int16)


# ---- spur subtraction on (harness spur=1): a carrier is acquired by the reference's own store_new_spur / spur_phase_lock and then
# tracked and subtracted by eliminate_spurs inside make_fft2
SPUR = {
    # steady carrier inside the mix1 passband: the baseband loses it (timf3 shows the subtraction)
    "spur_n10_n12": dict(base="n10_n12", nblk=160, max_fft2n=32, blockpower_block=0, spur_pnt=2193, spur_start=12, spur_speknum=8, tone=None),
    # a carrier that drifts 0.35 fft2 bins during the run (across a bin boundary): frequency drift in the loop (d2pha) and shift_spur_table
    "spur_n10_n12_drift": dict(base="n10_n12", nblk=200, max_fft2n=64, blockpower_block=0, spur_pnt=2597, spur_start=20, spur_speknum=12,
                               tone=(2600.8, 0.35, 3000.0), fq=2590.3),
    # the second fft off (Linrad's default in every mode, uivar.c:371): the spur lives in the fft1 transforms and fft1_c's AFC branch takes it
    # out (fft1afc_flag = 1: fft1.c:4196-4244; the search spectrum from the fft1 powers, :4428-4460); the carrier at fft1 bin 412 sits inside
    # the baseband that fft1_mix1_fixed cuts out
    "spur_n10_fft1": dict(base="n10_mix1only", nblk=120, max_fft1n=32, spur_pnt=409, spur_start=12, spur_speknum=8, tone=None, fq=420.3,
                          strong=[(-100.0, 600.0)]),
    # ... and at a four-step fft1 size (fft1_size 32768, k_fft1_cols / k_fft1_rows): the spur loop and the search rows work on the same ring
    "spur_n15_fft1": dict(base="n16_mix1only_big", n1=15, nblk=30, max_fft1n=16, spur_pnt=28726, spur_start=8, spur_speknum=4, tone=None, fq=28700.3,
                          strong=[(-9000.0, 3000.0), (12345.5, 600.0)], weak=[(12400.25, 150.0)], golden_stride=1, sumsq_blocks=4),
}


# the operator's clicks through the reference's whole init_spur_elimination (harness spur_click<i>_at / _pnt, spursub.c:181-343): the carrier of
# the base case (fft2 bin 2196.0) is taken on; a weaker one 3.4 bins above it is clicked next -- its window lands within four bins of the
# first (it is keyed on after the first lock: side by side the two are a broad peak, which spursearch_spectrum_cleanup wipes from the search
# spectrum, spursub.c:152-170), the ordering pass drops the weaker (the new, LAST one: by counting no_of_spurs down, without remove_spur, spursub.c:331-335); a third
# click takes a carrier BELOW the first: swap_spurs puts the list in order of frequency
SPUR_CLICKS = {
    "spur_n10_n12_clicks": dict(base="n10_n12", nblk=430, max_fft2n=32, blockpower_block=0, spur_speknum=8, tone=None,
                                tones=[(3000.2, 0.0, 1000.0), (3002.9, 0.42, 800.0), (1300.3, 0.0, 500.0)], clicks=[(30, 3000), (84, 3003), (90, 1300)]),
    # the second carrier the stronger of the two: the FIRST spur is the one dropped, remove_spur(0) moves the new one into its slot (spur.c:596)
    "spur_n10_n12_clicks_strong": dict(base="n10_n12", nblk=430, max_fft2n=32, blockpower_block=0, spur_speknum=8, tone=None,
                                       tones=[(3000.2, 0.0, 1000.0), (3002.6, 0.42, 2000.0), (1300.3, 0.0, 500.0)], clicks=[(30, 3000), (84, 3002), (90, 1300)]),
}


def spur_case(name):
    t = dict(SPUR[name] if name in SPUR else SPUR_CLICKS[name])
    d = case_params(t.pop("base"))
    if "clicks" in t:
        sp = {"spur_speknum": t.pop("spur_speknum")}
        for i, (at, pnt) in enumerate(t.pop("clicks")):
            sp[f"spur_click{i + 1}_at"], sp[f"spur_click{i + 1}_pnt"] = at, pnt
    else:
        sp = {k: t.pop(k) for k in ("spur_pnt", "spur_start", "spur_speknum")}
    tone = t.pop("tone")
    tones = t.pop("tones", [])
    d.update(t)
    iq = make_input(d).astype(np.float64)
    lim = make_liminfo(d)
    for tn in tones:                          # steady extra carriers (fft2 bin, switched on at this fraction of the run, amplitude), routed strong
        N1, N2 = 1 << d["n1"], 1 << d["n2"]
        c = int(round(N1 // 2 + (tn[0] - N2 / 2) * N1 / N2))
        lim[c - d["lim_halfwidth"]:c + d["lim_halfwidth"] + 1] = 1.0
        tt = np.arange(iq.size // 2, dtype=np.float64)
        ph = 2 * np.pi * ((tn[0] - N2 / 2) / N2) * tt
        on = tt >= tn[1] * tt.size
        iq[0::2] += tn[2] * np.cos(ph) * on
        iq[1::2] += tn[2] * np.sin(ph) * on
    if tone is not None:                      # (fft2 bin at the start, drift in fft2 bins over the run, amplitude)
        N1, N2 = 1 << d["n1"], 1 << d["n2"]
        c = int(round(N1 // 2 + (tone[0] - N2 / 2) * N1 / N2))       # routed with the strong signals like the other carriers
        lim[c - d["lim_halfwidth"]:c + d["lim_halfwidth"] + 1] = 1.0
        n = iq.size // 2
        tt = np.arange(n, dtype=np.float64)
        f0 = (tone[0] - N2 / 2) / N2              # cycles per sample
        df = tone[1] / N2 / n
        ph = 2 * np.pi * (f0 * tt + 0.5 * df * tt * tt)
        iq[0::2] += tone[2] * np.cos(ph)
        iq[1::2] += tone[2] * np.sin(ph)
    return d, sp, np.clip(np.round(iq), -32767, 32767).astype(np.int16), lim


# ---- linear ("clever") blanker on (harness clever=1): init_blanker (buf.c:1786-2057) derives the pulse tables from a synthetic amplitude
# calibration fft1_desired; the input carries band-limited pulses of exactly that response at fractional positions (fitted and
# subtracted), close pairs (unresolved: flagged 65, left to the stupid blanker) and a few rectangular pulses of another shape
CLEVER = {
    "clever_n10_n12": dict(base="n10_n12", nblk=96, blockpower_block=0, clever_factor=12.0, edge=0.18, pulses=60, pulse_seed=5, amp=(2500.0, 22000.0),
                           pairs=6, rects=4, pulse_period=0),
    # narrower passband (longer pulse response, larger fit sizes), lower threshold, the stupid blanker off: only fitted pulses leave
    "clever_n9_n11_only": dict(base="n9_n11_shift", nblk=96, clever_factor=8.0, edge=0.3, pulses=40, pulse_seed=6, amp=(1500.0, 9000.0),
                               pairs=3, rects=0, pulse_period=0, stupid=0, sample_shift=0),
}


# linear blanker and selective limiter together: selfreq_liminfo's liminfo_amplitude_factor (sellim.c:119-155) scales the pulse the
# blanker subtracts (blank1.c:143-144), so the two stages are coupled through one float that changes with every limiter update
CLEVER["clever_sellim_n10_n12"] = dict(base="n10_n12", nblk=120, blockpower_block=0, clever_factor=12.0, edge=0.18, pulses=70, pulse_seed=7, amp=(2500.0, 22000.0),
                                       pairs=4, rects=2, pulse_period=0,
                                       sellim=dict(maxlevel=12000, lim_groups=16, blocktime=0.0008, ston_fft1=30.0, bw_fftxpts=40))


def clever_desired(n1, edge):
    """amplitude calibration target: flat passband, raised-cosine skirts over `edge` of the band at either side (fft1 bin order)"""
    N1 = 1 << n1
    x = np.arange(N1) / N1
    w = np.ones(N1)
    e = x < edge
    w[e] = 0.5 - 0.5 * np.cos(np.pi * x[e] / edge)
    e = x > 1 - edge
    w[e] = 0.5 - 0.5 * np.cos(np.pi * (1 - x[e]) / edge)
    return w.astype(np.float32)


def clever_case(name):
    t = dict(CLEVER[name])
    d = case_params(t.pop("base"))
    cl = {k: t.pop(k) for k in ("clever_factor", "edge", "pulses", "pulse_seed", "amp", "pairs", "rects")}
    cl["sellim"] = t.pop("sellim", None)
    d.update(t)
    iq = make_input(d).astype(np.float64)
    N1 = 1 << d["n1"]
    des = clever_desired(d["n1"], cl["edge"])
    n = iq.size // 2
    rng = np.random.default_rng(cl["pulse_seed"])
    H = 256                                           # half length of the synthesised response
    k = np.fft.fftfreq(N1)                            # bin i of fft1 (DC at N1/2) is baseband frequency (i - N1/2)/N1
    spec = np.fft.ifftshift(des.astype(np.float64))

    def shaped(frac):                                 # response to a unit impulse at fractional delay frac, centred at index H
        h = np.fft.ifft(spec * np.exp(-2j * np.pi * k * (frac + H)))
        return h[:2 * H] / np.abs(np.fft.ifft(spec)[0])

    z = np.zeros(n, complex)
    pos = np.sort(rng.choice(np.arange(4 * N1, n - 4 * N1, 64), cl["pulses"], replace=False))
    for j, s in enumerate(pos):
        a = np.exp(rng.uniform(np.log(cl["amp"][0]), np.log(cl["amp"][1])))
        z[s - H:s + H] += a * np.exp(1j * rng.uniform(0, 6.28)) * shaped(rng.uniform(-0.5, 0.5))
        if j < cl["pairs"]:                           # an unresolved second pulse a few samples later
            off = int(rng.integers(3, 9))
            z[s - H + off:s + H + off] += 0.7 * a * np.exp(1j * rng.uniform(0, 6.28)) * shaped(rng.uniform(-0.5, 0.5))
    for s in rng.choice(np.arange(4 * N1, n - 4 * N1), cl["rects"], replace=False):
        z[s:s + 3] += cl["amp"][1] * np.exp(1j * rng.uniform(0, 6.28))
    iq[0::2] += z.real
    iq[1::2] += z.imag
    return d, cl, np.clip(np.round(iq), -32767, 32767).astype(np.int16), make_liminfo(d), des
