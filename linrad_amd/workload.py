"""Benchmark / smoke workloads: BASELINE.json configs made concrete, shared by bench.py and smoke()."""
import numpy as np

from .abi import default_config

# algorithmic HBM bytes per input complex sample, per stage (DESIGN.md "Roofline accounting")
ALG_BYTES = {"fft1": 24.0, "sumsq": 16.0, "timf2": 76.0, "blanker": 4.0, "fft2": 64.0,
             # k_fft1w = fft1 + fft1_c's sums + the weak stream of make_timf2 as ONE algorithm: what it has to move per new input sample
             # is the samples in (4 B, each entering two overlapping transforms: 8), the weak stream out (8), its power (4) and the
             # averaged power spectrum (4 B per bin and fft_avg1num = 5 blocks of N1/2 new samples: 1.6) -- the SURVEY 8d figures of the
             # three stages it replaces (24 + 16 + 52) price a spectrum round trip through HBM that no longer exists.
             # Its second pass (timf2s) writes the strong stream once (8); the handful of strong bins it reads is noise.
             "fft1w": 8.0 + 8.0 + 4.0 + 1.6, "timf2s": 8.0,
             # fft2 as ONE pass (fft2_size <= 16384: a transform fits a workgroup's LDS, k_fft2<n>): both timf2 streams in once (8 + 8; the
             # overlapped half of a transform stays in registers), the spectrum out (8 B x 2 transforms per sample at 50 % overlap) and the
             # averaged power (4 B x 2 / waterfall_avgnum = 1).  SURVEY 8d's 64 prices the four-step form: a scratch round trip and the
             # overlapped read taken twice.  (With cfg.fft2_float_sparse only the band mix1 cuts out is stored: the counters show 20.)
             "fft2_single": 8.0 + 8.0 + 16.0 + 1.0,
             # the part of either fft2 figure that is the spectrum ring (8 B x 2 transforms per sample at 50 % overlap): not written with
             # cfg.fft2_float_sparse (only the band mix1 cuts out is), so bench.py prices the stage without it in that mode
             "fft2_spectrum_out": 16.0}
ALG_BYTES_CHAIN = 184.0


def alg_bytes_chain(w):
    """algorithmic bytes per input sample of a bench workload: the 184 B of SURVEY 8d up to mix1, plus the narrowband side when
    it is configured -- timf3 runs at 1/64 of the input rate (mix1 bandwidth reduction 6), fft3 reads and writes it with 50 %
    overlap (8 + 8 bytes x 2), mix2 is a further decimation behind it: +0.5 B per input sample"""
    return ALG_BYTES_CHAIN + (0.5 if w.get("fft3_n") else 0.0)


def workload_name(w, batch):
    """key of a bench workload in profiles/*_traffic.json"""
    return f"n1_{w['fft1_n']}_n2_{w['fft2_n']}_n3_{w.get('fft3_n', 0)}_b{batch}"
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def level_gain(n1, att_n, sigma=64.0, target_pwr=400.0):
    """FIRST_FFT_GAIN putting the weak-signal noise power near the blanker start floor (SURVEY.md 8d level plan)."""
    N1 = 1 << n1
    per_gain = (1.0 / np.sqrt(3.0 / 8.0)) * N1 * 2.0 ** (-att_n) / (150.0 * N1 ** 0.6)
    return max(1, int(round(np.sqrt(target_pwr / (2 * sigma * sigma)) / per_gain)))


def chain_config(fft1_n=14, fft2_n=12, batch=256, device=0, fq_bin=None, fft3_n=0, mix2_n=0, rounds=1):
    """1-channel full chain fft1 -> timf2 -> blank1 -> fft2 -> mix1 (-> fft3 -> mix2 when fft3_n > 0) at BASELINE.json config
    sizes, batched; `rounds` batches may be in flight before the narrowband side is drained (coherent-combine mode)."""
    N1, N2 = 1 << fft1_n, 1 << fft2_n
    M1 = N1 // 2
    samples_per_batch = batch * M1
    k_fft2 = max(4, 2 * (samples_per_batch // (N2 // 2) + 2))
    pow2 = lambda v: 1 << int(np.ceil(np.log2(v)))  # noqa: E731
    # reference time constants at 10 Msps (buf.c:337-346), re-expressed per blanker call (one call per batch)
    avgnum_blocks = int((10e6 + M1 / 2) / M1)
    avgnum = max(2, avgnum_blocks // batch)
    cfg = default_config(
        fft1_n, fft2_n, device=device, fft1_gain=level_gain(fft1_n, 6), bckfft_att_n=6,
        timf1_bytes=pow2(4 * (samples_per_batch + 2 * N1)) * 2, max_fft1n=pow2(2 * batch),
        fft1_sumsq_bufsize=pow2(batch // 5 + 4 + 4) * N1, timf2pow_size=pow2(4 * max(samples_per_batch, 2 * N2)),
        max_fft2n=pow2(k_fft2), waterfall_avgnum=8, wf_xpixels=min(N2, 1024), wf_lines=64,
        timf2_noise_floor_avgnum=avgnum, blanker_info_update_interval=max(1, avgnum // 8),
        blanker_min_points=N2 // 3, mix1_bandwidth_reduction_n=6,
        timf3_size=pow2(4 * k_fft2 * max(8, N2 >> 6) * 2), max_batch=batch)
    if fft3_n:
        N3, Nm, Nm2 = 1 << fft3_n, max(8, N2 >> 6), 1 << mix2_n
        per_round = (samples_per_batch // (N2 // 2) + 2) * (Nm // 2)           # timf3 samples one batch produces
        k_fft3 = per_round // (N3 // 2) + 2
        cfg.fft3_n, cfg.fft3_sinpow, cfg.mix2_n = fft3_n, 2, mix2_n
        cfg.max_fft3n = max(8, pow2(2 * k_fft3))
        cfg.baseband_size = max(4 * Nm2, pow2(2 * max(1, rounds) * cfg.max_fft3n * (Nm2 // 2)))
        cfg.timf3_size = max(cfg.timf3_size, pow2(2 * (2 * max(1, rounds) * per_round + 4 * N3)))
    return cfg


def strong_liminfo(synth, fft1_n, halfwidth=3, min_amp=90.0):
    """Routing table: carriers well above the noise go to the strong path (what the selective limiter would decide)."""
    N1 = 1 << fft1_n
    lim = np.zeros(N1, np.float32)
    for i in range(synth.ncarriers):
        if synth.carrier_amp[i] >= min_amp:
            k = synth.carrier_bin[i] * N1 / synth.fft_size
            c = int(round(N1 // 2 + k))
            lim[max(0, c - halfwidth):min(N1, c + halfwidth + 1)] = 1.0
    return lim
