/*
 * linrad_hip.h -- C ABI of liblinrad_hip.so, the MI355X (gfx950) implementation of
 * Linrad's wideband hot path  fft1 -> timf2(+blank1) -> fft2 -> mix1.
 *
 * Plain C, plain pointers and sizes; no HIP or torch types cross this boundary.
 * Every entry point stands in for one `void f(void)` stage function of the
 * reference (fventuri/linrad) and is called from the same place the reference's
 * clFFT/cuFFT offload hooks in (fft1.c:3519-3553, wcw.c:535-575, 1174-1183).
 * The reference passes state through global ring pointers; here the caller hands
 * the same variables over in `lrh_ptrs` (in/out) and the stage advances exactly
 * the ones the reference function advances (citations on each field).
 *
 * All rings are DEVICE resident (PCIe cannot carry them);
 * `lrh_export` copies a span of any ring back for the GUI / tests.
 *
 * Return value: 0 on success, negative LRH_E* otherwise (the Linrad-side caller
 * maps them to lirerr(146x), convention lxsys.c:494-505).
 */
#ifndef LINRAD_HIP_H
#define LINRAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRH_ABI_VERSION 7      /* 2: lrh_fft1_b takes the worker handle; every entry point may be called from any thread.
                                  3: lrh_config.fft1_float_sparse / fft2_float_sparse (were reserved, 0 = as before); lrh_set_exchange, lrh_spur_acquire,
                                     lrh_set_correlation / lrh_fft1_corr_begin / _finish, lrh_flush, rings LRH_RING_FFT1_CORRSUM .. _SLOWCORR_TOT (additions); lrh_exchange_fn takes the caller's own span;
                                     lrh_sellim.sellim_par1 (the struct grew: struct_size tells a caller built against the older header apart)
                                  4: lrh_spur_permute (addition)
                                  5: lrh_stage_wait, lrh_export_begin / _end (additions); lrh_export waits without holding the context's lock
                                  6: lrh_set_basebraw_fir, lrh_spur_search_config / _get (additions); lrh_ptrs.timf3_py (was reserved[0]); cfg.fft3_sinpow takes every window */

enum {
  LRH_OK = 0,
  LRH_EINVAL = -1,    /* bad argument / unsupported configuration            */
  LRH_ENOMEM = -2,    /* device or host allocation failed                     */
  LRH_EDEVICE = -3,   /* HIP runtime error (no device, launch failure, ...)   */
  LRH_ESTATE = -4,    /* call out of order (e.g. table not set)               */
  LRH_ERANGE = -5,    /* frequency outside mix1 range (lirerr 1211/1212, mix1.c:787-796) */
  LRH_EINTERNAL = -6  /* a C++ exception (allocation failure, ...) was caught at the C boundary: lrh_last_error has its text */
};

/* Sizes and parameters. Names follow the reference globals / genparm[] slots. */
typedef struct lrh_config {
  int struct_size;              /* sizeof(lrh_config), ABI check                                   */
  int device;                   /* HIP device ordinal                                              */
  int rx_rf_channels;           /* ui.rx_rf_channels; 1 per context (channels shard one per GPU)   */
  /* fft1 (fft1.c:413-650, buf.c:193-304) */
  int fft1_n;                   /* log2 fft1_size, 6..15 (buf.c:335: the reference's maximum with the second fft on), 16 with
                                   second_fft_enable = 0 (fft0.c:1162-1169: fft1_permute is unsigned short).  Up to 14 one
                                   workgroup transforms a block in LDS; 15 and 16 take four-step fft1 / timf2 kernels through an
                                   HBM scratch and are restricted to int16 / int32 I/Q input, no sample_shift, sin^2 window (the
                                   limiter kernels keep their table in global memory at 15 and answer LRH_EINVAL at 16)        */
  int fft1_sinpow;              /* genparm[FIRST_FFT_SINPOW] 0..9 (fft0.c:812-921)                 */
  int fft1_gain;                /* genparm[FIRST_FFT_GAIN] (fft1.c:4653-4671)                      */
  int fft1_direction;           /* +1 / -1 (fft1.c:3660-3679)                                      */
  int fft_avg1num;              /* wg.fft_avg1num (fft1.c:4507-4520)                               */
  int fft_avg2num;              /* wg_fft_avg2num (fft1.c:4526-4605)                               */
  int timf1_bytes;              /* input ring size, power of two (buf.c:744-770)                   */
  int max_fft1n;                /* fft1_float ring length in transforms, pow2 >= 8 (buf.c:671-693) */
  int fft1_sumsq_bufsize;       /* floats, pow2 multiple of fft1_size (buf.c:698-737)              */
  int wg_xpoints;               /* wg.xpoints, sets the slow-average recalc stride (fft1.c:4571)   */
  int slowsum_fresh_recalc;     /* fresh_recalc of fft1.c:4546-4567 (2, 4 or 8)                    */
  /* timf2 (timf2.c:31-208, 689-1065) */
  int bckfft_att_n;             /* genparm[FIRST_BCKFFT_ATT_N]                                     */
  int timf2pow_size;            /* samples in the timf2 ring, pow2 (buf.c:371-399)                 */
  /* blank1 (blank1.c:684-1603, buf.c:337-346,418-431,2083-2086, hires_graph.c:1157-1162) */
  int stupid_bln_mode;          /* hg.stupid_bln_mode 0/1                                          */
  float stupid_bln_factor;      /* hg.stupid_bln_factor                                            */
  int blnfit_range;             /* 48 uncalibrated                                                 */
  int blanker_pulsewidth;       /* 0 uncalibrated                                                  */
  int timf2_noise_floor_avgnum;
  int blanker_info_update_interval;
  int blanker_min_points;       /* min_delay_time*ui.rx_ad_speed (blank1.c:712)                    */
  int timf2_noise_floor;        /* start value, 200 (buf.c:418)                                    */
  /* fft2 (fft2.c:86-141, 647-815, 1821-1846) */
  int fft2_n;                   /* log2 fft2_size                                                  */
  int fft2_sinpow;              /* genparm[SECOND_FFT_SINPOW]                                      */
  int max_fft2n;                /* fft2_float ring length in transforms (buf.c:459-471)            */
  int waterfall_avgnum;         /* wg.waterfall_avgnum                                             */
  int wf_first_xpoint;          /* hgwat_first_xpoint                                              */
  int wf_xpixels;               /* wg_xpixels                                                      */
  int wf_mode;                  /* 1: one point per pixel; k>1: hgwat_xpoints_per_pixel=k (max of
                                   group); k<0: hgwat_pixels_per_xpoint=-k (interpolate)           */
  int wf_lines;                 /* lines kept in wg_waterf (wg_waterf_size/wg_xpixels)             */
  /* mix1 (mix1.c:55-262, 781-861, 934-993) */
  int mix1_bandwidth_reduction_n; /* genparm[MIX1_BANDWIDTH_REDUCTION_N]: mix1.n = fft2_n - this   */
  int timf3_size;               /* floats per mix1 channel, pow2 (buf.c:640-663)                   */
  float fftx_points_per_hz;     /* fft2 bins per Hz (seldef.h:219)                                 */
  float mix1_lowest_fq, mix1_highest_fq;
  /* execution */
  int max_batch;                /* largest `batch` any call will pass                              */
  int second_fft_enable;        /* genparm[SECOND_FFT_ENABLE]; 0: fft1 -> fft1_c -> fft1_mix1_fixed, mix1 sized
                                   from fft1 (buf.c:316-332); the reference's own default (uivar.c:371)  */
  int timf2_blockpower_block;   /* floats per S/N-meter block (baseb_graph.c:1330); 0: powersum unused   */
  int timf2_blockpower_size;    /* ring length, pow2 (buf.c:409-412)                                     */
  int timf1_frame_channels;     /* RF channels interleaved in one timf1 frame {I0,Q0,I1,Q1,..} (ui.rx_ad_channels/2,
                                   fft1.c:2052-2055); 0/1: single channel                                 */
  int timf1_channel_index;      /* which of them this context (this GPU) processes                        */
  /* fft3 + mix2 (fft3.c:215-283, mix2.c:83-176): baseband filter / decimator behind mix1; fft3_n = 0 disables */
  int fft3_n;                   /* log2 fft3_size (baseb_graph.c:3332-3376), <= 14                         */
  int fft3_sinpow;              /* genparm[THIRD_FFT_SINPOW]: 0 none, 2 sin^2 (50 % overlap-add, mix2.c:158-176), other windows
                                   through the crossover functions of prepare_mixer (mix2.c:177-216)        */
  int mix2_n;                   /* log2 mix2.size (baseb_graph.c:1302-1323), <= fft3_n                     */
  int max_fft3n;                /* fft3 ring length in transforms (fft3_totsiz/fft3_block), pow2          */
  int baseband_size;            /* baseb_raw ring, complex samples, pow2                                   */
  /* input sample format (fft1.c:413-635) */
  int timf1_dword_input;        /* ui.rx_input_mode & DWORD_INPUT: timf1 holds int32 I,Q (18/24-bit hardware,
                                   expanded .raw recordings) instead of int16                              */
  int sample_shift;             /* ui.sample_shift: Q is taken sample_shift samples after I (fft1.c:470-482) */
  int blanker_channels;         /* 2: this context is one of two coupled RF channels (ui.rx_rf_channels = 2): the blanker
                                   decides on the channel power sum and averages both channels' noise (blank1.c:1017,
                                   1236-1300, 1510-1545, 1570), see lrh_blanker_begin; the fft2 waterfall line comes from
                                   both channels' sums, see lrh_fft2_xy_begin (default wg_waterf_yfac then carries the
                                   rx_rf_channels^2 of wide_graph.c:985).  0/1: single channel                        */
  int timf1_real_input;         /* 1: real samples (ui.rx_input_mode without IQ_DATA, "normal audio" / direct-sampling
                                   hardware): one transform takes 2*fft1_size reals and yields fft1_size bins 0..fs/2, the
                                   reference's default real version fft1_reherm_dit_one (fft1_re.c:32-131, fft1var.c:45,75);
                                   int16 or int32; timf1p_px advances 4*M1 bytes per block and channel like I/Q; two real
                                   channels per frame {a_k, b_k} (fft1_reherm_dit_two, fft1_re.c:133) with
                                   timf1_frame_channels = 2, one per context                                         */
  int fft1_float_sparse;        /* 1: nobody reads the fft1_float ring (second fft on: make_timf2 is its only consumer on the path).  Inside
                                   lrh_wideband_dsp at fft1_size 16384 the forward transform, fft1_c's sums and the weak half of make_timf2
                                   then run as one kernel and only the strong bins of a spectrum reach the ring (the second, sparse pass
                                   of make_timf2 reads them); LRH_RING_FFT1_FLOAT then holds those bins only.  0 (default): every bin is
                                   stored as before (lrh_export, lrh_fft1_mix1_*, the AFC window of the Linrad glue read it)          */
  int fft2_float_sparse;        /* 1: nobody reads whole fft2 spectra: the power sums and waterfall lines are formed inside the transform
                                   kernels and fft2_mix1_fixed cuts mix1.size bins around the selected frequency, so only that band
                                   (+- 64 bins) of every transform is stored and LRH_RING_FFT2_FLOAT / _FFT2_POWER hold that band only.
                                   The band follows lrh_set_mix1_selfreq at the time of each lrh_make_fft2.  Ignored (every bin stored)
                                   while spurs are tracked, with two coupled channels (there it only drops the per-transform cross
                                   products LRH_RING_FFT2_XYPOWER: the group sums and the waterfall line remain), and without a selected frequency.  0 (default):
                                   every bin is stored (lrh_export, lrh_fft2_mix1_afc, NET_RXOUT_FFT2, spur acquisition read it)    */
} lrh_config;

/*
 * The reference's global ring pointers, owned by the caller, passed in/out.
 * A stage advances only the fields the reference function it replaces advances.
 */
typedef struct lrh_ptrs {
  /* advanced by the caller, like wideband_dsp does (wcw.c:1036-1047) */
  int timf1p_px;                /* bytes                                                           */
  int fft1_pa, fft1_na, fft1_nm;
  /* fft1_c (fft1.c:4507-4523) */
  int fft1_nb, fft1_pb;
  int fft1_sumsq_pa, fft1_sumsq_counter, fft1_liminfo_cnt, fft1_sumsq_recalc;
  /* make_timf2 (timf2.c:127-128, 205-207) */
  int fft1_px, fft1_nx;
  int timf2_pa;
  int fft1_lowlevel_points;
  float fft1_lowlevel_fraction;
  /* first_noise_blanker (blank1.c:1458-1476, 1550-1601) */
  int timf2p_fit, timf2_pn2;
  int timf2_cleared_points_unused;  /* device-side; read with lrh_get_blanker_state             */
  int timf2_blanker_points;
  int blanker_info_update_counter;
  /* make_fft2 (fft2.c:672, 1831-1845) */
  int timf2_px;
  int fft2_na, fft2_pa, fft2_nb, fft2_nm;
  int wg_waterf_sum_counter, wg_waterf_ptr, fft2_liminfo_cnt;
  /* fft2_mix1_fixed (mix1.c:991-992); fft1_mix1_fixed advances fft1_nx/fft1_px/timf3_pa (mix1.c:1039-1041) */
  int fft2_nx, timf3_pa;
  /* compute_timf2_powersum (wcw.c:84-137) */
  int timf2_pb, timf2_blockpower_pa;
  /* make_fft3_all (fft3.c:784,797) and fft3_mix2 (mix2.c:1079,2057-2059) */
  int timf3_px, fft3_pa, fft3_px, baseb_pa;
  int timf3_py;                 /* fft3_mix2 with bg.mixer_mode = 2: where the newest fft3 transform's samples start in timf3 (mix2.c:229, 2060) */
  int reserved[5];
} lrh_ptrs;

/* device-resident blanker scalars (blnkdef.h:18-33, globdef.h:983) read back on demand */
typedef struct lrh_blanker_state {
  int timf2_noise_floor;            /* signed int, truncating recursion (blank1.c:1583)            */
  unsigned int stupid_bln_limit;
  float timf2_despiked_pwr[2];
  float timf2_despiked_pwrinc[2];
  float stupid_blanker_rate;
  int timf2_cleared_points;         /* since the last info update                                  */
  int last_call_cleared;            /* cleared_points of the most recent call                      */
  int slow_path_calls;              /* calls that fell back to the exact serial scan               */
  /* linear ("clever") blanker, lrh_set_blanker_tables */
  unsigned int clever_bln_limit;
  float clever_blanker_rate;
  int timf2_fitted_pulses;          /* since the last info update                                  */
  int last_call_fitted;             /* fitted_pulses of the most recent call                       */
  int last_call_rejected;           /* pulses of that call flagged 65 (bad_pulse, blank1.c:945)    */
  int clever_serial_calls;          /* calls whose region-parallel replay was discarded for the one-wave replay */
} lrh_blanker_state;

/* ---- linear ("clever") noise blanker: first_noise_blanker's pulse search / fit / subtract part (blank1.c:765-1003 with
   subtract_onechan_pulse :36-232 and set_flag :615-682).  It needs the pulse response of the calibrated receiver, which
   init_blanker (buf.c:1786-2057) derives from the amplitude calibration fft1_desired at start-up: those tables are an INPUT
   here, handed over once (the calibration itself -- cal*.c -- is outside this path).  Two coupled channels
   (cfg.blanker_channels = 2): install the same tables on both contexts; the fit is then the two-channel one (get_pulse_pol,
   transform_timf2_pol, subtract_twochan_pulse, blank1.c:232-609) and takes one more exchange, see lrh_blanker_begin.
   With tables installed lrh_first_noise_blanker first runs the pulse search over the span, then the stupid blanker
   (cfg.stupid_bln_mode) as before; timf2p_fit follows blank1.c:1458-1461 ((pf-16) & ~3: a pulse too close to the end of the
   span is left for the next call), which makes that pointer data dependent: the call reads one int back, so
   with clever mode on each lrh_first_noise_blanker waits for its own device work and lrh_wideband_dsp runs the serial schedule.
   On the device the span is cut into regions at quiet stretches of the candidate samples and the regions are replayed in parallel,
   each exactly in the reference's order; if two regions' reach ever overlaps the span is restored and replayed by one wave
   (lrh_blanker_state.clever_serial_calls counts those calls). */
#define LRH_BLN_INFO_SIZE 7       /* blnkdef.h:5  */
#define LRH_MAX_REFPULSES 256     /* blnkdef.h:6  */
typedef struct lrh_bln_info { int size; float rest; float avgmax; } lrh_bln_info;   /* BLANKER_CONTROL_INFO, blnkdef.h:8-14 */
typedef struct lrh_blanker_tables {
  int clever_bln_mode;              /* hg.clever_bln_mode: 1 = limit follows the noise floor, 2 = fixed clever_bln_limit */
  float clever_bln_factor;          /* hg.clever_bln_factor (mode 1)                                                    */
  unsigned int clever_bln_limit;    /* hg.clever_bln_limit at start                                                     */
  int refpul_size;                  /* power of two, 4..256                                                             */
  int largest_blnfit;               /* 0..LRH_BLN_INFO_SIZE-1; cfg.blnfit_range must be bln[largest_blnfit].size/2 + cfg.blanker_pulsewidth */
  float liminfo_amplitude_factor;
  lrh_bln_info bln[LRH_BLN_INFO_SIZE];
  const float *refpulse;            /* blanker_refpulse  [LRH_MAX_REFPULSES][refpul_size][2]                            */
  const float *phasefunc;           /* blanker_phasefunc [refpul_size][2]                                               */
  const int *pulindex;              /* blanker_pulindex  [LRH_MAX_REFPULSES]                                            */
} lrh_blanker_tables;

/* mix1 phase bookkeeping (seldef.h:205-214), host side, double-checked against mix1.c:781-861 */
typedef struct lrh_mix1_state {
  double mix1_selfreq;              /* < 0: channel not selected (mix1_clear)                      */
  int mix1_point, mix1_old_point;
  float mix1_phase, mix1_phase_step, mix1_phase_rot, mix1_old_phase;
} lrh_mix1_state;

typedef enum lrh_ring {
  LRH_RING_TIMF1 = 0,           /* int16                                                           */
  LRH_RING_FFT1_FLOAT,          /* float [max_fft1n][N1][2]                                        */
  LRH_RING_FFT1_SUMSQ,          /* float [fft1_sumsq_bufsize]                                      */
  LRH_RING_FFT1_SLOWSUM,        /* float [N1]                                                      */
  LRH_RING_TIMF2_FLOAT,         /* float [timf2pow_size][4] {wRe,wIm,sRe,sIm}; offset/count multiples of 4 */
  LRH_RING_TIMF2_PWR,           /* float [timf2pow_size]                                           */
  LRH_RING_FFT2_FLOAT,          /* float [max_fft2n][N2][2]                                        */
  LRH_RING_FFT2_POWER,          /* float [max_fft2n][N2]                                           */
  LRH_RING_FFT2_POWERSUM,       /* float [N2]                                                      */
  LRH_RING_WG_WATERF,           /* int16 [wf_lines][wf_xpixels]                                    */
  LRH_RING_TIMF3_FLOAT,         /* float [timf3_size]                                              */
  LRH_RING_TIMF2_BLOCKPOWER,    /* float [timf2_blockpower_size]                                   */
  LRH_RING_FFT3,                /* float [max_fft3n][fft3_size][2]                                 */
  LRH_RING_BASEB_RAW,           /* float [baseband_size][2]                                        */
  LRH_RING_FFT2_XYPOWER,        /* float [max_fft2n][N2][4] = TWOCHAN_POWER {x2,y2,im_xy,re_xy} (globdef.h:1371-1376);
                                   two coupled channels only, filled by lrh_fft2_xy_finish                */
  LRH_RING_FFT2_XYSUM,          /* float [N2][4]: fft2_xysum, the running sum of the current waterfall group */
  LRH_RING_FFT1_CORRSUM,        /* float [fft1_sumsq_bufsize][2]: fft1_corrsum (correlation spectrum, lrh_set_correlation)   */
  LRH_RING_FFT1_SLOWCORR,       /* float [N1][2]: fft1_slowcorr                                                               */
  LRH_RING_FFT1_SLOWCORR_TOT,   /* double [N1][2]: fft1_slowcorr_tot (count in elements of 8 bytes)                           */
  LRH_RING_COUNT
} lrh_ring;

typedef struct lrh_ctx lrh_ctx;

/* ---- lifetime: replaces cufftPlanMany/cudaMalloc at wcw.c:553-575 and the never-freed
        buffers / destroy_clFFT_plan at wcw.c:1174-1183 ---- */
int  lrh_abi_version(void);
/* sizeof of the public structs as this library was compiled, for bindings that restate them (ctypes, cgo ...): index = the order of
   the enum; 0 for an unknown index */
enum { LRH_SZ_CONFIG = 0, LRH_SZ_PTRS, LRH_SZ_BLANKER_STATE, LRH_SZ_BLANKER_TABLES, LRH_SZ_MIX1_STATE, LRH_SZ_SELLIM, LRH_SZ_SPUR, LRH_SZ_AFC, LRH_SZ_SYNTH };
size_t lrh_sizeof(int which);
int  lrh_config_defaults(lrh_config *cfg, int fft1_n, int fft2_n);     /* fills reference defaults (uivar.c:371) */
int  lrh_open(const lrh_config *cfg, lrh_ctx **out);
void lrh_close(lrh_ctx *ctx);
const char *lrh_last_error(const lrh_ctx *ctx);
int  lrh_get_derived(const lrh_ctx *ctx, int *fft1_interleave_points, int *fft2_interleave_points,
                     int *mix1_size, int *mix1_interleave_points, int *timf3_block);
void lrh_ptrs_init(const lrh_ctx *ctx, lrh_ptrs *p);                    /* zero start state (buf.c:400-408) */

/* ---- tables the control plane owns ---- */
int lrh_set_filtercorr(lrh_ctx *ctx, const float *fft1_filtercorr /* 2*N1, NULL: uncalibrated default
                                                                     of clear_fft1_filtercorr, fft1.c:4673-4724 */);
int lrh_set_liminfo(lrh_ctx *ctx, const float *liminfo /* N1 floats, 0 = weak (timf2.c:50) */);
/* I/Q mirror-image calibration, fft1_calibrate_flag & CALIQ (fft1.c:3598-3658): fft1_foldcorr, N1 complex floats in
   the transform's bin order; every bin is orthogonalised against its mirror image before the filter correction.
   NULL switches it off (the default: uncalibrated). */
int lrh_set_foldcorr(lrh_ctx *ctx, const float *fft1_foldcorr);
/* Phasing of the second RF channel, pg_ch2_c1 / pg_ch2_c2 (pol_graph.c:160-170): fft1_b ends by turning every bin of
   channel 2 by (c1 - j c2) (fft1.c:4064-4080).  Channels are sharded one per context here, so the context that carries
   the second channel is given the pair; (1, 0) = off (the default).  The constant is folded into the filter correction
   that follows it in fft1_c. */
int lrh_set_ch2_phasing(lrh_ctx *ctx, float c1, float c2);
int lrh_set_waterfall_yfac(lrh_ctx *ctx, const float *wg_waterf_yfac /* N1 floats, NULL: make_wg_yfac default */);
int lrh_get_table(lrh_ctx *ctx, const char *name, float *dst, int count); /* "fft1_window","fft2_window",
                                                                     "mix1_fqwin","fft1_filtercorr","wg_waterf_yfac" */

/* ---- selective limiter on the device-resident spectra (sellim.c:738-1157 fft1_update_liminfo + sellim.c:38-157
   selfreq_liminfo).  The reference runs it in wideband_dsp whenever fft1_c has completed an averaging period
   (fft1_liminfo_cnt changed, wcw.c:1124-1128); it reads fft1_sumsq at fft1_sumsq_pa and fft1_slowsum, keeps liminfo[],
   old_liminfo[] and the per-bin hold-off counters liminfo_wait[], and make_timf2 routes with the result.  Here all of that
   stays on the device: one call = one run of the reference function; the routing words k_timf2 consumes are rebuilt on the
   device in stream order, so the next lrh_make_timf2 routes with the new table and no spectrum or table crosses PCIe.
   Host-visible by-product: the count of weak bins (fft1_lowlevel_points, timf2.c:37-52), read back asynchronously -- with
   `exact_stats` the call waits for it (the reference's value at once); without, the count of the newest update whose readback has
   arrived is installed whenever lrh_make_timf2 or the next update looks (lrh_sync installs the newest), i.e. the statistic (not
   the routing) lags by however far the host runs ahead of the device, and the host never waits for work it has only just enqueued.
   lrh_set_liminfo and this call may be mixed: both replace the table in force.
   The reference looks right after fft1_c has closed an averaging period (p->fft1_sumsq_counter == 0): the slot at fft1_sumsq_pa then
   holds the sums of a ring lap ago, and those are what it reads.  A call in the middle of a period (p->fft1_sumsq_counter != 0:
   batched rounds whose length is no multiple of fft_avg1num) would find the unfinished sums of the period in progress in that slot;
   it reads the newest finished period (one slot back) instead. */
typedef struct lrh_sellim {
  int struct_size;
  int sellim_maxlevel;          /* genparm[SELLIM_MAXLEVEL] (uivar.c:371: 12000)                               */
  int spek_avgnum;              /* wg.spek_avgnum                                                              */
  float fft1_blocktime;         /* seconds between fft1 transforms: sets the one-second hold-off (sellim.c:771) */
  float blanker_ston_fft1;      /* hg.blanker_ston_fft1 (hires_graph.c:710)                                    */
  int sellim_par2, sellim_par3, sellim_par4, sellim_par5, sellim_par6, sellim_par7, sellim_par8;   /* hg.sellim_par* (hires_graph.c:1175-1189) */
  int liminfo_group_points;     /* buf.c:816-824                                                               */
  int fft1_first_point, fft1_last_point, fft1_first_inband, fft1_last_inband;   /* set_fft1_endpoints, fft1.c:4607-4650 */
  int baseband_bw_fftxpts;      /* baseb_graph.c:1165 (an int in the reference, uidef.h:157)                   */
  int ston_scale;               /* mg.scale_type == MG_SCALE_STON                                              */
  int exact_stats;              /* 1: wait for the weak-bin count (see above)                                  */
  /* lrh_fft2_update_liminfo only */
  float blanker_ston_fft2;      /* hg.blanker_ston_fft2 (hires_graph.c:722)                                    */
  float fft2_blocktime;         /* seconds between fft2 transforms (buf.c:456)                                 */
  int sellim_par1;              /* hg.sellim_par1: 2 (hires_graph.c:1175, the reference's setting), 1 or 0     */
  /* selfreq_liminfo's liminfo_amplitude_factor (sellim.c:119-155): the calibrated form needs the amplitude calibration */
  const float *fft1_desired;    /* N1 floats, or NULL: uncalibrated form (share of strong bins in the passband) */
} lrh_sellim;
int lrh_fft1_update_liminfo(lrh_ctx *ctx, lrh_ptrs *p, const lrh_sellim *par);
/* fft2_update_liminfo (sellim.c:159-736; hg.sellim_par1 = 2 is the reference's setting and what is described here): after
   make_fft2 has completed a waterfall line (fft2_liminfo_cnt, wcw.c:1129-1133) the summed fft2 power spectrum -- averaged over the
   fft2 bins of every fft1 bin -- gives a second, finer look: group minima -> global noise floor -> bins above
   0.5 * blanker_ston_fft2 * floor join the strong signals for the hold-off time, and a table that has grown beyond a quarter of
   the passband is thinned.  Ends with selfreq_liminfo like the fft1 variant.  Same table, same hand-over to make_timf2.
   sellim_par1 = 0 (sellim.c:170-281): the median of all fft2 bin powers is the noise floor; the band edges below 2 % of it and
   every fft1 bin holding an fft2 bin above blanker_ston_fft2 * median join the strong signals.  sellim_par1 = 1 (sellim.c:283-533):
   between attenuated carriers every weak-signal region of six bins or more gets a noise floor of its own (mean of the bins within
   2 (1 + 2 / waterfall_avgnum) of its in-band minimum); bins above blanker_ston_fft2 * floor join, then whole regions and single bins
   above the same factor times the length-weighted mean floor of all regions. */
int lrh_fft2_update_liminfo(lrh_ctx *ctx, lrh_ptrs *p, const lrh_sellim *par);
/* The limiter calls of wideband_dsp's loop (wcw.c:1124-1133) inside lrh_wideband_dsp: with parameters installed here every round
   of that call ends with fft1_update_liminfo when fft1_c has completed an averaging period since the last look (fft1_liminfo_cnt),
   and, with fft2_too, fft2_update_liminfo when make_fft2 has completed a waterfall line (fft2_liminfo_cnt) -- the reference's own
   cadence in batched operation: once per pass of the loop, however many periods the pass covered.  par = NULL: off (the caller
   makes the calls itself).  fft2_too keeps lrh_wideband_dsp off its one-round-late schedule (the fft2 of a round must have run). */
int lrh_wideband_limiter(lrh_ctx *ctx, const lrh_sellim *par, int fft2_too);
/* liminfo_amplitude_factor as selfreq_liminfo left it (the linear blanker scales its reference pulse with it, blank1.c:143-144);
   the limiter calls above keep it current on the device; a host that supplies tables itself (lrh_set_liminfo) sets it here */
int lrh_get_liminfo_amplitude_factor(lrh_ctx *ctx, float *factor);   /* synchronous */
int lrh_set_liminfo_amplitude_factor(lrh_ctx *ctx, float factor);
int lrh_get_liminfo(lrh_ctx *ctx, float *liminfo /* N1 floats: the table in force */);     /* synchronous */

/* ---- spur (carrier) subtraction in the fft2 spectra: eliminate_spurs (spur.c:36-494), called by make_fft2 between the transform
   and the power sums (fft2.c:647-652) when spurs are being tracked.  Every spur has a phase-locked loop over the last spur_speknum
   transforms of its SPUR_WIDTH = 7 bins (refine_pll_parameters spur.c:634-680, spur_phase_parameters spur.c:1427-1652): the new
   transform's bins join the history, the loop's phase / frequency / drift / amplitude are refined, and amplitude x the reference
   line shape spur_spectra at the predicted phase is subtracted from the new transform -- before mix1, the power sums and the
   waterfall see it.  Built: this tracking / subtraction path, one RF channel, float spectra, on the device inside lrh_make_fft2.
   Control plane, stays with the host like the AFC: finding spurs and the first lock (init_spur_elimination, store_new_spur,
   spur_phase_lock, spursub.c) hand their result over with lrh_spur_set; a spur whose lock is lost (flag != 0, spur.c:238,
   293-297) is kept in the reference's unlocked bookkeeping (history copied, flag counted up) and reported by lrh_spur_get
   for the host to re-lock (spur_relock, spur.c:682) or drop. */
#define LRH_SPUR_WIDTH 7            /* SPUR_WIDTH = SPUR_SIZE - 1 (globdef.h:173-174, seldef.h:6) */
#define LRH_SPUR_SPECTRA (256 * 8)  /* NO_OF_SPUR_SPECTRA * SPUR_SIZE floats (globdef.h:175)      */
typedef struct lrh_spur {
  int spur_location;            /* first of the SPUR_WIDTH fft2 bins                              */
  int spur_flag;                /* 0 locked; > 0 unlocked since that many transforms              */
  float spur_freq;              /* bins, with decimals                                            */
  float spur_d0pha, spur_d1pha, spur_d2pha;   /* PLL phase and its first two differences per transform */
  float spur_ampl, spur_noise, spur_avgd2;
} lrh_spur;
/* spur_speknum (buf.c:1141: 0.1 genparm[SPUR_TIMECONSTANT] / fftx_blocktime, >= 4, 4 spur_speknum <= max_fft2n) and the table of
   reference line shapes (init_spur_spectra, spursub.c:824-940: depends only on the fft2 window); the derived constants
   (sp_numsub, sp_avgnum, spur_max_d2, spur_minston, spur_weiold / weinew, spur_linefit, buf.c:1149-1170; spur_freq_factor,
   buf.c:480) follow from it.  max_spurs = genparm[MAX_NO_OF_SPURS]; 0 switches the feature off. */
int lrh_spur_config(lrh_ctx *ctx, int max_spurs, int spur_speknum, const float *spur_spectra);
/* the control plane's hand-over: n spurs with their loop state and, per spur, the history the loop works on -- spur_table
   [max_fft2n][SPUR_WIDTH][2] (the bins of past transforms), spur_signal [max_fft2n][2], spur_ind [max_fft2n] (buf.c:1114-1131) */
int lrh_spur_set(lrh_ctx *ctx, int n, const lrh_spur *spurs, const float *spur_table, const float *spur_signal, const int *spur_ind);
int lrh_spur_get(lrh_ctx *ctx, int max, lrh_spur *spurs, int *n);          /* synchronous */
/* The control plane drops spurs or reorders its list (remove_spur spur.c:596-631: the last spur takes the place of a dropped one;
   swap_spurs spursub.c:755: the list is kept in order of frequency, init_spur_elimination spursub.c:315-343): afterwards the device tracks
   n spurs, number i being what was number src[i] (loop state and history move along).  n <= the current count.  Synchronous. */
int lrh_spur_permute(lrh_ctx *ctx, int n, const int *src);
/* The search for NEW spurs (spur_removal -> init_spur_elimination, wcw.c:204-247, spursub.c:181): make_fft2 sums the power rows of
   3 spur_speknum transforms into spursearch_powersum and, with the next row, forms spursearch_spectrum and cleans it
   (fft2.c:673-699; spursearch_spectrum_cleanup, spursub.c:40-175: noise floor off, every peak that does not look like a spur's line
   shape wiped).  With a range configured lrh_make_fft2 does that on the device for every transform it makes -- behind the spur
   subtraction, like the reference -- on a stream of its own; the control plane fetches the finished search spectrum
   (spectrum[0 .. last - first], may be NULL), spur_search_threshold, the number of search spectra completed so far and
   spursearch_sum_counter, and walks it as before (autospur_point, spursub.c:268-290).  (0, 0) switches the search off.
   Needs lrh_spur_config and cfg.fft2_float_sparse = 0.  The getter is synchronous. */
int lrh_spur_search_config(lrh_ctx *ctx, int spur_search_first_point, int spur_search_last_point);
int lrh_spur_search_get(lrh_ctx *ctx, float *spursearch_spectrum, float *spur_search_threshold, int *completed, int *spursearch_sum_counter);
/* Acquisition on the device-resident spectra: store_new_spur (spursub.c:619-751: the seven bins from `pnt` of the
   newest spur_speknum transforms join the history, the summed power gives the frequency with decimals) and spur_phase_lock
   (spursub.c:1247-1426 with verify_spur_pll :1428-1843: up to five rounds of the loop on that history, accepted when the corrections
   have died down and the residual over the window is spectrally flat) for the next free spur number.  p->fft2_na = ffts_na, the ring
   position behind the newest transform (wcw.c:288-289).  *locked = 1: the spur is tracked from the next lrh_make_fft2 on
   (no_of_spurs++, spursub.c:309) and the carrier has been taken out of the spur_speknum transforms the loop was closed on as well
   (initial_remove_spur, spursub.c:346-470, the call that follows the lock in init_spur_elimination); 0: no lock, nothing changed.
   The candidates come from the search spectrum (lrh_spur_search_get above).  Needs cfg.fft2_float_sparse = 0.  Synchronous. */
int lrh_spur_acquire(lrh_ctx *ctx, const lrh_ptrs *p, int pnt, int *locked);

/* ---- producer side: what finish_rx_read (rxin.c:1143-1436) makes visible in timf1 ---- */
int lrh_timf1_write(lrh_ctx *ctx, const void *src, int byte_offset, int nbytes);   /* host -> device ring, wraps */
void *lrh_timf1_device_ptr(lrh_ctx *ctx);                                         /* for device-resident producers */
/* Streaming producer: the same copy without the host wait.  It runs on the context's own copy stream, behind the fft1
   launches already enqueued (they may still read the ring) and ahead of every later lrh_fft1_b / lrh_wideband_dsp call
   (which wait for it on the device), so the PCIe transfer of the next samples overlaps the chain working on the previous
   ones.  `src` must stay untouched until lrh_timf1_write_wait returns (or the next lrh_sync); page-locked memory --
   lrh_host_register on Linrad's timf1 arena once after get_buffers, lrh_host_unregister before free_buffers (buf.c:2105;
   the shim never frees host memory) -- makes the copy a true DMA.  One producer style per context: the
   synchronous lrh_timf1_write travels on the main stream and is not ordered against copies still queued here.
   Round 6: a source that does NOT lie in a span registered with lrh_host_register is copied through a page-locked staging buffer of
   the library and the call returns when the samples are on the device (same order on the copy stream): the library never lets the
   HIP runtime pin pageable caller memory on the fly -- neither here nor in any table upload or read-back.
   Calls out of a registered span are NOTED and issued merged: by the producer's own call once 256 kB have gathered (LRH_IN_MERGE_KB; 0: every
   call copies at once), by the next reader of timf1 for what it reads, by lrh_timf1_write_wait / lrh_sync / lrh_host_unregister for all of it
   -- a receiver that hands over one fft1 block per call would otherwise make a hipMemcpyAsync per 32 kB.  A reader waits for the issue that
   carries ITS samples, not for everything a producer running ahead has queued (profiles/r06_glue_trace.txt).  The contract is the one above:
   `src` untouched until lrh_timf1_write_wait / lrh_sync -- Linrad's timf1 ring, which keeps a block for a whole lap, does that by itself. */
int lrh_timf1_write_async(lrh_ctx *ctx, const void *src, int byte_offset, int nbytes);
int lrh_timf1_write_wait(lrh_ctx *ctx);
int lrh_host_register(lrh_ctx *ctx, void *ptr, size_t bytes);
int lrh_host_unregister(lrh_ctx *ctx, void *ptr);
/* One read of an 18-bit .raw recording (rx_file_input, rxin.c:1643-1644): `packed_bytes` (multiple of 9) of packed
   data, 9 bytes per four int32 components, are expanded like expand_rawdat (csplit.c:20-73: value left-justified in
   32 bits, bit 13 set for the truncated half LSB) into the timf1 ring at byte_offset (multiple of 16, wraps).
   Needs timf1_dword_input = 1.  The expansion runs on the device; only the packed bytes cross PCIe. */
int lrh_timf1_write_packed18(lrh_ctx *ctx, const void *src, int byte_offset, int packed_bytes);

/* ---- threading ----
   Linrad calls the stage functions from its stage threads: wideband_dsp (or up to six fft1_b workers, wcw.c:476-500),
   timf2_routine (fft1_c, make_timf2, first_noise_blanker; wcw.c:401-441), second_fft (make_fft2; wcw.c:250-304) and
   narrowband_dsp (fft2_mix1_*; wcw.c:1240-1405), ordered by its events EVENT_TIMF1 / TIMF2 / FFT2 / FFT1_READY
   (thrdef.h:136-170).  Every entry point of this library may be called from any host thread at any time: a context
   serialises its callers while a call does its pointer bookkeeping and enqueues its device work (a batched lrh_wideband_dsp
   that has run several rounds ahead of the device sleeps inside that lock until a staging slot is free -- milliseconds), and
   the device executes the work in the order of the calls.  The producer side -- lrh_timf1_write_async / lrh_timf1_write_wait --
   has a lock of its own and is never held up by a stage call: an input thread hands its block over at once.  So the reference's event order is all a caller has to
   keep, exactly as for the CPU functions; a stage call has "completed" for that purpose when it returns (its writes
   are stream-ordered ahead of everything enqueued later).  One lrh_ptrs may be shared by the threads like the
   reference's globals are -- each stage advances only its own fields -- or each thread may keep the fields it owns.
   examples/lrh_threads.c drives a context this way. */

/* ---- stages ---- */
/* fft1_b (fft1def.h:363; fft1.c:3302, mode 7 semantics fft1.c:413-447 + fft0.c:161 + fft1.c:637) for `batch`
   consecutive blocks; block i reads at timf1p_ref + i*timf1_blockbytes and writes transform
   (fft1_pa/fft1_block + i) & fft1n_mask. The filter correction of fft1_c (fft1.c:4119-4127) is applied
   in the store epilogue. The caller advances timf1p_px / fft1_pa / fft1_na as wcw.c:1036-1047.
   `handle` is the reference's gpu_handle_number (fft1.c:3302, wcw.c:500): 0 for the caller's own thread
   (no_of_fft1b == 0, wcw.c:1036), 1..6 for the workers THREAD_FFT1B1..6 (MAX_FFT1_THREADS, thrdef.h:107).  A worker handle
   has its own HIP stream, so transforms handed to different workers overlap on the device like the reference's
   workers overlap on CPU cores; the next reader of fft1_float (lrh_fft1_c, lrh_make_timf2, lrh_fft1_mix1_*, lrh_export)
   waits for them on the device.  The dispatcher retires workers in order before it advances fft1_pa
   (wcw.c:1005-1032); here a worker call returns as soon as its launch is enqueued.
   timf1 must keep the ONE block behind timf1p_ref until the call's kernels have run (lrh_timf1_write_async orders itself behind
   them): the fused fft1 + timf2 kernel rebuilds the overlap partner of the call's first transform from there instead of carrying
   it in memory.  When a handle-0 call does not start where the previous one stopped, or the filter tables changed in between,
   the ring does not hold that partner: with cfg.fft1_float_sparse = 0 the call runs as fft1 kernel + timf2 kernel (partner = the
   previous spectrum in fft1_float, which is the reference's carry, timf2.c:1018-1025); with the sparse ring the overlap of that one
   transform starts over as at the start of a stream. */
int lrh_fft1_b(lrh_ctx *ctx, int handle, int timf1p_ref, int fft1_pa, int batch);
/* fft1_c (fft1def.h:364; fft1.c:4085-4524): power accumulation into fft1_sumsq, slow average
   (update_fft1_slowsum fft1.c:4526-4605 + new_fft1_averages wide_graph.c:1003-1052), fft1_nb/pb advance. */
int lrh_fft1_c(lrh_ctx *ctx, lrh_ptrs *p, int batch);
/* make_timf2 (fft1def.h:349; timf2.c:31-208, 689-1065) */
int lrh_make_timf2(lrh_ctx *ctx, lrh_ptrs *p, int batch);
/* first_noise_blanker (fft2def.h:64; blank1.c:684-1603), stupid blanker + noise statistics */
/* ---- two coupled RF channels, one per context (cfg.blanker_channels = 2, cfg.timf1_channel_index = the channel) ----
   The reference keeps both channels in one array: timf2_pwr_float holds |w0|^2+|w1|^2, the stupid blanker compares that sum
   with the limit, clears both channels' samples, and the noise floor is the mean of the two channels' despiked powers.
   With one channel per GPU these are two sum exchanges per call, which the caller performs (RCCL all-reduce in the
   product, plain adds in tests):
     lrh_blanker_begin        gathers the own power of the samples the coming call will scan into exchange buffer
                              LRH_X_PWR (count floats; 0: the call will return early on the rate limit)
     -> all-reduce(sum) LRH_X_PWR[0..count) over the two contexts
     lrh_first_noise_blanker  scans the summed power, clears, leaves the own channel's every-4th-sample mean power
                              (blank1.c:1512-1541) in slot timf1_channel_index of LRH_X_STAT (2 floats, other slot 0)
     -> all-reduce(sum) LRH_X_STAT[0..2)
     lrh_blanker_finish       the statistics / threshold update of blank1.c:1542-1601 with both channels' values
   With the linear blanker's tables installed (lrh_set_blanker_tables on both contexts) the pulse fit works on both channels at
   once (get_pulse_pol, transform_timf2_pol, subtract_twochan_pulse, blank1.c:232-609, 984-992): it finds the polarisation of a
   pulse from the two channels' samples around its peak and takes the fitted pulse out of both.  Both contexts then run the same
   search on the same data and keep their own channel's result, which takes one more exchange before the scan:
     lrh_blanker_begin        also copies the own weak samples from blnfit_range before the span to blnfit_range behind it into
                              slot timf1_channel_index of LRH_X_WEAK (float [2][n][2]); `count` of LRH_X_PWR is then the span plus
                              blnfit_range (the search reads the summed power that far ahead)
     lrh_blanker_weak_span    n*2 = floats per slot of LRH_X_WEAK for this call (0: nothing to gather)
     -> all-gather of the two slots of LRH_X_WEAK (beside the all-reduce of LRH_X_PWR)  */
enum { LRH_X_PWR = 0, LRH_X_STAT = 1, LRH_X_BINS = 2, LRH_X_POL = 3, LRH_X_WEAK = 4, LRH_X_SPEC = 5 };
int lrh_blanker_begin(lrh_ctx *ctx, const lrh_ptrs *p, int *count);
int lrh_blanker_weak_span(lrh_ctx *ctx, size_t *count);
int lrh_blanker_finish(lrh_ctx *ctx, lrh_ptrs *p);
/* Cross products of the two channels' fft2 spectra (make_fft2's two-channel branch, fft2.c:1622-1640: per transform and bin
   TWOCHAN_POWER {x2 = |X|^2, y2 = |Y|^2, im_xy = Xim*Yre - Xre*Yim, re_xy = Xre*Yre + Xim*Yim},
   summed over wg.waterfall_avgnum transforms in fft2_xysum) and the two-channel waterfall line, which shows the
   polarisation-independent power  (x2+y2) + 2 (re_xy^2 + im_xy^2 - x2 y2) / (x2+y2)  of the sums (fft2.c:1700-1815).
   A cross product needs both channels' bins, so with one channel per GPU this is an all-gather:
     at = *p;  lrh_make_fft2(ctx, p, batch);        each context transforms its own channel (no waterfall line of its own
                                                     when cfg.blanker_channels = 2; the pointers advance as usual)
     lrh_fft2_xy_begin(ctx, &at, batch, &count)     copies the `batch` new transforms (fft2_float from at.fft2_na on) into
                                                     slot cfg.timf1_channel_index of LRH_X_BINS: float [2][count],
                                                     count = batch * 2 * N2
     -> all-gather of the two slots (each context receives the other channel's slot)
     lrh_fft2_xy_finish(ctx, &at, batch)            fills LRH_RING_FFT2_XYPOWER / _XYSUM and writes the waterfall lines that
                                                     complete within the batch at at.wg_waterf_ptr downwards
   Both contexts end with the same rings (like an all-reduce result); calling finish on one of them is enough for a GUI. */
int lrh_fft2_xy_begin(lrh_ctx *ctx, const lrh_ptrs *at, int batch, size_t *count);
int lrh_fft2_xy_finish(lrh_ctx *ctx, const lrh_ptrs *at, int batch);
int lrh_exchange_ptr(lrh_ctx *ctx, int which, void **device_ptr);           /* for collectives on lrh_stream(ctx) */
int lrh_exchange_read(lrh_ctx *ctx, int which, float *dst, size_t off, size_t count);   /* synchronous, for tests / host exchange */
int lrh_exchange_write(lrh_ctx *ctx, int which, const float *src, size_t off, size_t count);
/* Correlation spectrum of the two channels (genparm[FFT1_CORRELATION_SPECTRUM] = 1, buf.c:1223-1233): beside fft1_sumsq, fft1_c forms
   per bin fft1_corrsum = sum over the averaging period of 2 X conj(Y) (fft1.c:4146-4150, 4189-4193: re = 2 (Xre Yre + Xim Yim),
   im = 2 (Xim Yre - Xre Yim), X = channel 0), and update_fft1_slowsum keeps its sliding sum over wg_fft_avg2num periods in
   fft1_slowcorr (refreshed from scratch like fft1_slowsum, wide_graph.c:1033-1050, no floor) and the sum of everything since the last
   reset in fft1_slowcorr_tot (double; slowcorr_tot_avgnum transforms; fft1.c:4584-4603).  A cross spectrum needs both channels' bins: with
   one channel per context this is an all-gather of the new transforms, 8 bytes per bin and transform -- an opt-in for installations
   that display the correlation spectrum:
     lrh_set_correlation(ctx, 1)                        once (0: off, rings freed; also clears like clear_fft1_correlation, fft1.c:5386)
     at = *p;  lrh_fft1_c(ctx, p, batch);               each context's own sums as always
     lrh_fft1_corr_begin(ctx, &at, batch, &count)       the batch's transforms (ring slots from at.fft1_nb) into slot
                                                        cfg.timf1_channel_index of LRH_X_SPEC: float [2][count], count = batch * 2 * N1
     -> all-gather of the two slots
     lrh_fft1_corr_finish(ctx, &at, batch)              fft1_corrsum of the periods the batch touches, fft1_slowcorr / _tot after every
                                                        completed period, in the order of update_fft1_slowsum
   Both contexts end with the same rings.  lrh_wideband_dsp makes these calls itself on coupled contexts when the mode is on.
   (fft1_correlation_flag >= 2, the correlation receiver with its double-precision mixer, is not built.) */
int lrh_set_correlation(lrh_ctx *ctx, int on);
int lrh_fft1_corr_begin(lrh_ctx *ctx, const lrh_ptrs *at, int batch, size_t *count);
int lrh_fft1_corr_finish(lrh_ctx *ctx, const lrh_ptrs *at, int batch);
int lrh_get_slowcorr_tot_avgnum(lrh_ctx *ctx, int *n);
int lrh_first_noise_blanker(lrh_ctx *ctx, lrh_ptrs *p);
/* install / remove the linear blanker's tables (see lrh_blanker_tables); the arrays are copied */
int lrh_set_blanker_tables(lrh_ctx *ctx, const lrh_blanker_tables *t);   /* NULL: clever blanker off again */
/* make_fft2 until FFT2_COMPLETE (fft2def.h:61; fft2.c:52-1848, mode 15), `batch` transforms.
   The caller checks (timf2_pn2-timf2_px) >= 4*N2 per transform as wcw.c:265-275 does. */
int lrh_make_fft2(lrh_ctx *ctx, lrh_ptrs *p, int batch);
/* fft2_mix1_fixed (fft2def.h:62; mix1.c:934-993 + set_mix1_phases mix1.c:781-861 + do_mix1 mix1.c:55-195) */
int lrh_fft2_mix1_fixed(lrh_ctx *ctx, lrh_ptrs *p, int batch);
/* fft1_mix1_fixed (fft2def.h / mix1.c:995-1042): the second-fft-disabled chain picks mix1.size bins straight from
   fft1_float at fft1_px; needs second_fft_enable == 0 */
int lrh_fft1_mix1_fixed(lrh_ctx *ctx, lrh_ptrs *p, int batch);

/* AFC variants (mix1.c:863-932, 1044-1097 with do_mix1_afc, mix1.c:648-768): the centre frequency comes per transform
   from mix1_fq_mid[nx] instead of mix1_selfreq, and the slope / curvature / start tables the AFC reads back are kept
   like the reference does.  The tables are rings of max_fft2n floats (max_fft1n with the second fft off) owned by
   the caller, as the reference's globals are (buf.c:1089-1092; initial values -1, -1, 0, 0: buf.c:1255-1258).  The
   caller supplies mix1_fq_mid for transform nx and for nx+1 (mix1.c:662-666); entries may be clamped (mix1.c:724-726).
   do_mix1 discards the frequency drift it is handed (mix1.c:103), so the signal path equals the fixed variant at the
   per-transform frequency; a frequency must be selected (lrh_set_mix1_selfreq >= 0) as in the reference. */
typedef struct lrh_afc {
  float *mix1_fq_mid, *mix1_fq_slope, *mix1_fq_curv, *mix1_fq_start;
  float baseband_bw_hz;
} lrh_afc;
int lrh_fft2_mix1_afc(lrh_ctx *ctx, lrh_ptrs *p, int batch, lrh_afc *afc);
int lrh_fft1_mix1_afc(lrh_ctx *ctx, lrh_ptrs *p, int batch, lrh_afc *afc);
/* make_fft3_all, transform part (fft3def.h; fft3.c:215-283): windowed e^{+j} transform of timf3 at timf3_px with DC at
   fft3_size/2, `batch` transforms spaced fft3_new_points; the GUI power averages of fft3.c:470-760 are not built */
int lrh_make_fft3_all(lrh_ctx *ctx, lrh_ptrs *p, int batch);
/* fft3_mix2, mixer_mode 1 part (mix2.c:145-176; two channels mix2.c:313-628): mix2.size bins around fft3_size/2 times
   bg_filterfunc, fftback, overlap-add into baseb_raw.  The rest of the reference function (carrier filter, demodulators)
   is not on this path; the harness stops the compiled reference at its thread-command check (mix2.c:749) to pin this part. */
int lrh_fft3_mix2(lrh_ctx *ctx, lrh_ptrs *p, int batch);
/* Two coupled channels: fft3_mix2 first turns the channel pair (X, Y) into the wanted polarisation and its orthogonal,
     A = c1 X + (c2 - j c3) Y   -> baseb_raw            B = c1 Y - (c2 + j c3) X   -> baseb_raw_orthog
   (mix2.c:340-343, 377-380; pg.c1..c3 of pol_graph.c), the coherent combine of the two receivers.  One channel per GPU:
   each context weighs its own bins into both sums, the sums are an all-reduce, and the context of channel 0 carries on
   with A, that of channel 1 with B, so each runs one back transform and its LRH_RING_BASEB_RAW is the reference's
   baseb_raw resp. baseb_raw_orthog:
     lrh_set_pol(ctx, c1, c2, c3)                 once (and whenever the control plane changes the polarisation)
     lrh_mix2_pol_begin(ctx, p, batch, &count)    own contributions of the `batch` transforms at p->fft3_px to
                                                  LRH_X_POL: float [2 (A,B)][batch][mix2.size][2], count floats in all
     -> all-reduce(sum) LRH_X_POL[0..count)
     lrh_fft3_mix2(ctx, p, batch)                 filter, back transform, overlap-add of A (channel 0) or B (channel 1)
   Without lrh_set_pol a coupled context filters its own channel, like a single-channel one. */
int lrh_set_pol(lrh_ctx *ctx, float c1, float c2, float c3);
/* The same step for more than two receivers (phased array, one channel per GPU; beyond the reference, which stops at two
   channels -- parity unpinned past lrh_set_pol's case): this channel's complex weights in the two sums, A = sum_c wa_c X_c and
   B = sum_c wb_c X_c (two beams).  lrh_set_pol(c1,c2,c3) is wa = c1, wb = -(c2 + j c3) on channel 0 and wa = c2 - j c3,
   wb = c1 on channel 1.  Needs fft3 configured; works without cfg.blanker_channels (then every context carries on with A,
   the context with timf1_channel_index 1 of a coupled pair with B). */
int lrh_set_combine_weights(lrh_ctx *ctx, float wa_re, float wa_im, float wb_re, float wb_im);
int lrh_mix2_pol_begin(lrh_ctx *ctx, const lrh_ptrs *p, int batch, size_t *count);
int lrh_set_bg_filterfunc(lrh_ctx *ctx, const float *bg_filterfunc /* fft3_size floats (baseb_graph.c:1246) */);
/* bg.mixer_mode = 2 (mix2.c:217-246): lrh_fft3_mix2 then forms baseb_raw with the FIR basebraw_fir[0 .. pts) (symmetric, centre pts / 2, odd pts;
   make_bg_filter derives it from the filter function, baseb_graph.c:1560-1634) on the timf3 samples, decimating by fft3_size / mix2.size, instead
   of filtering fft3's bins; p->timf3_py walks like timf3_px.  fir = NULL: back to mixer_mode 1.  One channel. */
int lrh_set_basebraw_fir(lrh_ctx *ctx, const float *basebraw_fir, int basebraw_fir_pts);
/* compute_timf2_powersum (wcw.c:80-138): weak-signal power per block of released timf2 data, for the S/N meter */
int lrh_compute_timf2_powersum(lrh_ctx *ctx, lrh_ptrs *p);
int lrh_set_mix1_selfreq(lrh_ctx *ctx, double fq);        /* mix1_selfreq[0]; <0 deselects               */
int lrh_get_mix1_state(lrh_ctx *ctx, lrh_mix1_state *st);

/* Whole wideband chain for `nblocks` fft1 blocks in the order of wideband_dsp's single-CPU branch
   (wcw.c:1036-1118), batched `batch` blocks at a time: the blanker and fft2/mix1 run once per batch.  With fft3
   configured (cfg.fft3_n > 0) and a frequency selected the narrowband side follows mix1 like do_fft3 / do_mix2 follow
   EVENT_FFT3 / EVENT_MIX2 (wcw.c:1788,1828; fft3.c:35-60; mix2.c:41-80): every transform timf3 holds through
   lrh_make_fft3_all and lrh_fft3_mix2 -- unless a coherent combine is set (lrh_set_pol / lrh_set_combine_weights): its
   collective belongs between lrh_mix2_pol_begin and lrh_fft3_mix2, so the caller then runs those calls itself.
   The pointers in *p are final when the call returns.  Device work of the call's LAST batch (its blanker, fft2 / mix1 and narrowband
   launches) may still be held back at that moment, because large batches run a schedule that issues them one batch late beside the
   next batch's transforms: the next lrh_wideband_dsp issues them as if both calls had been one, and so does -- first thing -- every
   other entry point that reads or changes what the chain has produced (lrh_export*, lrh_sync, the state getters, the stage and table
   functions ...; not the producer-side lrh_timf1_write*).  A caller that passes one batch per call therefore gets the same schedule
   as one that passes many.  LRH_PERSIST=0 in the environment makes every call issue all of its work before it returns. */
int lrh_wideband_dsp(lrh_ctx *ctx, lrh_ptrs *p, int nblocks, int batch);
/* Two coupled channels (cfg.blanker_channels = 2) through lrh_wideband_dsp: the exchanges that sit between the stage calls (see
   lrh_blanker_begin, lrh_fft2_xy_begin, lrh_mix2_pol_begin) are made by a function the caller registers.  At each exchange point the
   library has enqueued everything that fills the buffer on `stream` (the context's own stream, a hipStream_t) and calls
       fn(user, which, op, device_ptr, count, stream, own)
   which must enqueue the collective ON THAT STREAM (RCCL: ncclAllReduce / ncclAllGather with the stream; torch: under an
   ExternalStream) and return 0 without waiting; the library then enqueues the consumers behind it.  op LRH_XOP_SUM: in-place all-reduce
   (sum) of `count` floats at device_ptr; op LRH_XOP_GATHER: all-gather of the two slots of `count` floats each that start at device_ptr
   (slot r is the one channel r filled: the caller's own slot is its send buffer -- unless `own` is not NULL: then the caller's `count`
   floats lie at `own`, still where the stage before left them (LRH_X_BINS: the fft2 ring, which saves a device copy of every transform),
   its own slot has NOT been filled and need not be, and the collective has to deliver the partner's slot only; fn(..., own) with a
   plain all-gather: send buffer `own`, receive buffer device_ptr).  Nothing waits on the host, so a whole call is enqueued
   like a single-channel one: fft1 / sums / weak stream fused as usual, blanker_begin -> SUM(LRH_X_PWR) [-> GATHER(LRH_X_WEAK) with the
   linear blanker's tables] -> blanker -> SUM(LRH_X_STAT) -> blanker_finish, make_fft2 -> GATHER(LRH_X_BINS) -> cross products and
   waterfall line, mix1, fft3 -> SUM(LRH_X_POL) (after lrh_set_pol) -> mix2, once per batch.  fn = NULL removes it (the stage calls
   remain available; lrh_wideband_dsp then refuses coupled contexts as before). */
enum { LRH_XOP_SUM = 0, LRH_XOP_GATHER = 1 };
typedef int (*lrh_exchange_fn)(void *user, int which, int op, void *device_ptr, size_t count, void *stream, const void *own);
int lrh_set_exchange(lrh_ctx *ctx, lrh_exchange_fn fn, void *user);

/* ---- host-visible side outputs ---- */
/* lrh_export: synchronous for the caller (dst is filled on return), ordered behind everything the calls so far have enqueued.  Spans of up to
   512 KiB travel on a copy stream of the context's own into a page-locked slot and the caller waits for them WITHOUT holding the context's
   lock: the other stage threads go on enqueueing meanwhile, and the wait covers the work queued up to this call, not what they add. */
int lrh_export(lrh_ctx *ctx, lrh_ring ring, void *dst, size_t offset_elems, size_t count_elems);
/* The same in two halves, for products nobody needs before the next call of the stage that made them (the averaged spectra the graphs draw):
   lrh_export_begin enqueues the copy behind everything the calls so far have enqueued and returns a ticket at once; lrh_export_end(ticket) waits
   for that copy (without the context's lock) and fills the `dst` given to begin, which must stay valid until then.  Ticket 0: the span was
   fetched by begin itself (no slot free, or more than 512 KiB) and lrh_export_end has nothing to do.  At most 16 tickets at a time. */
int lrh_export_begin(lrh_ctx *ctx, lrh_ring ring, void *dst, size_t offset_elems, size_t count_elems, int *ticket);
int lrh_export_end(lrh_ctx *ctx, int ticket);
int lrh_get_blanker_state(lrh_ctx *ctx, lrh_blanker_state *st);                                   /* synchronous */
/* Payload of the NET_RXOUT_TIMF2 multicast (what MAP65 and slave Linrads receive; rxin.c:944-966, float form): for `count`
   samples from timf2 position timf2_pt (in floats like the reference's pointer, a multiple of 4; wraps) one complex float
   hg_map65_gain * (weak + hg.map65_strong * strong) each, formed on the device from the planar rings, so 8 bytes per
   sample cross PCIe instead of 16.  The packet framing (NET_RX_STRUCT header, block numbers, 1392-byte payloads,
   globdef.h:1283-1294) stays with the host's network thread.  Synchronous. */
int lrh_export_timf2_net(lrh_ctx *ctx, float *dst, int timf2_pt, int count, float map65_gain, float map65_strong);
/* Payload of the NET_RXOUT_FFT1 multicast (wcw.c:1024-1043, network.c:383-388): `batch` transforms as fft1_b leaves them -- window,
   transform, DC at fft1_size/2, mirror-image calibration and passband direction, but NOT the filter correction of fft1_c -- for the
   blocks starting at timf1p_ref, 2*fft1_size floats each.  The hot path folds the correction into the transform's store and never
   holds this form, so it is recomputed from the timf1 ring (which keeps at least a second of input, buf.c:744-770): one extra
   fft1 pass, only paid by installations that multicast this stage.  Synchronous. */
int lrh_export_fft1_net(lrh_ctx *ctx, float *dst, int timf1p_ref, int batch);
/* same span, device-to-device into a caller-owned device buffer (e.g. the RCCL exchange buffer of the
   cross-channel power sum, fft1.c:4138); synchronous on the context stream */
int lrh_export_device(lrh_ctx *ctx, lrh_ring ring, void *dst_device, size_t offset_elems, size_t count_elems);
/* Same copy without the host-side wait: it is ordered on the context's stream, whose handle (a hipStream_t) lets the
   caller chain its own device work -- e.g. an RCCL all-reduce of the exported block -- with stream/event waits only. */
int lrh_export_device_async(lrh_ctx *ctx, lrh_ring ring, void *dst_device, size_t offset_elems, size_t count_elems);
void *lrh_stream(lrh_ctx *ctx);
/* A consumer ordered only on lrh_stream (RCCL on a cached lrh_exchange_ptr, a torch ExternalStream) must call lrh_flush first: after a
   batched lrh_wideband_dsp the launches of its last round may still be held back (see there); lrh_flush issues them without waiting. */
int lrh_flush(lrh_ctx *ctx);
int lrh_sync(lrh_ctx *ctx);
/* Back-pressure for a caller that drives the stages from its own threads, one per stage, as Linrad does (THREAD_TIMF2: fft1_c / make_timf2 /
   the blanker, wcw.c:401-441; THREAD_SECOND_FFT: make_fft2, wcw.c:250-304; the narrowband thread: fft2_mix1_*, wcw.c:1240-1405).  The stage
   calls only ENQUEUE device work, so a host that is faster than the device would run ahead of it by whatever the rings hold -- seconds in
   Linrad's sizing -- and every hand-over would be a call of a single block.  lrh_stage_wait(ctx, stage) returns once the device has finished the
   work the LAST call of that stage enqueued (at once if there was none).  It holds no lock while it waits: other threads keep enqueueing.
   Called at the top of a stage's stand-in before it counts what has accumulated (integration/hipshim.c), the stage then takes everything that
   arrived meanwhile in one call, and the call size follows the load: one block per call while the device keeps up, larger batches -- whose
   device time hardly grows -- when it does not.
   LRH_STAGE_TIMF2: behind lrh_make_timf2's kernels; LRH_STAGE_FFT2: behind lrh_make_fft2's; LRH_STAGE_MIX1: behind lrh_fft2_mix1_* / lrh_fft1_mix1_*. */
enum { LRH_STAGE_TIMF2 = 0, LRH_STAGE_FFT2 = 1, LRH_STAGE_MIX1 = 2, LRH_STAGE_COUNT = 3 };
int lrh_stage_wait(lrh_ctx *ctx, int stage);
/* ... and the same with `lag` calls left in flight (0 .. 3): returns once the device has finished the call `lag` before the newest one of that
   stage, so the stage's thread enqueues call n while the device still works on calls n-1 .. n-lag (round 6: with lag 0 a thread alternates
   between enqueueing and waiting and the device idles while the host enqueues; wcw.c:401-441 -- Linrad's own stage threads never wait for each other
   beyond the ring pointers either).  The products a lagged caller reads back belong to the call it has waited for. */
int lrh_stage_wait_lag(lrh_ctx *ctx, int stage, int lag);
/* Self-test of the C boundary (no device needed): raises a C++ exception inside the library -- kind 0 std::bad_alloc, 1 std::out_of_range,
   2 a value not derived from std::exception -- and returns what every entry point returns in that case: LRH_EINTERNAL, never a call of
   std::terminate (a C host reports it like any other code, lxsys.c:494-505).  Any other kind: LRH_OK. */
int lrh_selftest_exception(int kind);

/* ---- measurement hooks (bench.py): HIP events on the context's own stream ---- */
int lrh_timer_start(lrh_ctx *ctx);
int lrh_timer_stop(lrh_ctx *ctx, float *elapsed_ms);      /* synchronises on the stop event */
/* per-kernel accumulated time since the last reset, measured with hipEvents around each launch when enabled.
   on = 1: lrh_wideband_dsp falls back to its serial order (stand-alone kernel times); on = 2: the two-stream schedule is
   kept, so a stage's time is what it takes next to the kernels it really shares the chip with */
int lrh_profile_enable(lrh_ctx *ctx, int on);
int lrh_profile_get(lrh_ctx *ctx, const char *kernel, double *total_ms, long *launches);

/* ---- deterministic test signal (the build's own generator; role of internal_generator rxin.c:43-188).
        Host code, no GPU needed. Interleaved int16 I,Q. ---- */
typedef struct lrh_synth {
  uint64_t seed;
  float noise_sigma;            /* LSB per component                                              */
  int ncarriers;
  float carrier_bin[16];        /* cycles per fft_size samples                                     */
  float carrier_amp[16];        /* LSB                                                             */
  int fft_size;                 /* reference size for carrier_bin                                  */
  int pulse_period;             /* samples; 0: none                                                */
  int pulse_len;
  float pulse_amp;
  float chan_phase;             /* radians applied to carriers (multi-channel sky phase)           */
} lrh_synth;
void lrh_synth_defaults(lrh_synth *s, int fft1_size, int channel);       /* SURVEY.md 8(d) signal */
int  lrh_synth_iq(const lrh_synth *s, int64_t first_sample, int64_t nsamples, int16_t *dst);

#ifdef __cplusplus
}
#endif
#endif /* LINRAD_HIP_H */
