/*
 * linrad_oracle.c -- TEST INFRASTRUCTURE ONLY (see linrad_oracle.h).
 *
 * Plain-C float32 restatement of the reference hot path.  Every function cites the
 * reference lines it follows (paths relative to fventuri/linrad).  Not a copy: the
 * reference keeps state in ~80 globals and unrolls each FFT variant by hand; here
 * one context owns the rings and two generic radix-2 kernels do all transforms.
 * Where the order of float operations decides the result (butterflies, phase
 * accumulators, truncating conversions) the reference order is kept on purpose.
 */
#include "linrad_oracle.h"
typedef float f32;                 /* the ABI's float, whatever the arithmetic below runs in */
#ifdef LRO_F64
/* liblinrad_oracle64.so (make oracle64): the same source with every `float` below a double -- tables, rings, accumulators.  It is the
   "truth" the parity tests measure BOTH float32 implementations against (the compiled reference's goldens and the HIP path):
   tests/golden/make_truth.py, tests/paritylib.py.  The ABI is unchanged: arrays cross the boundary as float32 (f32) and are
   converted there; lro_export_f64 hands out the rings unrounded. */
#define float double
#endif
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PI_L 3.1415926535897932   /* globdef.h:93 */
#define NATLOG 2.718281828459045
#define FFT1_SMALL 0.00000001F    /* globdef.h:88 */
#define FFT2_WATERFALL_ZERO 0.012 /* graphcal.h:8 */

typedef struct { float sin, cos; } cosin_t;   /* COSIN_TABLE, globdef.h */
/* arrays across the ABI: plain copies in the float32 build, conversions in the float64 one */
static void in_f32(float *dst, const f32 *src, size_t n) { for (size_t i = 0; i < n; i++) dst[i] = src[i]; }
static void out_f32(f32 *dst, const float *src, size_t n) { for (size_t i = 0; i < n; i++) dst[i] = (f32)src[i]; }
#ifdef LRO_F64
#define LRO_F32_ONLY(c) return LRH_ESTATE      /* entry points the truth build does not serve (their hand-over is float32 memory) */
#else
#define LRO_F32_ONLY(c) do { } while (0)
#endif

struct lro_ctx {
  lrh_config cfg;
  int N1, I1, M1, N2, I2, M2, Nm, Im, Mm, mix1_n;
  /* tables */
  cosin_t *fft1tab, *fft2tab, *mix1tab;
  float *fft1_window;          /* mode-1 storage, fft0.c:907-920 */
  float *fft1_inverted_window; /* mode 3 */
  float *fft1_filtercorr;      /* 2*N1 */
  float *fft1_desired;
  float *fft2_window;          /* mode 4, N2 */
  float *mix1_fqwin;           /* mode 5, Nm/2+1 */
  float *fft1_foldcorr;        /* N1 complex, NULL: no I/Q mirror-image calibration (fft1_calibrate_flag & CALIQ) */
  /* two coupled channels (cfg.blanker_channels = 2): summed power ring the blanker decides on, exchange buffers, and
     what lro_first_noise_blanker leaves for lro_blanker_finish */
  float *xpol; float pol_w[4]; int pol_set, pol_batch;   /* this channel's weights (wa_re, wa_im, wb_re, wb_im) */   /* LRH_X_POL [2][max_fft3n][Nm2][2]; pg.c1..c3 */
  float *xbins, *fft2_xypower, *fft2_xysum;   /* LRH_X_BINS [2][max_fft2n][N2][2]; TWOCHAN_POWER rings (fft2.c:1622-1640) */
  float *pwr_sum, *xbuf; float xstat[2]; int x_pbeg, x_count, fin_pending, fin_do_update; float fin_llf;
  /* linear blanker on two coupled channels: both channels' weak samples around the span (exchange LRH_X_WEAK, all-gather) and the
     partner channel's samples in ring places (2 floats per sample) */
  float *xweak, *tf_partner; int x_span, xw_count;
  float ch2_c1, ch2_c2; int ch2_set;   /* pg_ch2_c1 / pg_ch2_c2 when this context carries the second RF channel */
  float *mix1_window, *mix1_sin2win, *mix1_cos2win; int Xm;   /* crossover-window mix1 (prepare_mixer, buf.c:55-111); Xm = crossover_points */
  float *wg_waterf_yfac;       /* N1 */
  float *liminfo;
  void *sellim;                /* state of lro_fft1_update_liminfo (old_liminfo, liminfo_wait, ...), see there */
  void *spurs;                 /* lro_spurs: spurs being tracked (lro_spur_config / lro_spur_set), NULL: none */
  /* rings */
  int16_t *timf1;
  float *fft1_float, *fft1_sumsq, *fft1_slowsum;
  float *timf2_float, *timf2_pwr;
  float *fft2_float, *fft2_power, *fft2_powersum;
  int16_t *wg_waterf;
  double *wf_pre;              /* float64 build only: the waterfall lines before the truncation to short, same places as wg_waterf */
  float *timf3_float;
  float *timf2_blockpower;
  /* fft3 / mix2 */
  int N3, I3, M3, Nm2, Im2, Mm2;
  cosin_t *fft3tab, *mix2tab;
  float *fft3_window, *fft3, *bg_filterfunc, *baseb_raw;
  float *basebraw_fir; int basebraw_fir_pts;                  /* bg.mixer_mode = 2 (lro_set_basebraw_fir), NULL: mixer_mode 1 */
  float *mix2_window, *mix2_sin2win, *mix2_cos2win; int Xm2;  /* crossover-window mix2 (prepare_mixer(&mix2, THIRD_FFT_SINPOW), baseb_graph.c:899) */
  float *tmp;                  /* scratch, 8*max(N1,N2) floats */
  /* masks */
  int fft1n_mask, fft1_mask, fft1_sumsq_mask, timf2pow_mask, timf2_mask, fft2n_mask, timf3_mask, timf1_bytemask;
  /* blanker scalars (blnkvar.c) */
  lrh_blanker_state bs;
  /* linear blanker (lro_set_blanker_tables) */
  float amp_factor;            /* liminfo_amplitude_factor (sellim.c:119-155; blank1.c:143) */
  lrh_sellim wl_par; int wl_on, wl_fft2, wl_cnt1, wl_cnt2;   /* lro_wideband_limiter */
  lrh_exchange_fn xfn; void *xuser;                          /* lro_set_exchange */
  int corr_on, slowcorr_tot_avgnum; float *xspec, *fft1_corrsum, *fft1_slowcorr; double *fft1_slowcorr_tot;   /* lro_set_correlation */
  lrh_blanker_tables bt; f32 *bt_refpulse, *bt_phasefunc; int *bt_pulindex; unsigned char *blanker_flag; int clever_on;
  /* mix1 scalars (selvar.c) */
  lrh_mix1_state ms;
  double old_mix1_selfreq;
};

/* ------------------------------------------------------------------ tables */

/* make_sincos, fft0.c:1263-1284: angle accumulated in double, stored as float */
static void make_sincos(int size, cosin_t *tab)
{
  double x = 0, step = (double)(PI_L / (size / 2));
  for (int i = 0; i < size / 2; i++) { tab[i].sin = (float)sin(x); tab[i].cos = (float)cos(x); x += step; }
}

/* make_window, fft0.c:812-921.  mo: 1 interleaved half storage, 3 inverted, 4 full symmetric,
   5 erfc edge (sz/2+1 points).  n: sin power 1..7, 8 Gaussian, 9 erfc. */
static void make_window(int mo, int sz, int n, float *win);
void lro_make_window(int mo, int sz, int n, f32 *win)
{
  const size_t cnt = (size_t)(mo == 2 ? sz + 1 : (mo == 5 || mo == 3) ? sz / 2 + 1 : sz);
  float *t = calloc(cnt + 2, sizeof(float));
  in_f32(t, win, cnt);                 /* (n == 0 leaves the caller's array as it was) */
  make_window(mo, sz, n, t);
  out_f32(win, t, cnt);
  free(t);
}
static void make_window(int mo, int sz, int n, float *win)
{
  double x, z, sumsq = 0, e1, e2;
  int i, size = sz;
  if (mo == 2) size = 2 * sz;          /* half window of a 2*sz-point transform in natural order, win[0..sz] (fft0.c:838) */
  if (mo == 5) {
    e1 = 3.2; e2 = 13.0 / sz;
    for (i = 0; i <= sz / 2; i++) { win[i] = 0.5F * (float)erfc(e1); e1 -= e2; }
    return;
  }
  if (n == 0) return;
  z = n;
  float *h = (float *)malloc(sizeof(float) * (size / 2 + 1));
  if (n == 9) {
    e1 = 4.4; e2 = 40.0 / size; if (size < 128) e2 /= 1.5; if (size < 64) e2 /= 1.7;
    for (i = 0; i <= size / 2; i++) { h[i] = 0.5F * (float)erfc(e1); sumsq += h[i] * h[i]; e1 -= e2; }
  } else if (n == 8) {
    e1 = 0; e2 = 9.8 / size;
    for (i = size / 2; i >= 0; i--) { h[i] = (float)pow(NATLOG, -e1 * e1); sumsq += h[i] * h[i]; e1 += e2; }
  } else {
    x = 0;
    for (i = 0; i <= size / 2; i++) { h[i] = (float)pow(sin(x), z); sumsq += h[i] * h[i]; x += PI_L / size; }
  }
  if (mo == 3) {                       /* inverted window over size/2+1 points, fft0.c:883-891 */
    win[0] = 1;
    for (i = 1; i <= size / 2; i++) win[i] = 1 / h[i];
    free(h); return;
  }
  z = 1 / sqrt(2 * sumsq / size);      /* unit mean square, fft0.c:892-896 */
  for (i = 0; i <= size / 2; i++) h[i] *= (float)z;
  if (mo == 2) {
    for (i = 0; i <= size / 2; i++) win[i] = h[i];
  } else if (mo == 4) {
    for (i = 0; i <= size / 2; i++) win[i] = h[i];
    for (i = size / 2 + 1; i < size; i++) win[i] = h[size - i];
  } else {                             /* mo 1: win[2i]=w[i], win[2i+1]=w[N/2-i], fft0.c:907-920 */
    for (i = 0; i < size / 2; i++) { win[2 * i] = h[i]; win[2 * i + 1] = h[size / 2 - i]; }
  }
  free(h);
}

/* make_interleave_ratio, buf.c:113-136 */
static float interleave_ratio(int sinpow)
{
  if (sinpow == 0) return 0;
  if (sinpow == 9) return 0.625f;
  if (sinpow == 8) return 0.8f;
  return (float)(2 * asin(pow(0.5, 1.0 / sinpow)) / PI_L);
}

/* clear_fft1_filtercorr + make_filcorrstart, fft1.c:4653-4724 (int16 IQ input, complex fft) */
static void default_filtercorr(lro_ctx *c)
{
  int N = c->N1;
  float start = 150 * (float)N * (float)pow((double)N, -0.4);
  const int real = c->cfg.timf1_real_input != 0;
  if (c->cfg.timf1_dword_input) { start *= 4096; start *= real ? 16 : 12; }   /* make_filcorrstart, fft1.c:4656-4663: left-justified int32; real: permute == 2 */
  start = (float)c->cfg.fft1_gain / start;
  for (int i = 0; i < N; i++) { c->fft1_desired[i] = 1; c->fft1_filtercorr[2 * i] = start; c->fft1_filtercorr[2 * i + 1] = 0; }
  float t1 = 0.125F * (float)PI_L, t2 = 0, t3;
  int i = 0, k = N - 1;
  while (t2 < 0.5 * PI_L) {
    t3 = (float)(sin(t2) * sin(t2));
    if (!real) { c->fft1_desired[i] = t3; c->fft1_filtercorr[2 * i] = t3 * start; }     /* fft1.c:4707-4711: the low edge only for I/Q */
    c->fft1_desired[k] = t3; c->fft1_filtercorr[2 * k] = t3 * start;
    t2 += t1; i++; k--;
  }
}

/* make_wg_yfac, wide_graph.c:955-1001, second-fft branch, 1 channel, float fft2 */
static void default_yfac(lro_ctx *c)
{
  float t1 = (float)(FFT2_WATERFALL_ZERO) / ((float)c->N2 * (float)c->N1);
  t1 /= (float)sqrt((float)(c->cfg.waterfall_avgnum));
  t1 *= (float)(1 << (2 * c->cfg.bckfft_att_n));
  t1 *= (float)(1 + 1 / (0.5 + c->cfg.fft1_sinpow));
  if (c->cfg.blanker_channels == 2) t1 *= 4.0f;             /* ui.rx_rf_channels^2, wide_graph.c:985 */
  for (int i = 0; i < c->N1; i++)
    c->wg_waterf_yfac[i] = (c->fft1_desired[i] > 0.3162278) ? t1 / (float)pow(c->fft1_desired[i], 2.0) : t1 * 10;
  c->wg_waterf_yfac[0] = t1; c->wg_waterf_yfac[c->N1 - 1] = t1;
}

/* ------------------------------------------------------------------ FFT kernels */

static unsigned bitrev(unsigned v, int n) { unsigned r = 0; for (int i = 0; i < n; i++) { r = (r << 1) | (v & 1); v >>= 1; } return r; }

/* Radix-2 decimation in frequency, natural in -> bit-reversed out, twiddle e^{+j s theta} from a float table
   (the loop of bulk_of_dif fft0.c:161-195 / fftback fft0.c:481-518 / big_fftforward fft0.c:753-793, all stages).
   stride = nch*2 floats per point, so two interleaved streams can share one pass (timf2.c:689). */
static void dif_stages(int N, int n, float *x, const cosin_t *tab, int sgn, int stride)
{
  int half = N / 2, inc = 1;
  for (int st = 0; st < n; st++) {
    for (int base = 0; base < N; base += 2 * half) {
      int it = 0;
      for (int a = base; a < base + half; a++) {
        int b = a + half;
        float *pa = x + (size_t)a * stride, *pb = x + (size_t)b * stride;
        float t1 = pa[0], t2 = pb[0], t3 = pa[1], t4 = pb[1];
        float x1 = t1 - t2, x2 = t3 - t4;
        pa[0] = t1 + t2; pa[1] = t3 + t4;
        float co = tab[it].cos, si = tab[it].sin;
        if (sgn > 0) { pb[0] = co * x1 - si * x2; pb[1] = si * x1 + co * x2; }
        else         { pb[0] = co * x1 + si * x2; pb[1] = -si * x1 + co * x2; }
        it += inc;
      }
    }
    inc *= 2; half /= 2;
  }
}

static void bitrev_inplace(int N, int n, float *x, int stride)
{
  for (unsigned i = 0; i < (unsigned)N; i++) {
    unsigned j = bitrev(i, n);
    if (j > i) for (int k = 0; k < 2; k++) { float t = x[(size_t)i * stride + k]; x[(size_t)i * stride + k] = x[(size_t)j * stride + k]; x[(size_t)j * stride + k] = t; }
  }
}

static void fft_natural(int n, float *x, int dir)
{
  int N = 1 << n; cosin_t *tab = malloc(sizeof(cosin_t) * (N / 2 + 1)); make_sincos(N, tab);
  dif_stages(N, n, x, tab, dir, 2); bitrev_inplace(N, n, x, 2); free(tab);
}
static void fft_abi(int n, f32 *x, int dir)
{
  const size_t cnt = (size_t)2 << n;
  float *t = malloc(sizeof(float) * cnt);
  in_f32(t, x, cnt); fft_natural(n, t, dir); out_f32(x, t, cnt); free(t);
}
void lro_fft_forward(int n, f32 *x) { fft_abi(n, x, -1); }
void lro_fft_backward(int n, f32 *x) { fft_abi(n, x, +1); }

/* ------------------------------------------------------------------ open / close */

static int ispow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
#ifdef LRO_F64
static void *zal(size_t n) { return calloc(2 * n + 64, 1); }     /* byte counts below are written for 4-byte floats */
#else
static void *zal(size_t n) { return calloc(n + 64, 1); }
#endif

/* prepare_mixer, buf.c:55-111: inverted window (mode 3) and the sin^2 / cos^2 crossover functions of a mixer whose window is neither
   none nor sin^2; returns crossover_points */
static int prepare_mixer(int size, int interleave, int newp, int sp, float *window, float *sin2win, float *cos2win)
{
  int X = 0;
  if (sp == 0 || sp == 2) return 0;
  make_window(3, size, sp, window);
  if (sp == 9) X = size / 8;
  else if (sp == 8) X = size / 16;
  else {
    unsigned int i = interleave / 2;
    float t1 = window[i];
    while (window[i] < 30 * t1 && i > 0) { i--; X++; }
    if (X > 0.75 * newp) X = 0.75 * newp;
    if (X > interleave / 2) X = interleave / 2;
  }
  float t1 = 0.25 * PI_L / X;
  unsigned int j = (size - newp) / 2, k = j;
  k += X / 2; j -= X / 2;
  for (int i = 0; i < X; i++) {
    cos2win[i] = window[k] * pow(cos(t1), 2.0);
    sin2win[i] = window[j] * pow(sin(t1), 2.0);
    k--; j++;
    t1 += 0.5 * PI_L / X;
  }
  return X;
}

int lro_open(const lrh_config *cfg, lro_ctx **out)
{
  if (!cfg || !out || cfg->struct_size != (int)sizeof(lrh_config)) return LRH_EINVAL;
  if (cfg->rx_rf_channels != 1) return LRH_EINVAL;
#ifdef LRO_F64
  if (cfg->blanker_channels == 2) return LRH_EINVAL;       /* the truth build serves the single-channel chain */
#endif
  if (cfg->fft1_n < 6 || cfg->fft1_n > 16 || cfg->fft2_n < 6 || cfg->fft2_n > 22) return LRH_EINVAL;
  if (!ispow2(cfg->timf1_bytes) || !ispow2(cfg->max_fft1n) || !ispow2(cfg->fft1_sumsq_bufsize) ||
      !ispow2(cfg->timf2pow_size) || !ispow2(cfg->max_fft2n) || !ispow2(cfg->timf3_size)) return LRH_EINVAL;
  lro_ctx *c = zal(sizeof(*c)); if (!c) return LRH_ENOMEM;
  c->cfg = *cfg;
  int N1 = c->N1 = 1 << cfg->fft1_n, N2 = c->N2 = 1 << cfg->fft2_n;
  /* buf.c:303-304 */
  c->I1 = (int)(1 + interleave_ratio(cfg->fft1_sinpow) * N1); c->I1 &= 0xfffe;
  if (cfg->second_fft_enable) {
    /* buf.c:432-455: mix1 first, fft2 interleave re-derived from it */
    c->mix1_n = cfg->fft2_n - cfg->mix1_bandwidth_reduction_n; if (c->mix1_n < 3) c->mix1_n = 3;
    c->Nm = 1 << c->mix1_n;
    c->Im = (int)(interleave_ratio(cfg->fft2_sinpow) * c->Nm); c->Im &= 0xfffffffe; c->Mm = c->Nm - c->Im;
    c->I2 = c->Im * (N2 / c->Nm); c->M2 = N2 - c->I2;
  } else {
    /* buf.c:315-327: mix1 sized from fft1, fft1 interleave re-derived so it divides evenly */
    c->mix1_n = cfg->fft1_n - cfg->mix1_bandwidth_reduction_n; if (c->mix1_n < 3) c->mix1_n = 3;
    c->Nm = 1 << c->mix1_n;
    c->Im = (int)(interleave_ratio(cfg->fft1_sinpow) * c->Nm); c->Im &= 0xfffffffe; c->Mm = c->Nm - c->Im;
    c->I1 = c->Im * (N1 / c->Nm);
    c->I2 = 0; c->M2 = N2;
  }
  c->M1 = N1 - c->I1;
  if (cfg->fft1_sumsq_bufsize < 2 * N1 || cfg->timf2pow_size < 2 * N1 || cfg->timf2pow_size < 2 * N2) { free(c); return LRH_EINVAL; }
  c->fft1n_mask = cfg->max_fft1n - 1; c->fft1_mask = cfg->max_fft1n * 2 * N1 - 1; c->fft1_sumsq_mask = cfg->fft1_sumsq_bufsize - 1;
  c->timf2pow_mask = cfg->timf2pow_size - 1; c->timf2_mask = 4 * cfg->timf2pow_size - 1; c->fft2n_mask = cfg->max_fft2n - 1;
  c->timf3_mask = cfg->timf3_size - 1; c->timf1_bytemask = cfg->timf1_bytes - 1;
  int NM = N1 > N2 ? N1 : N2;
  c->fft1tab = zal(sizeof(cosin_t) * N1); c->fft2tab = zal(sizeof(cosin_t) * N2); c->mix1tab = zal(sizeof(cosin_t) * c->Nm);
  c->fft1_window = zal(4 * (N1 + 8)); c->fft1_inverted_window = zal(4 * (N1 + 8)); c->fft1_filtercorr = zal(8 * N1);
  c->fft1_desired = zal(4 * N1); c->fft2_window = zal(4 * (N2 + 8)); c->mix1_fqwin = zal(4 * (c->Nm + 8));
  c->wg_waterf_yfac = zal(4 * N1); c->liminfo = zal(4 * N1);
  c->timf1 = zal(cfg->timf1_bytes); c->fft1_float = zal(sizeof(float) * cfg->max_fft1n * 2 * N1);
  c->fft1_sumsq = zal(4 * (size_t)cfg->fft1_sumsq_bufsize); c->fft1_slowsum = zal(4 * N1);
  c->timf2_float = zal(16 * (size_t)cfg->timf2pow_size); c->timf2_pwr = zal(4 * (size_t)cfg->timf2pow_size);
  if (cfg->blanker_channels == 2) { c->pwr_sum = zal(4 * (size_t)cfg->timf2pow_size); c->xbuf = zal(4 * (size_t)cfg->timf2pow_size); c->x_count = -1;
    c->xbins = zal(sizeof(float) * 4 * (size_t)cfg->max_fft2n * N2); c->fft2_xypower = zal(sizeof(float) * 4 * (size_t)cfg->max_fft2n * N2);
    c->fft2_xysum = zal(sizeof(float) * 4 * (size_t)N2);
  }
  if (cfg->fft3_n) c->xpol = zal(sizeof(float) * 4 * (size_t)cfg->max_fft3n * (1 << cfg->mix2_n));
  c->fft2_float = zal(sizeof(float) * 2 * N2 * cfg->max_fft2n); c->fft2_power = zal(sizeof(float) * N2 * cfg->max_fft2n);
  c->fft2_powersum = zal(4 * N2);
  c->wg_waterf = zal(2 * (size_t)cfg->wf_lines * cfg->wf_xpixels + 64);
#ifdef LRO_F64
  c->wf_pre = calloc((size_t)cfg->wf_lines * cfg->wf_xpixels + 16, sizeof(double));
#endif
  c->timf3_float = zal(4 * (size_t)cfg->timf3_size + 16 * c->Nm);
  c->timf2_blockpower = zal(4 * (size_t)(cfg->timf2_blockpower_size > 0 ? cfg->timf2_blockpower_size : 1));
  if (cfg->fft3_n > 0) {
    /* baseb_graph.c:636-645: mix2 interleave first, fft3 interleave re-derived from it */
    if (cfg->mix2_n < 3 || cfg->mix2_n > cfg->fft3_n || !ispow2(cfg->max_fft3n) || !ispow2(cfg->baseband_size)) { free(c); return LRH_EINVAL; }
    c->N3 = 1 << cfg->fft3_n; c->Nm2 = 1 << cfg->mix2_n;
    c->Im2 = (int)(interleave_ratio(cfg->fft3_sinpow) * c->Nm2); c->Im2 &= 0xfffffffe; c->Mm2 = c->Nm2 - c->Im2;
    c->I3 = c->Im2 * (c->N3 / c->Nm2); c->M3 = c->N3 - c->I3;
    c->fft3tab = zal(sizeof(cosin_t) * c->N3); c->mix2tab = zal(sizeof(cosin_t) * c->Nm2);
    c->fft3_window = zal(4 * (c->N3 + 8)); c->fft3 = zal(sizeof(float) * 2 * c->N3 * cfg->max_fft3n);
    c->bg_filterfunc = zal(4 * c->N3); c->baseb_raw = zal(8 * (size_t)cfg->baseband_size + 16 * c->Nm2);
    make_sincos(c->N3, c->fft3tab); make_sincos(c->Nm2, c->mix2tab);
    if (cfg->fft3_sinpow) make_window(1, c->N3, cfg->fft3_sinpow, c->fft3_window);     /* baseb_graph.c:3680 */
    for (int i = 0; i < c->N3; i++) c->bg_filterfunc[i] = 1.0f;
    c->mix2_window = zal(4 * (c->Nm2 + 8)); c->mix2_sin2win = zal(4 * (c->Nm2 + 8)); c->mix2_cos2win = zal(4 * (c->Nm2 + 8));
    c->Xm2 = prepare_mixer(c->Nm2, c->Im2, c->Mm2, cfg->fft3_sinpow, c->mix2_window, c->mix2_sin2win, c->mix2_cos2win);
  }
  c->tmp = zal(sizeof(float) * 8 * (NM > (1 << cfg->fft3_n) ? NM : (1 << cfg->fft3_n)));
  make_sincos(N1, c->fft1tab); make_sincos(N2, c->fft2tab); make_sincos(c->Nm, c->mix1tab);
  if (cfg->fft1_sinpow) make_window(cfg->timf1_real_input ? 2 : 1, N1, cfg->fft1_sinpow, c->fft1_window);   /* fft_cntrl[].window, fft1var.c:45,50 */
  if (cfg->fft1_sinpow != 0 && cfg->fft1_sinpow != 2) make_window(3, N1, cfg->fft1_sinpow, c->fft1_inverted_window);
  if (cfg->fft2_sinpow) make_window(4, N2, cfg->fft2_sinpow, c->fft2_window);
  make_window(5, c->Nm, 4, c->mix1_fqwin);            /* buf.c:1297 */
  {                                                        /* prepare_mixer(&mix1, ..), buf.c:1290 */
    const int sp = cfg->second_fft_enable ? cfg->fft2_sinpow : cfg->fft1_sinpow;
    c->mix1_window = zal(4 * (c->Nm + 8)); c->mix1_sin2win = zal(4 * (c->Nm + 8)); c->mix1_cos2win = zal(4 * (c->Nm + 8));
    c->Xm = prepare_mixer(c->Nm, c->Im, c->Mm, sp, c->mix1_window, c->mix1_sin2win, c->mix1_cos2win);
  }
  default_filtercorr(c); default_yfac(c);
  /* blanker start state: buf.c:418-431, hires_graph.c:1157-1162 */
  c->bs.timf2_noise_floor = cfg->timf2_noise_floor;
  c->amp_factor = 1;                                    /* init_blanker, buf.c:1909 */
  c->bs.timf2_despiked_pwr[0] = (float)cfg->timf2_noise_floor; c->bs.timf2_despiked_pwrinc[0] = 1;
  c->bs.timf2_despiked_pwr[1] = 0; c->bs.timf2_despiked_pwrinc[1] = 0;
  c->bs.stupid_bln_limit = (unsigned int)((float)cfg->timf2_noise_floor * cfg->stupid_bln_factor);
  c->ms.mix1_selfreq = -1; c->ms.mix1_point = -1; c->old_mix1_selfreq = -1;
  *out = c; return LRH_OK;
}

void lro_close(lro_ctx *c)
{
  if (!c) return;
  void *v[] = { c->fft1tab, c->fft2tab, c->mix1tab, c->fft1_window, c->fft1_inverted_window, c->fft1_filtercorr, c->fft1_desired,
                c->fft2_window, c->mix1_fqwin, c->mix1_window, c->mix1_sin2win, c->mix1_cos2win, c->wg_waterf_yfac, c->liminfo, c->timf1, c->fft1_float, c->fft1_sumsq, c->fft1_slowsum,
                c->timf2_float, c->timf2_pwr, c->fft2_float, c->fft2_power, c->fft2_powersum, c->wg_waterf, c->timf3_float, c->tmp, c->timf2_blockpower,
                c->fft3tab, c->mix2tab, c->fft3_window, c->fft3, c->bg_filterfunc, c->baseb_raw, c->fft1_foldcorr, c->pwr_sum, c->xbuf, c->xweak, c->tf_partner, c->xbins, c->fft2_xypower, c->fft2_xysum, c->xpol,
                c->bt_refpulse, c->bt_phasefunc, c->bt_pulindex, c->blanker_flag, c->wf_pre, c->mix2_window, c->mix2_sin2win, c->mix2_cos2win, c->basebraw_fir };
  for (size_t i = 0; i < sizeof(v) / sizeof(v[0]); i++) free(v[i]);
  if (c->sellim) {                     /* lro_sellim_state, defined with lro_fft1_update_liminfo */
    struct { float *old; unsigned char *wait; float *tmp, *group_min; int a, b, d; float *ftmp; float *rn; int *rf, *rl; } *s = c->sellim;
    free(s->old); free(s->wait); free(s->tmp - 8); free(s->group_min); free(s->ftmp - 8); free(s->rn); free(s->rf); free(s->rl); free(s);
  }
  free(c);
}

void lro_ptrs_init(const lro_ctx *c, lrh_ptrs *p)
{
  memset(p, 0, sizeof(*p));
  p->fft1_lowlevel_fraction = .75f;    /* buf.c:343 */
  p->fft1_sumsq_recalc = 0;            /* set_fft1_endpoints, fft1.c:4646 */
  (void)c;
}

int lro_get_derived(const lro_ctx *c, int *i1, int *i2, int *ms, int *mi, int *t3b)
{
  if (i1) *i1 = c->I1; if (i2) *i2 = c->I2; if (ms) *ms = c->Nm; if (mi) *mi = c->Im; if (t3b) *t3b = 2 * c->Mm;
  return LRH_OK;
}

int lro_set_filtercorr(lro_ctx *c, const f32 *fc) { if (fc) in_f32(c->fft1_filtercorr, fc, 2 * (size_t)c->N1); else default_filtercorr(c); return LRH_OK; }
int lro_set_liminfo(lro_ctx *c, const f32 *l) { in_f32(c->liminfo, l, c->N1); return LRH_OK; }
int lro_set_ch2_phasing(lro_ctx *c, f32 c1, f32 c2) { c->ch2_c1 = c1; c->ch2_c2 = c2; c->ch2_set = 1; return LRH_OK; }
int lro_set_foldcorr(lro_ctx *c, const f32 *foldcorr)
{
  if (!foldcorr) { free(c->fft1_foldcorr); c->fft1_foldcorr = NULL; return LRH_OK; }
  if (!c->fft1_foldcorr) c->fft1_foldcorr = malloc(sizeof(float) * 2 * c->N1);
  in_f32(c->fft1_foldcorr, foldcorr, 2 * (size_t)c->N1);
  return LRH_OK;
}
int lro_set_waterfall_yfac(lro_ctx *c, const f32 *y) { if (y) in_f32(c->wg_waterf_yfac, y, c->N1); else default_yfac(c); return LRH_OK; }

int lro_get_table(lro_ctx *c, const char *name, f32 *dst, int count)
{
  const float *src = NULL; int n = 0;
  if (!strcmp(name, "fft1_window")) { src = c->fft1_window; n = c->N1; }
  else if (!strcmp(name, "fft2_window")) { src = c->fft2_window; n = c->N2; }
  else if (!strcmp(name, "mix1_fqwin")) { src = c->mix1_fqwin; n = c->Nm / 2 + 1; }
  else if (!strcmp(name, "fft1_filtercorr")) { src = c->fft1_filtercorr; n = 2 * c->N1; }
  else if (!strcmp(name, "wg_waterf_yfac")) { src = c->wg_waterf_yfac; n = c->N1; }
  else if (!strcmp(name, "fft1_inverted_window")) { src = c->fft1_inverted_window; n = c->N1 / 2 + 1; }
  else if (!strcmp(name, "fft3_window")) { src = c->fft3_window; n = c->N3; }
  else return LRH_EINVAL;
  if (count > n) count = n;
  out_f32(dst, src, (size_t)count); return count;
}

int lro_timf1_write(lro_ctx *c, const void *src, int off, int nbytes)
{
  const char *s = src; char *d = (char *)c->timf1;
  for (int i = 0; i < nbytes; i++) d[(off + i) & c->timf1_bytemask] = s[i];
  return LRH_OK;
}

/* expand_rawdat, csplit.c:20-73: 9 packed bytes -> four int32 components {0, (2 bits)|0x20, lo, hi} */
int lro_timf1_write_packed18(lro_ctx *c, const void *src, int off, int packed_bytes)
{
  if (!c->cfg.timf1_dword_input || packed_bytes % 9 || (off & 15)) return LRH_EINVAL;
  const unsigned char *r = src; unsigned char *d = (unsigned char *)c->timf1;
  int i = off & c->timf1_bytemask;
  for (int j = 0; j < packed_bytes; j += 9) {
    unsigned char m = r[j + 8], n;
    for (int k = 0; k < 4; k++) {
      n = m & 0xc0; n |= 0x20;
      d[i + 4 * k] = 0; d[i + 4 * k + 1] = n; d[i + 4 * k + 2] = r[j + 2 * k]; d[i + 4 * k + 3] = r[j + 2 * k + 1];
      m <<= 2;
    }
    i = (i + 16) & c->timf1_bytemask;
  }
  return LRH_OK;
}

/* ------------------------------------------------------------------ fft1 */

/* fft1_b mode 7: fft1win_dif_one (fft1.c:413-447) + bulk_of_dif (fft0.c:161) + dif_permute_one (fft1.c:637-650),
   then the direction flip of fft1.c:3660-3679.  Result: out[k] = conj(FFT(x*w))[(k - N/2) mod N]. */
/* Real samples, fft1 version 2: fft1_reherm_dit_one (fft1_re.c:32-131).  2N reals from p0 = ref - 2*I1, sample ia and
   sample 2N-1-ia both times fft1_window[ia] (mode 2), the real transform Z = sum x e^{-j 2 pi nk / 2N} (the reference runs
   a split-radix real-to-Hermitian one, fft0.c:33; here the plain complex transform of (x, 0)), then
     direction > 0:  out[k] = (Im Z_k, Re Z_k), k = 1..N-1;  out[0] = (Re Z_N, Re Z_0)
     direction < 0:  out[N-k] = (Re Z_k, Im Z_k), k = 1..N-1;  out[0] = (Re Z_N, Re Z_N): the loop of fft1_re.c:123-128
                     runs to k = N, where the "imaginary" slot tmp[2N-k] is the Nyquist term itself
   and fft1_b is done (goto fft_done, fft1.c:3379: no CALIQ, no second flip). */
static void fft1_one_real(lro_ctx *c, int timf1p_ref, float *out)
{
  const int N = c->N1, n = c->cfg.fft1_n;
  const int dword = c->cfg.timf1_dword_input != 0, esz = dword ? 4 : 2;
  const int32_t *t32 = (const int32_t *)c->timf1;
  const int m = c->timf1_bytemask / esz;
  int pa = (timf1p_ref / esz - c->I1 * 2 + m + 1) & m;
  float *z = c->tmp;                            /* 4*N floats */
  const int win = c->cfg.fft1_sinpow != 0;
  for (int ia = 0; ia < 2 * N; ia++) {
    const float w = win ? c->fft1_window[ia < N ? ia : 2 * N - 1 - ia] : 1.0f;
    z[2 * ia] = (dword ? (float)t32[pa] : (float)c->timf1[pa]) * w; z[2 * ia + 1] = 0;
    pa = (pa + 1) & m;
  }
  fft_natural(n + 1, z, -1);
  if (c->cfg.fft1_direction > 0) {
    out[0] = z[2 * N]; out[1] = z[0];
    for (int k = 1; k < N; k++) { out[2 * k] = z[2 * k + 1]; out[2 * k + 1] = z[2 * k]; }
  } else {
    for (int k = 1; k < N; k++) { out[2 * (N - k)] = z[2 * k]; out[2 * (N - k) + 1] = z[2 * k + 1]; }
    out[0] = z[2 * N]; out[1] = z[2 * N];
  }
}

static void fft1_one(lro_ctx *c, int timf1p_ref, float *out)
{
  if (c->cfg.timf1_real_input) { fft1_one_real(c, timf1p_ref, out); return; }
  int N = c->N1, n = c->cfg.fft1_n, nn = N / 2;
  const int dword = c->cfg.timf1_dword_input != 0, esz = dword ? 4 : 2;   /* fft1.c:420 / :526 */
  const int32_t *t32 = (const int32_t *)c->timf1;
  int m = c->timf1_bytemask / esz;
  /* frames of C channels {I0,Q0,I1,Q1,..} (fft1win_dif_chan, fft1.c:2041-2055: p0 = ref/2 - 4 I1, channel stride 4 shorts): this
     context's channel is cfg.timf1_channel_index; one channel: C = 1 */
  const int C = c->cfg.timf1_frame_channels > 1 ? c->cfg.timf1_frame_channels : 1, st = 2 * C;
  int p0 = timf1p_ref / esz; p0 = (p0 - c->I1 * st + 2 * (C > 1 ? c->cfg.timf1_channel_index : 0) + 4 * (m + 1)) & m;
  int pa = p0, pb = (pa + N * C) & m, pa1 = pa, pb1 = pb;
  const int sh = C > 1 ? 0 : c->cfg.sample_shift;
  if (sh < 0) { pa1 = (pa + 2 * sh + m + 1) & m; pb1 = (pa1 + N) & m; }            /* fft1.c:472-476 */
  else if (sh > 0) { pa = (pa - 2 * sh + m + 1) & m; pb = (pb - 2 * sh + m + 1) & m; } /* fft1.c:478-482 */
  float *z = c->tmp;
  int win = c->cfg.fft1_sinpow != 0;
  for (int ia = 0; ia < nn; ia++) {           /* window, negate Q, natural order; I from pa/pb, Q from pa1/pb1 */
    float wa = win ? c->fft1_window[2 * ia] : 1.0f, wb = win ? c->fft1_window[2 * ia + 1] : 1.0f;
    float t1, t2, t3, t4;
    if (dword) { t1 = t32[pa] * wa; t2 = t32[pa1 + 1] * wa; t3 = t32[pb] * wb; t4 = t32[pb1 + 1] * wb; }
    else { t1 = c->timf1[pa] * wa; t2 = c->timf1[pa1 + 1] * wa; t3 = c->timf1[pb] * wb; t4 = c->timf1[pb1 + 1] * wb; }
    z[2 * ia] = t1; z[2 * ia + 1] = -t2; z[2 * (ia + nn)] = t3; z[2 * (ia + nn) + 1] = -t4;
    pa = (pa + st) & m; pb = (pb + st) & m; pa1 = (pa1 + st) & m; pb1 = (pb1 + st) & m;
  }
  dif_stages(N, n, z, c->fft1tab, +1, 2);
  for (unsigned i = 0; i < (unsigned)N; i++) {   /* bit reversal + half swap: make_permute(1,..) fft0.c:1147-1207 */
    unsigned k = (bitrev(i, n) + nn) & (N - 1);
    out[2 * k] = z[2 * i]; out[2 * k + 1] = z[2 * i + 1];
  }
  if (c->fft1_foldcorr) {                       /* I/Q mirror-image cancellation, CALIQ (fft1.c:3607-3658), m = 1, pc = 0 */
    const float *fc = c->fft1_foldcorr;
    int ib = 1, ic = N - 1, ia;
    if (c->cfg.fft1_direction > 0) {
      for (ia = 1; ia < nn; ia++, ib++, ic--) {
        float t1 = out[2 * ib] * fc[2 * ia] - out[2 * ib + 1] * fc[2 * ia + 1];
        float t2 = out[2 * ib] * fc[2 * ia + 1] + out[2 * ib + 1] * fc[2 * ia];
        out[2 * ib] -= out[2 * ic] * fc[2 * ic] + out[2 * ic + 1] * fc[2 * ic + 1];
        out[2 * ib + 1] -= out[2 * ic] * fc[2 * ic + 1] - out[2 * ic + 1] * fc[2 * ic];
        out[2 * ic] -= t1;
        out[2 * ic + 1] += t2;
      }
    } else {
      for (ia = 1; ia < nn; ia++, ib++, ic--) {
        float t1 = out[2 * ic] - out[2 * ib] * fc[2 * ia] + out[2 * ib + 1] * fc[2 * ia + 1];
        float t2 = out[2 * ic + 1] + out[2 * ib] * fc[2 * ia + 1] + out[2 * ib + 1] * fc[2 * ia];
        float t3 = out[2 * ib] - out[2 * ic] * fc[2 * ic] - out[2 * ic + 1] * fc[2 * ic + 1];
        float t4 = out[2 * ib + 1] - out[2 * ic] * fc[2 * ic + 1] + out[2 * ic + 1] * fc[2 * ic];
        out[2 * ib + 1] = t1; out[2 * ib] = t2; out[2 * ic + 1] = t3; out[2 * ic] = t4;
      }
      float t = out[2 * ib]; out[2 * ib] = out[2 * ib + 1]; out[2 * ib + 1] = t;
      t = out[0]; out[0] = out[1]; out[1] = t;
    }
  } else if (c->cfg.fft1_direction < 0) {       /* fft1.c:3660-3679 with fft1_first_sym_point = 0 */
    for (int ib = 1, ic = N - 1; ib < nn; ib++, ic--) {
      float t1 = out[2 * ic], t2 = out[2 * ic + 1];
      out[2 * ic + 1] = out[2 * ib]; out[2 * ic] = out[2 * ib + 1];
      out[2 * ib + 1] = t1; out[2 * ib] = t2;
    }
    float t = out[2 * nn]; out[2 * nn] = out[2 * nn + 1]; out[2 * nn + 1] = t;
    t = out[0]; out[0] = out[1]; out[1] = t;
  }
  if (c->ch2_set && (c->ch2_c1 != 1.0F || c->ch2_c2 != 0.0F)) {   /* phasing of the second channel, fft1.c:4064-4080 (m = 0) */
    for (int i = 0; i < N; i++) {
      float t1 = out[2 * i], t2 = out[2 * i + 1];
      out[2 * i] = t1 * c->ch2_c1 + t2 * c->ch2_c2;
      out[2 * i + 1] = t2 * c->ch2_c1 - t1 * c->ch2_c2;
    }
  }
}

int lro_fft1_b(lro_ctx *c, int handle, int timf1p_ref, int fft1_pa, int batch)
{
  (void)handle;                                /* gpu_handle_number: which worker thread calls (fft1.c:3302); no meaning on the CPU */
  int blockbytes = c->M1 * (c->cfg.timf1_dword_input ? 8 : 4) * (c->cfg.timf1_frame_channels > 1 ? c->cfg.timf1_frame_channels : 1);
  for (int b = 0; b < batch; b++) {
    int nb = ((fft1_pa / (2 * c->N1)) + b) & c->fft1n_mask;
    float *out = c->fft1_float + (size_t)nb * 2 * c->N1;
    fft1_one(c, (timf1p_ref + b * blockbytes) & c->timf1_bytemask, out);
    /* filter correction of fft1_c (fft1.c:4119-4127), applied here like the HIP path does */
    const float *fc = c->fft1_filtercorr;
    for (int i = 0; i < c->N1; i++) {
      float t1 = out[2 * i] * fc[2 * i] - out[2 * i + 1] * fc[2 * i + 1];
      out[2 * i + 1] = out[2 * i + 1] * fc[2 * i] + out[2 * i] * fc[2 * i + 1];
      out[2 * i] = t1;
    }
  }
  return LRH_OK;
}

/* NET_RXOUT_FFT1 payload (wcw.c:1024-1043, network.c:383-388): the transforms as fft1_b leaves them, i.e. without the filter correction
   that lro_fft1_b folds into its store; recomputed from the timf1 ring like lrh_export_fft1_net does */
int lro_export_fft1_net(lro_ctx *c, f32 *dst, int timf1p_ref, int batch)
{
  if (!c || !dst || batch < 1) return LRH_EINVAL;
  int blockbytes = c->M1 * (c->cfg.timf1_dword_input ? 8 : 4) * (c->cfg.timf1_frame_channels > 1 ? c->cfg.timf1_frame_channels : 1);   /* whole frames, as lro_fft1_b steps */
  float *t = malloc(sizeof(float) * 2 * c->N1);
  for (int b = 0; b < batch; b++) { fft1_one(c, (timf1p_ref + b * blockbytes) & c->timf1_bytemask, t); out_f32(dst + (size_t)b * 2 * c->N1, t, 2 * (size_t)c->N1); }
  free(t);
  return LRH_OK;
}

/* new_fft1_averages, wide_graph.c:1003-1032 */
static void new_fft1_averages(lro_ctx *c, int ptr, int ia, int ib)
{
  int N = c->N1, a2 = c->cfg.fft_avg2num;
  int p0 = (ptr - (a2 - 1) * N + c->cfg.fft1_sumsq_bufsize) & c->fft1_sumsq_mask;
  for (int i = ia; i <= ib; i++) c->fft1_slowsum[i] = c->fft1_sumsq[p0 + i];
  p0 = (p0 + N) & c->fft1_sumsq_mask;
  for (int m = 1; m < a2; m++) {
    for (int i = ia; i <= ib; i++) { c->fft1_slowsum[i] += c->fft1_sumsq[p0 + i]; if (c->fft1_slowsum[i] < FFT1_SMALL) c->fft1_slowsum[i] = FFT1_SMALL; }
    p0 = (p0 + N) & c->fft1_sumsq_mask;
  }
}

/* update_fft1_slowsum, fft1.c:4526-4605 (no correlation, change_fft1_flag clear) */
static void update_fft1_slowsum(lro_ctx *c, lrh_ptrs *p)
{
  int N = c->N1, first = 0, last = N - 1;
  int pa = p->fft1_sumsq_pa;
  int pb = (pa - c->cfg.fft_avg2num * N + c->cfg.fft1_sumsq_bufsize) & c->fft1_sumsq_mask;
  if (p->fft1_sumsq_recalc == last) p->fft1_sumsq_recalc = first;
  int ia = p->fft1_sumsq_recalc;
  p->fft1_sumsq_recalc += c->cfg.wg_xpoints / c->cfg.slowsum_fresh_recalc;
  if (p->fft1_sumsq_recalc > last) p->fft1_sumsq_recalc = last;
  new_fft1_averages(c, pa, ia, p->fft1_sumsq_recalc);
  for (int i = first; i < ia; i++) {
    c->fft1_slowsum[i] += c->fft1_sumsq[pa + i] - c->fft1_sumsq[pb + i];
    if (c->fft1_slowsum[i] < FFT1_SMALL) c->fft1_slowsum[i] = FFT1_SMALL;
  }
  for (int i = p->fft1_sumsq_recalc + 1; i <= last; i++) {
    c->fft1_slowsum[i] += c->fft1_sumsq[pa + i] - c->fft1_sumsq[pb + i];
    if (c->fft1_slowsum[i] < FFT1_SMALL) c->fft1_slowsum[i] = FFT1_SMALL;
  }
}

/* ---- correlation spectrum (fft1_correlation_flag == 1): fft1_c's fft1_corrsum (fft1.c:4146-4150, 4189-4193) and update_fft1_slowsum's
   fft1_slowcorr / fft1_slowcorr_tot (fft1.c:4584-4603, new_fft1_averages wide_graph.c:1033-1050), one channel per context with the
   transforms exchanged through LRH_X_SPEC (include/linrad_hip.h) */
int lro_set_correlation(lro_ctx *c, int on)
{
  LRO_F32_ONLY(c);
  if (!c || c->cfg.blanker_channels != 2) return LRH_ESTATE;
  free(c->xspec); free(c->fft1_corrsum); free(c->fft1_slowcorr); free(c->fft1_slowcorr_tot);
  c->xspec = NULL; c->fft1_corrsum = NULL; c->fft1_slowcorr = NULL; c->fft1_slowcorr_tot = NULL;
  c->corr_on = 0; c->slowcorr_tot_avgnum = 0;
  if (!on) return LRH_OK;
  c->xspec = calloc((size_t)4 * c->cfg.max_batch * c->N1, sizeof(float)); c->fft1_corrsum = calloc((size_t)2 * c->cfg.fft1_sumsq_bufsize, sizeof(float));
  c->fft1_slowcorr = calloc((size_t)2 * c->N1, sizeof(float)); c->fft1_slowcorr_tot = calloc((size_t)2 * c->N1, sizeof(double));
  if (!c->xspec || !c->fft1_corrsum || !c->fft1_slowcorr || !c->fft1_slowcorr_tot) return LRH_ENOMEM;
  c->corr_on = 1;
  return LRH_OK;
}
int lro_get_slowcorr_tot_avgnum(lro_ctx *c, int *n) { if (!c || !n) return LRH_EINVAL; *n = c->slowcorr_tot_avgnum; return LRH_OK; }
int lro_fft1_corr_begin(lro_ctx *c, const lrh_ptrs *at, int batch, size_t *count)
{
  if (!c || !at || !count || batch < 1 || batch > c->cfg.max_batch) return LRH_EINVAL;
  if (!c->corr_on) return LRH_ESTATE;
  const int N = c->N1;
  float *slot = c->xspec + (size_t)(c->cfg.timf1_channel_index & 1) * batch * 2 * N;
  for (int b = 0; b < batch; b++) memcpy(slot + (size_t)b * 2 * N, c->fft1_float + (size_t)((at->fft1_nb + b) & c->fft1n_mask) * 2 * N, sizeof(float) * 2 * N);
  *count = (size_t)batch * 2 * N;
  return LRH_OK;
}
int lro_fft1_corr_finish(lro_ctx *c, const lrh_ptrs *at, int batch)
{
  if (!c || !at || batch < 1 || batch > c->cfg.max_batch) return LRH_EINVAL;
  if (!c->corr_on) return LRH_ESTATE;
  const int N = c->N1, last = N - 1, avg2 = c->cfg.fft_avg2num, bufsize = c->cfg.fft1_sumsq_bufsize, mask = c->fft1_sumsq_mask;
  int counter = at->fft1_sumsq_counter, pa = at->fft1_sumsq_pa, recalc = at->fft1_sumsq_recalc;
  for (int b = 0; b < batch; b++) {
    const float *x = c->xspec + (size_t)b * 2 * N, *y = c->xspec + ((size_t)batch + b) * 2 * N;
    float *cs = c->fft1_corrsum + 2 * (size_t)pa;
    for (int i = 0; i < N; i++) {
      const float re = 2 * (x[2 * i] * y[2 * i] + x[2 * i + 1] * y[2 * i + 1]), im = 2 * (x[2 * i + 1] * y[2 * i] - x[2 * i] * y[2 * i + 1]);
      if (counter == 0) { cs[2 * i] = re; cs[2 * i + 1] = im; } else { cs[2 * i] += re; cs[2 * i + 1] += im; }
    }
    if (++counter < c->cfg.fft_avg1num) continue;
    counter = 0;
    /* update_fft1_slowsum's walk for the correlation sums: the same refresh window as fft1_slowsum */
    const int pb = (pa - avg2 * N + bufsize) & mask;
    if (recalc == last) recalc = 0;
    const int ia = recalc;
    recalc += c->cfg.wg_xpoints / c->cfg.slowsum_fresh_recalc;
    if (recalc > last) recalc = last;
    { int p0 = (pa - (avg2 - 1) * N + bufsize) & mask;
      for (int i = ia; i <= recalc; i++) { c->fft1_slowcorr[2 * i] = c->fft1_corrsum[2 * (p0 + i)]; c->fft1_slowcorr[2 * i + 1] = c->fft1_corrsum[2 * (p0 + i) + 1]; }
      p0 = (p0 + N) & mask;
      for (int m = 1; m < avg2; m++) {
        for (int i = ia; i <= recalc; i++) { c->fft1_slowcorr[2 * i] += c->fft1_corrsum[2 * (p0 + i)]; c->fft1_slowcorr[2 * i + 1] += c->fft1_corrsum[2 * (p0 + i) + 1]; }
        p0 = (p0 + N) & mask;
      } }
    for (int i = 0; i < N; i++) {
      if (i >= ia && i <= recalc) continue;
      c->fft1_slowcorr[2 * i] += c->fft1_corrsum[2 * (pa + i)] - c->fft1_corrsum[2 * (pb + i)];
      c->fft1_slowcorr[2 * i + 1] += c->fft1_corrsum[2 * (pa + i) + 1] - c->fft1_corrsum[2 * (pb + i) + 1];
    }
    for (int i = 0; i < N; i++) { c->fft1_slowcorr_tot[2 * i] += c->fft1_corrsum[2 * (pa + i)]; c->fft1_slowcorr_tot[2 * i + 1] += c->fft1_corrsum[2 * (pa + i) + 1]; }
    c->slowcorr_tot_avgnum += c->cfg.fft_avg1num;
    pa = (pa + N) & mask;
  }
  return LRH_OK;
}

static void lro_spur_hook(lro_ctx *c, int na);      /* eliminate_spurs, defined with the spur tracking at the end of this file */
static void spur_search_row(lro_ctx *c, const float *pwra);
/* fft1_c, fft1.c:4085-4201 + 4507-4523 (1 channel, fft1afc_flag <= 0); the complex multiply already done in lro_fft1_b */
int lro_fft1_c(lro_ctx *c, lrh_ptrs *p, int batch)
{
  int N = c->N1;
  for (int b = 0; b < batch; b++) {
    const float *z = c->fft1_float + (size_t)p->fft1_nb * 2 * N;
    float *sum = c->fft1_sumsq + p->fft1_sumsq_pa;
    if (c->spurs && !c->cfg.second_fft_enable) {         /* the last step of fft1 when AFC runs from fft1 (fft1afc_flag > 0, fft1.c:4196-4244, 4428-4460): */
      lro_spur_hook(c, p->fft1_nb);                      /* eliminate_spurs on the new transform, then the search spectrum's row of its powers */
      float *pw = c->tmp;
      for (int i = 0; i < N; i++) pw[i] = z[2 * i] * z[2 * i] + z[2 * i + 1] * z[2 * i + 1];
      spur_search_row(c, pw);
    }
    if (p->fft1_sumsq_counter == 0) for (int i = 0; i < N; i++) sum[i] = z[2 * i] * z[2 * i] + z[2 * i + 1] * z[2 * i + 1];
    else                            for (int i = 0; i < N; i++) sum[i] += z[2 * i] * z[2 * i] + z[2 * i + 1] * z[2 * i + 1];
    p->fft1_sumsq_counter++;
    if (p->fft1_sumsq_counter >= c->cfg.fft_avg1num) {
      p->fft1_sumsq_counter = 0;
      update_fft1_slowsum(c, p);
      if (c->cfg.second_fft_enable) p->fft1_liminfo_cnt++;      /* fft1.c:4515-4518 */
      p->fft1_sumsq_pa = (p->fft1_sumsq_pa + N) & c->fft1_sumsq_mask;
    }
    p->fft1_nb = (p->fft1_nb + 1) & c->fft1n_mask;
    p->fft1_pb = p->fft1_nb * 2 * N;
  }
  return LRH_OK;
}

/* ------------------------------------------------------------------ timf2 */

/* make_timf2 (timf2.c:31-75,127-128,205-207) + fft1back_one (timf2.c:689-967: DFT with e^{-j} kernel on the
   weak and the strong stream at once) + fft1back_fp_finish (timf2.c:970-1064) */
int lro_make_timf2(lro_ctx *c, lrh_ptrs *p, int batch)
{
  int N = c->N1, n = c->cfg.fft1_n;
  float ampfac = (float)(1.0 / (1 << c->cfg.bckfft_att_n));
  for (int b = 0; b < batch; b++) {
    const float *x = c->fft1_float + p->fft1_px;
    float *s = c->tmp;                       /* {wRe,wIm,sRe,sIm} per bin */
    int lowlevel = 0;
    for (int i = 0; i < N; i++) {
      if (c->liminfo[i] == 0) { lowlevel++; s[4 * i] = x[2 * i]; s[4 * i + 1] = x[2 * i + 1]; s[4 * i + 2] = 0; s[4 * i + 3] = 0; }
      else { s[4 * i] = 0; s[4 * i + 1] = 0; s[4 * i + 2] = x[2 * i]; s[4 * i + 3] = x[2 * i + 1]; }
    }
    dif_stages(N, n, s, c->fft1tab, -1, 4); dif_stages(N, n, s + 2, c->fft1tab, -1, 4);
    bitrev_inplace(N, n, s, 4); bitrev_inplace(N, n, s + 2, 4);
    int p0 = p->timf2_pa, pp = p0 / 4, m = c->timf2_mask;
    float *tf = c->timf2_float, *pw = c->timf2_pwr;
    if (c->I1 == 0) {                         /* timf2.c:986-1000 */
      for (int i = 0; i < N; i++) {
        int q = (p0 + 4 * i) & m;
        for (int k = 0; k < 4; k++) tf[q + k] = ampfac * s[4 * i + k];
        pw[q / 4] = tf[q] * tf[q] + tf[q + 1] * tf[q + 1];
      }
    } else if (c->I1 == N / 2) {              /* sin^2: overlap-add, timf2.c:1003-1026 */
      int k = N / 2;
      for (int i = 0; i < k; i++) {
        for (int j = 0; j < 4; j++) tf[p0 + j] += ampfac * s[4 * i + j];
        pw[pp] = tf[p0] * tf[p0] + tf[p0 + 1] * tf[p0 + 1];
        pp++; p0 += 4;
      }
      p0 &= m;
      for (int i = 4 * k; i < 4 * N; i++) { tf[p0] = s[i] * ampfac; p0++; }
    } else {                                  /* other windows: centre part x inverted window, timf2.c:1031-1061 */
      int ia = c->I1 / 2, ib = N / 2, kk = ia * 4;
      for (int i = ia; i < ib; i++) {
        float t1 = c->fft1_inverted_window[i] * ampfac;
        for (int j = 0; j < 4; j++) tf[p0 + j] = t1 * s[kk + j];
        pw[p0 >> 2] = tf[p0] * tf[p0] + tf[p0 + 1] * tf[p0 + 1];
        p0 = (p0 + 4) & m; kk += 4;
      }
      for (int i = ib; i > ia; i--) {
        float t1 = c->fft1_inverted_window[i] * ampfac;
        for (int j = 0; j < 4; j++) tf[p0 + j] = t1 * s[kk + j];
        pw[p0 >> 2] = tf[p0] * tf[p0] + tf[p0 + 1] * tf[p0 + 1];
        p0 = (p0 + 4) & m; kk += 4;
      }
    }
    p->fft1_px = (p->fft1_px + 2 * N) & c->fft1_mask;
    p->fft1_nx = (p->fft1_nx + 1) & c->fft1n_mask;
    p->fft1_lowlevel_points = lowlevel;
    p->fft1_lowlevel_fraction = 0.02 * (49 * p->fft1_lowlevel_fraction + lowlevel / ((float)(N - 1)));
    p->timf2_pa = (p->timf2_pa + 4 * c->M1) & m;
  }
  return LRH_OK;
}

/* ------------------------------------------------------------------ blanker */

/* first_noise_blanker, blank1.c:684-715 (range, rate limit), 1003-1087 (stupid blanker, 1 channel float),
   1458-1603 (pointers, every-4th-sample noise statistics, threshold update). Clever blanker needs a pulse
   calibration and is forced off without one (hires_graph.c:1196). */
static int blanker_update(lro_ctx *c, lrh_ptrs *p, float t1, int do_update, float llf, int chans);

/* two coupled channels, see include/linrad_hip.h: the own power of the span the next call scans goes to the exchange buffer */
int lro_blanker_begin(lro_ctx *c, const lrh_ptrs *p, int *count)
{
  if (c->cfg.blanker_channels != 2) return LRH_ESTATE;
  const int mask = c->timf2pow_mask, pbeg = p->timf2p_fit;
  int pend = (p->timf2_pa / 4 - c->cfg.blnfit_range + mask) & mask;
  pend &= 0xfffffffc;
  *count = 0; c->x_count = -1;
  if (((pend - pbeg + 1 + mask) & mask) < c->cfg.blanker_min_points) return LRH_OK;
  c->x_pbeg = pbeg; c->x_span = (pend - pbeg) & mask; c->xw_count = 0;
  /* linear blanker: its search and fits read (and rewrite) up to blnfit_range samples beyond the span, and +-blnfit_range of both channels' samples */
  const int R = c->clever_on ? c->cfg.blnfit_range : 0;
  c->x_count = c->x_span + R;
  for (int q = 1; q <= c->x_count; q++) c->xbuf[q - 1] = c->timf2_pwr[(pbeg + q) & mask];
  if (c->clever_on) {
    const int nw = c->x_span + 2 * R + 1, ci = c->cfg.timf1_channel_index & 1;
    float *slot = c->xweak + (size_t)ci * 2 * nw;
    for (int i = 0; i < nw; i++) { const int pos = (pbeg - R + i) & mask; slot[2 * i] = c->timf2_float[4 * pos]; slot[2 * i + 1] = c->timf2_float[4 * pos + 1]; }
    c->xw_count = 2 * nw;
  }
  *count = c->x_count;
  return LRH_OK;
}
int lro_blanker_weak_span(lro_ctx *c, size_t *count)
{
  if (c->cfg.blanker_channels != 2 || !count) return LRH_ESTATE;
  *count = c->x_count > 0 ? (size_t)c->xw_count : 0;
  return LRH_OK;
}
int lro_blanker_finish(lro_ctx *c, lrh_ptrs *p)
{
  if (c->cfg.blanker_channels != 2 || !c->fin_pending) return LRH_ESTATE;
  c->fin_pending = 0;
  c->bs.timf2_despiked_pwrinc[0] += c->xstat[0];          /* blank1.c:1538-1540 */
  c->bs.timf2_despiked_pwrinc[1] += c->xstat[1];
  return blanker_update(c, p, c->xstat[0] + c->xstat[1], c->fin_do_update, c->fin_llf, 2);
}
static size_t exchange_cap(const lro_ctx *c, int which)
{
  if (which == LRH_X_POL) return c->xpol ? 4 * (size_t)c->cfg.max_fft3n * c->Nm2 : 0;
  if (which == LRH_X_WEAK) return c->xweak ? 4 * (size_t)c->cfg.timf2pow_size : 0;
  return which == LRH_X_PWR ? (size_t)c->cfg.timf2pow_size : which == LRH_X_STAT ? 2 : 4 * (size_t)c->cfg.max_fft2n * c->N2;
}
int lro_exchange_ptr(lro_ctx *c, int which, void **ptr)
{
  if (which == LRH_X_POL && ptr) { if (!c->xpol) return LRH_ESTATE; *ptr = c->xpol; return LRH_OK; }
  if (which == LRH_X_SPEC && ptr) { if (!c->xspec) return LRH_ESTATE; *ptr = c->xspec; return LRH_OK; }
  if (c->cfg.blanker_channels != 2 || !ptr) return LRH_ESTATE;
  if (which == LRH_X_WEAK) { if (!c->xweak) return LRH_ESTATE; *ptr = c->xweak; return LRH_OK; }
  if (which != LRH_X_PWR && which != LRH_X_STAT && which != LRH_X_BINS && which != LRH_X_POL) return LRH_EINVAL;
  if (which == LRH_X_POL && !c->xpol) return LRH_ESTATE;
  *ptr = which == LRH_X_PWR ? (void *)c->xbuf : which == LRH_X_STAT ? (void *)c->xstat : which == LRH_X_BINS ? (void *)c->xbins : (void *)c->xpol;
  return LRH_OK;
}
int lro_exchange_read(lro_ctx *c, int which, f32 *dst, size_t off, size_t count)
{
  LRO_F32_ONLY(c);
  void *q; int rc = lro_exchange_ptr(c, which, &q); if (rc) return rc;
  if (off + count > exchange_cap(c, which)) return LRH_EINVAL;
  memcpy(dst, (float *)q + off, 4 * count); return LRH_OK;
}
int lro_exchange_write(lro_ctx *c, int which, const f32 *src, size_t off, size_t count)
{
  LRO_F32_ONLY(c);
  void *q; int rc = lro_exchange_ptr(c, which, &q); if (rc) return rc;
  if (off + count > exchange_cap(c, which)) return LRH_EINVAL;
  memcpy((float *)q + off, src, 4 * count); return LRH_OK;
}

/* ---- linear ("clever") blanker: blank1.c:36-232 (subtract_onechan_pulse), :615-682 (set_flag), :765-1003 (search loop) ---- */
int lro_set_blanker_tables(lro_ctx *c, const lrh_blanker_tables *t)
{
  if (t) LRO_F32_ONLY(c);
  free(c->bt_refpulse); free(c->bt_phasefunc); free(c->bt_pulindex); free(c->blanker_flag); free(c->xweak); free(c->tf_partner);
  c->bt_refpulse = c->bt_phasefunc = NULL; c->bt_pulindex = NULL; c->blanker_flag = NULL; c->clever_on = 0; c->xweak = c->tf_partner = NULL;
  if (!t) return LRH_OK;
  int rs = t->refpul_size, pw = c->cfg.blanker_pulsewidth;
  if (t->clever_bln_mode < 1 || t->clever_bln_mode > 2 || rs < 4 || rs > 256 || (rs & (rs - 1)) || t->largest_blnfit < 0 ||
      t->largest_blnfit >= LRH_BLN_INFO_SIZE || !t->refpulse || !t->phasefunc || !t->pulindex || pw < 1 || 2 * pw >= rs) return LRH_EINVAL;
  for (int i = 0; i <= t->largest_blnfit; i++)
    if (t->bln[i].size < 4 || t->bln[i].size > rs || (t->bln[i].size & 1) || (i && t->bln[i].size <= t->bln[i - 1].size)) return LRH_EINVAL;
  if (c->cfg.blnfit_range != t->bln[t->largest_blnfit].size / 2 + pw) return LRH_EINVAL;      /* buf.c:2057 */
  for (int i = 0; i < LRH_MAX_REFPULSES; i++) if (t->pulindex[i] < 0 || t->pulindex[i] >= LRH_MAX_REFPULSES) return LRH_EINVAL;
  c->bt = *t;
  size_t nr = (size_t)2 * LRH_MAX_REFPULSES * rs;
  c->bt_refpulse = malloc(4 * nr); c->bt_phasefunc = malloc(8 * rs); c->bt_pulindex = malloc(4 * LRH_MAX_REFPULSES);
  c->blanker_flag = calloc(1, c->cfg.timf2pow_size);
  if (c->cfg.blanker_channels == 2) { c->xweak = calloc(4, 4 * (size_t)c->cfg.timf2pow_size); c->tf_partner = calloc(4, 2 * (size_t)c->cfg.timf2pow_size); }
  memcpy(c->bt_refpulse, t->refpulse, 4 * nr); memcpy(c->bt_phasefunc, t->phasefunc, 8 * rs); memcpy(c->bt_pulindex, t->pulindex, 4 * LRH_MAX_REFPULSES);
  c->bt.refpulse = c->bt_refpulse; c->bt.phasefunc = c->bt_phasefunc; c->bt.pulindex = c->bt_pulindex;
  c->bs.clever_bln_limit = t->clever_bln_limit;
  c->amp_factor = t->liminfo_amplitude_factor;
  c->clever_on = 1;
  return LRH_OK;
}

/* blank1.c:615-682: flag +-pulsewidth around p_max and on outwards for as long as the power keeps falling */
static void clever_set_flag(lro_ctx *c, const float *pw, unsigned char value, int p_max, int pbeg, int pend)
{
  const int mask = c->timf2pow_mask;
  unsigned char *fl = c->blanker_flag;
  fl[p_max] = value;
  int pa = p_max, pb = p_max;
  for (int i = 0; i < c->cfg.blanker_pulsewidth; i++) { pb = (pb + mask) & mask; pa = (pa + 1) & mask; fl[pa] = value; fl[pb] = value; }
  int p0 = pb; pb = (pb + mask) & mask;
  if (!(((pb - pbeg + mask) & mask) > mask / 2))
    while (pw[pb] < pw[p0] && pb != pbeg) { fl[pb] = value; p0 = pb; pb = (pb + mask) & mask; }
  p0 = pa; pa = (pa + 1) & mask;
  if (!(((pend - pa + mask) & mask) > mask / 2))
    while (pw[pa] < pw[p0] && pa != pend) { p0 = pa; fl[pa] = value; pa = (pa + 1) & mask; }
}

/* blank1.c:36-232.  Quirk kept: the restore after a failed subtraction (:190-208) adds the cross terms with the signs of the
   subtraction, so the samples are NOT returned to their old values exactly when blanker_phase_c2 != 0. */
static float clever_subtract(lro_ctx *c, int p_max, int sub_size)
{
  const int mask = c->timf2pow_mask, rs = c->bt.refpul_size, pwid = c->cfg.blanker_pulsewidth;
  float *tf = c->timf2_float, *pw = c->timf2_pwr;
  const f32 *phf = c->bt_phasefunc, *rp = c->bt_refpulse;
  float in[2 * 257];
  int k = rs - 2 * pwid, i = 0;
  for (int q = p_max - pwid; q <= p_max + pwid; q++) {
    int pa = 4 * (q & mask);
    float t1 = tf[pa], t2 = tf[pa + 1], t3 = phf[k], t4 = phf[k + 1];
    in[i] = t1 * t3 + t2 * t4; in[i + 1] = t2 * t3 - t1 * t4; k += 2; i += 2;
  }
  int imax = pwid;
  float c1 = 0, c2 = 0;
  for (i = imax - 1; i <= imax + 1; i++) { float t1 = in[2 * i], t2 = in[2 * i + 1], t3 = sqrt(t1 * t1 + t2 * t2); c1 += t3 * t1; c2 += t3 * t2; }
  float t1 = c1 * c1 + c2 * c2;
  if (t1 < 32) return -1;
  t1 = sqrt(t1); c1 /= t1; c2 /= t1;
  int wid = 2 * pwid;
  float t3 = 0, t4 = 0;
  for (i = 0; i <= wid; i++) {
    float a = in[2 * i], b = in[2 * i + 1];
    in[2 * i] = c1 * a + c2 * b; in[2 * i + 1] = c1 * b - c2 * a;
    t3 += in[2 * i] * in[2 * i]; t4 += in[2 * i + 1] * in[2 * i + 1];
  }
  if (t4 > 0.25 * t3) return -1.;
  t4 = in[2 * imax - 2] - in[2 * imax + 2];
  t3 = 2 * (in[2 * imax - 2] + in[2 * imax + 2] - 2 * in[2 * imax]);
  if (t3 == 0) return -2.;
  t4 /= t3;
  if (t4 < 0) t4 = -sqrt(0.5) * sqrt(-t4); else t4 = sqrt(0.5) * sqrt(t4);
  int j = LRH_MAX_REFPULSES * (t4 + 0.5) + 0.5;
  if (j < 0) j = 0;
  if (j >= LRH_MAX_REFPULSES) j = LRH_MAX_REFPULSES - 1;
  int m = 2 * c->bt_pulindex[j] * rs;
  c1 *= in[2 * imax] * c->amp_factor; c2 *= in[2 * imax] * c->amp_factor;
  t3 = 0; t4 = 0;
  k = rs - sub_size;
  for (int q = p_max - sub_size / 2; q <= p_max + sub_size / 2; q++) {
    int pos = q & mask, p0 = 4 * pos;
    float r1 = rp[m + k], r2 = rp[m + k + 1];
    float re = tf[p0] - c1 * r1 + c2 * r2, im = tf[p0 + 1] - c1 * r2 - c2 * r1;
    tf[p0] = re; tf[p0 + 1] = im;
    float pn = re * re + im * im;
    t3 += pw[pos]; pw[pos] = pn; k += 2; t4 += pn;
  }
  float retval = t4 / t3;
  if (retval > 0.5) {
    k = rs - sub_size;
    for (int q = p_max - sub_size / 2; q <= p_max + sub_size / 2; q++) {
      int pos = q & mask, p0 = 4 * pos;
      float r1 = rp[m + k], r2 = rp[m + k + 1];
      float re = tf[p0] + c1 * r1 + c2 * r2, im = tf[p0 + 1] + c1 * r2 - c2 * r1;
      tf[p0] = re; tf[p0 + 1] = im; pw[pos] = re * re + im * im; k += 2;
    }
    return -5;
  }
  return retval;
}

/* Two coupled channels, blank1.c:984-992: get_pulse_pol (:433-562) finds the polarisation of the pulse from the two channels'
   samples around the peak, transform_timf2_pol (:565-609) forms the one signal that carries it, subtract_twochan_pulse (:232-430)
   fits the reference pulse to that and takes its share out of both channels.  X is channel 0, Y channel 1; this context owns one
   of them (ring stride 4) and holds the partner's samples of the exchanged span (stride 2).  pw is the ring of summed powers the
   search runs on; the own channel's power ring follows the own samples.  Both contexts of a pair compute the same thing. */
static float clever_subtract2(lro_ctx *c, float *pw, int p_max, int sub_size)
{
  const int mask = c->timf2pow_mask, rs = c->bt.refpul_size, pwid = c->cfg.blanker_pulsewidth, ci = c->cfg.timf1_channel_index & 1;
  float *chp[2]; int chs[2];
  chp[ci] = c->timf2_float; chs[ci] = 4; chp[1 - ci] = c->tf_partner; chs[1 - ci] = 2;
  const f32 *phf = c->bt_phasefunc, *rp = c->bt_refpulse;
  float x2 = 0, y2 = 0, re_xy = 0, im_xy = 0;
  for (int q = p_max - pwid; q <= p_max + pwid; q++) {
    const int pos = q & mask;
    const float re_x = chp[0][chs[0] * pos], im_x = chp[0][chs[0] * pos + 1], re_y = chp[1][chs[1] * pos], im_y = chp[1][chs[1] * pos + 1];
    x2 += re_x * re_x + im_x * im_x; y2 += re_y * re_y + im_y * im_y;
    re_xy += re_x * re_y + im_x * im_y; im_xy += im_x * re_y - re_x * im_y;
  }
  float t1 = x2 + y2;
  x2 /= t1; y2 /= t1; re_xy /= t1; im_xy /= t1;
  float t2 = re_xy * re_xy + im_xy * im_xy;
  const float noi2 = x2 * y2 - t2;
  if (noi2 > 0.15) return -1;
  const float x2s = x2 - noi2, y2s = y2 - noi2;
  float pc1, pc2, pc3;
  if (x2s > 0) {
    pc1 = sqrt(x2s);
    if (y2s > 0 && t2 > 0) {
      const float sina = sqrt(y2s);
      pc2 = sina * re_xy / sqrt(t2); pc3 = sina * im_xy / sqrt(t2);
      t1 = sqrt(pc1 * pc1 + pc2 * pc2 + pc3 * pc3);
      pc1 /= t1; pc2 /= t1; pc3 /= t1;
    } else { if (x2 > y2) { pc1 = 1; pc2 = 0; } else { pc1 = 0; pc2 = 1; } pc3 = 0; }
  } else { pc1 = 0; pc2 = 1; pc3 = 0; }
  float in[2 * 257];
  int k = rs - 2 * pwid, i = 0;
  for (int q = p_max - pwid; q <= p_max + pwid; q++) {
    const int pos = q & mask;
    const float re_x = chp[0][chs[0] * pos], im_x = chp[0][chs[0] * pos + 1], re_y = chp[1][chs[1] * pos], im_y = chp[1][chs[1] * pos + 1];
    const float a = pc1 * re_x + pc2 * re_y - pc3 * im_y, b = pc1 * im_x + pc2 * im_y + pc3 * re_y;
    const float t3 = phf[k], t4 = phf[k + 1];
    in[i] = a * t3 + b * t4; in[i + 1] = b * t3 - a * t4; k += 2; i += 2;
  }
  const int imax = pwid, wid = 2 * pwid;
  float c1 = 0, c2 = 0;
  for (i = imax - 1; i <= imax + 1; i++) { const float a = in[2 * i], b = in[2 * i + 1], t3 = sqrt(a * a + b * b); c1 += t3 * a; c2 += t3 * b; }
  t1 = sqrt(c1 * c1 + c2 * c2);
  if (t1 < 4) return -1;
  c1 /= t1; c2 /= t1;
  float t3 = 0, t4 = 0;
  for (i = 0; i <= wid; i++) {
    const float a = in[2 * i], b = in[2 * i + 1];
    in[2 * i] = c1 * a + c2 * b; in[2 * i + 1] = c1 * b - c2 * a;
    t3 += in[2 * i] * in[2 * i]; t4 += in[2 * i + 1] * in[2 * i + 1];
  }
  if (t4 > 0.25 * t3) return -1.;
  t4 = in[2 * imax - 2] - in[2 * imax + 2];
  t3 = 2 * (in[2 * imax - 2] + in[2 * imax + 2] - 2 * in[2 * imax]);
  if (t3 == 0) return -2.;
  t4 /= 2 * t3;
  if (t4 < 0) t4 = -sqrt(-t4); else t4 = sqrt(t4);
  int j = LRH_MAX_REFPULSES * (t4 + 0.5) + 0.5;
  if (j < 0) j = 0;
  if (j >= LRH_MAX_REFPULSES) j = LRH_MAX_REFPULSES - 1;
  const int m = 2 * c->bt_pulindex[j] * rs;
  c1 = c1 * in[2 * imax] * c->amp_factor; c2 = c2 * in[2 * imax] * c->amp_factor;
  for (int pass = 0; pass < 2; pass++) {                 /* 0: subtract; 1: put back when too little went away (:377-428) */
    const float sg = pass ? 1.f : -1.f;
    t3 = 0; t4 = 0;
    k = rs - sub_size;
    for (int q = p_max - sub_size / 2; q <= p_max + sub_size / 2; q++) {
      const int pos = q & mask;
      float *x = chp[0] + chs[0] * pos, *y = chp[1] + chs[1] * pos;
      const float r1 = rp[m + k], r2 = rp[m + k + 1];
      const float re_a = c1 * r1 - c2 * r2, im_a = c1 * r2 + c2 * r1;
      const float re_x = x[0] + sg * (pc1 * re_a), im_x = x[1] + sg * (pc1 * im_a);
      const float re_y = y[0] + sg * (pc2 * re_a + pc3 * im_a), im_y = y[1] + sg * (pc2 * im_a - pc3 * re_a);
      x[0] = re_x; x[1] = im_x; y[0] = re_y; y[1] = im_y;
      const float pn = re_x * re_x + im_x * im_x + re_y * re_y + im_y * im_y;
      t3 += pw[pos]; pw[pos] = pn; t4 += pn; k += 2;
      c->timf2_pwr[pos] = ci ? re_y * re_y + im_y * im_y : re_x * re_x + im_x * im_x;
    }
    if (pass) return -5;
    if (!(t4 / t3 > 0.5)) break;
  }
  return t4 / t3;
}

/* the search loop of first_noise_blanker, blank1.c:765-1003; returns pf (where the scan stopped) */
static int clever_search(lro_ctx *c, float *pw, int pbeg, int pend, int *fitted_out, int *rejected_out)
{
  const int mask = c->timf2pow_mask, R = c->cfg.blnfit_range;
  lrh_blanker_state *s = &c->bs;
  unsigned char *fl = c->blanker_flag;
  float avgpwr[LRH_BLN_INFO_SIZE];
  for (int i = 0; i < LRH_BLN_INFO_SIZE; i++) avgpwr[i] = 0;
  int p0 = pbeg, pf = pbeg, fitted = 0, rejected = 0;
  fl[p0] = 0;
  while (p0 != pend) { p0 = (p0 + 1) & mask; fl[p0] = 0; }
  const unsigned int nfl = s->clever_bln_limit;
  const float sizlim = 0.1 * s->timf2_noise_floor;
  for (;;) {
    while ((pw[pf] <= nfl || fl[pf] > 64) && pf != pend) pf = (pf + 1) & mask;
    if (pf == pend) break;
    p0 = (pf + mask) & mask;                       /* pf-1 (the reference lets -1 through: its first use is p0+1 masked) */
    int p_max = pf, m = R;
    float powermax = 10;
    while (p0 != pend && m > 0) {
      p0 = (p0 + 1) & mask;
      if (pw[p0] > powermax && fl[p0] < 64) { fl[p0] = 1; powermax = pw[p0]; p_max = p0; m = R; }
      m--;
    }
    if (m > 0) break;
    int no_pulse = 0;
    p0 = (p_max + mask) & mask;
    if (fl[p0] >= 64) { pf = p_max; no_pulse = 1; }
    else {
      p0 = (p_max + 1) & mask;
      if (p0 == pend) break;
      if (fl[p0] >= 64) no_pulse = 1;            /* blank1.c:834: joins no_pulse WITHOUT moving pf to p_max */
    }
    if (no_pulse) {
      while ((pw[pf] <= powermax || fl[pf] > 64) && pf != pend) { powermax = pw[pf]; pf = (pf + 1) & mask; }
      if (pf == pend) break;
      continue;
    }
    int bln_no = 0, pa = (p_max + 1) & mask, pb = (p_max + mask) & mask, k = 2;
    for (;;) {
      powermax = pw[p_max];
      float t1 = powermax * c->bt.bln[bln_no].rest;
      if (t1 < sizlim) break;
      t1 = 0;
      while (k < c->bt.bln[bln_no].size) { t1 += pw[pa] + pw[pb]; pa = (pa + 1) & mask; pb = (pb + mask) & mask; k += 2; }
      avgpwr[bln_no] = t1 / powermax;
      bln_no++;
      if (bln_no > c->bt.largest_blnfit) break;
    }
    bln_no--;
    while (bln_no >= 0 && avgpwr[bln_no] > c->bt.bln[bln_no].avgmax) bln_no--;
    float rv = -1;
    if (bln_no >= 0) rv = c->cfg.blanker_channels == 2 ? clever_subtract2(c, pw, p_max, c->bt.bln[bln_no].size) : clever_subtract(c, p_max, c->bt.bln[bln_no].size);
    if (rv < 0) { clever_set_flag(c, pw, 65, p_max, pbeg, pend); rejected++; continue; }
    fitted++;
    clever_set_flag(c, pw, 66, p_max, pbeg, pend);
  }
  *fitted_out = fitted; *rejected_out = rejected;
  return pf;
}

int lro_first_noise_blanker(lro_ctx *c, lrh_ptrs *p)
{
  const int mm = 4, mask = c->timf2pow_mask;
  const int coupled = c->cfg.blanker_channels == 2;    /* one of two channels: decisions on the exchanged power sum */
  lrh_blanker_state *s = &c->bs;
  float *own = c->timf2_pwr, *tf = c->timf2_float;
  float *pw = coupled ? c->pwr_sum : own;
  int pbeg = p->timf2p_fit;
  int pend = (p->timf2_pa / mm - c->cfg.blnfit_range + mask) & mask;
  pend &= 0xfffffffc;
  int total = (pend - pbeg + 1 + mask) & mask;
  if (total < c->cfg.blanker_min_points) return LRH_OK;
  if (coupled) {
    if (c->fin_pending) return LRH_ESTATE;               /* lro_blanker_finish of the previous call is missing */
    if (c->x_pbeg != pbeg || c->x_count < 0 || c->x_span != ((pend - pbeg) & mask)) return LRH_ESTATE;   /* lro_blanker_begin not called for this span */
    for (int q = 1; q <= c->x_count; q++) pw[(pbeg + q) & mask] = c->xbuf[q - 1];
    if (c->clever_on) {                                /* the partner's samples take their ring places */
      if (c->xw_count <= 0) return LRH_ESTATE;
      const int R = c->cfg.blnfit_range, nw = c->xw_count / 2;
      const float *slot = c->xweak + (size_t)(1 - (c->cfg.timf1_channel_index & 1)) * 2 * nw;
      for (int i = 0; i < nw; i++) { const int pos = (pbeg - R + i) & mask; c->tf_partner[2 * pos] = slot[2 * i]; c->tf_partner[2 * pos + 1] = slot[2 * i + 1]; }
    }
    c->x_count = -1;
  }
#define CLR(pos) do { pw[pos] = 0; own[pos] = 0; tf[4 * (pos)] = 0; tf[4 * (pos) + 1] = 0; } while (0)
  int cleared = 0, fitted = 0, rejected = 0, pf = pend;
  if (c->clever_on) pf = clever_search(c, pw, pbeg, pend, &fitted, &rejected);
  if (c->cfg.stupid_bln_mode != 0) {
    unsigned int nfl = s->stupid_bln_limit;
    int p0 = pbeg, ifirst = 0, pk = p0;
    int clr1 = (c->cfg.blanker_pulsewidth + 1) >> 1, clr2 = c->cfg.blanker_pulsewidth + 1;
    float pulmax = 0, totnoise = (float)(s->timf2_noise_floor * (coupled ? 2 : 1));    /* blank1.c:1017 */
    while (p0 != pend) {
      p0 = (p0 + 1) & mask;
      if (pw[p0] > nfl) {
        if (ifirst == 0) pk = p0;
        if (pw[p0] > pulmax) pulmax = pw[p0];
        ifirst++;
        cleared++;
        CLR(p0);
      } else if (ifirst != 0) {
        ifirst = 0;
        float t1 = pulmax / totnoise;
        pulmax = 0;
        if (t1 > 4) {
          if (t1 > 10000) t1 = 10000;          /* 40 dB cap */
          t1 = sqrt(t1) / 100;
          int pa = pk, i = clr1 * t1 + 0.5;
          for (int j = 0; j < i; j++) { pa = (pa + mask) & mask; CLR(pa); cleared++; }
          pa = p0; i = clr2 * t1 + 0.5;
          for (int j = 0; j < i; j++) { CLR(pa); pa = (pa + 1) & mask; cleared++; }
        }
      }
    }
  }
#undef CLR
  s->last_call_cleared = cleared; s->last_call_fitted = fitted; s->last_call_rejected = rejected;
  p->timf2p_fit = c->clever_on ? ((pf - 16 + mask) & (mask & ~3)) : pend;      /* blank1.c:1458-1465 */
  p->timf2_pn2 = mm * pend;
  int m = (p->timf2p_fit - pbeg + 1 + mask) & mask;
  s->timf2_fitted_pulses += fitted;
  s->timf2_cleared_points += cleared;
  p->timf2_blanker_points += m;
  if (p->timf2_blanker_points == 0) return LRH_OK;
  int k = m - cleared - fitted; if (k < m / 25) k = m / 25; k = (k + 2) / 4; if (k < 1) k = 1;
  float t1 = 0;
  /* one channel: the (cleared) power ring; two: each channel's own weak power, t3*t3+t4*t4 of blank1.c:1516-1523 */
  for (int p0 = pbeg; p0 != p->timf2p_fit;) { p0 = (p0 + 4) & mask; t1 += coupled ? tf[4 * p0] * tf[4 * p0] + tf[4 * p0 + 1] * tf[4 * p0 + 1] : pw[p0]; }
  t1 /= k; if (t1 < 10) t1 = 10;
  p->blanker_info_update_counter++;
  int do_update = 0;
  if (p->blanker_info_update_counter >= c->cfg.blanker_info_update_interval) {
    if (p->fft1_lowlevel_fraction < 0.1) p->blanker_info_update_counter--;
    else do_update = 1;
  }
  if (coupled) {                               /* the partner's value arrives by exchange: lro_blanker_finish goes on */
    c->xstat[0] = c->xstat[1] = 0; c->xstat[c->cfg.timf1_channel_index & 1] = t1;
    c->fin_pending = 1; c->fin_do_update = do_update; c->fin_llf = p->fft1_lowlevel_fraction;
    return LRH_OK;
  }
  s->timf2_despiked_pwrinc[0] += t1;
  return blanker_update(c, p, t1, do_update, p->fft1_lowlevel_fraction, 1);
}

/* statistics / threshold update, blank1.c:1542-1601; t1 = this call's mean power summed over the channels */
static int blanker_update(lro_ctx *c, lrh_ptrs *p, float t1, int do_update, float llf, int chans)
{
  lrh_blanker_state *s = &c->bs;
  if (do_update) {
    int iv = c->cfg.blanker_info_update_interval;
    s->timf2_despiked_pwr[0] = s->timf2_despiked_pwrinc[0] / (iv * llf);
    s->timf2_despiked_pwr[1] = s->timf2_despiked_pwrinc[1] / (iv * llf);
    s->clever_blanker_rate = 100. * (float)s->timf2_fitted_pulses / p->timf2_blanker_points;
    if (s->clever_blanker_rate > 99) s->clever_blanker_rate = 99;
    s->stupid_blanker_rate = 100. * (float)s->timf2_cleared_points / p->timf2_blanker_points;
    if (s->stupid_blanker_rate > 99) s->stupid_blanker_rate = 99;
    s->timf2_noise_floor = (s->timf2_despiked_pwr[0] + s->timf2_despiked_pwr[1]) / chans;
    if (c->cfg.stupid_bln_mode == 1) {
      if (s->stupid_blanker_rate > 20) {
        if (s->timf2_noise_floor < 30) s->timf2_noise_floor = 30;
        t1 = 0.01 * pow(s->stupid_blanker_rate - 20.0, 2.);
        if (t1 > 10) t1 = 10;
        s->timf2_noise_floor *= 1 + t1;
      } else {
        int an = c->cfg.timf2_noise_floor_avgnum;
        s->timf2_noise_floor = ((an - 1) * s->timf2_noise_floor + t1) / an;
      }
      s->stupid_bln_limit = s->timf2_noise_floor * c->cfg.stupid_bln_factor;
    }
    if (c->clever_on && c->bt.clever_bln_mode == 1) s->clever_bln_limit = s->timf2_noise_floor * c->bt.clever_bln_factor;
    p->blanker_info_update_counter = 0;
    s->timf2_despiked_pwrinc[0] = 1; s->timf2_despiked_pwrinc[1] = 1;
    s->timf2_fitted_pulses = 0; s->timf2_cleared_points = 0; p->timf2_blanker_points = 0;
  }
  return LRH_OK;
}

/* ------------------------------------------------------------------ fft2 */

/* waterfall geometry shared by harness, oracle and product (see DESIGN.md "waterfall line") */
static void wf_geometry(const lro_ctx *c, int *hg_xpp, int *hg_ppx, int *wg_xpp, int *wg_ppx, int *wg_first)
{
  int r = c->N2 / c->N1; if (r < 1) r = 1;
  int mode = c->cfg.wf_mode;
  if (mode == 1) { *hg_xpp = 1; *hg_ppx = 1; } else if (mode > 1) { *hg_xpp = mode; *hg_ppx = 0; } else { *hg_xpp = 0; *hg_ppx = -mode; }
  if (*hg_xpp > 0 && *hg_xpp >= r) { *wg_xpp = *hg_xpp / r; *wg_ppx = 0; }
  else { *wg_xpp = 0; *wg_ppx = *hg_ppx > 0 ? *hg_ppx * r : r / (*hg_xpp > 0 ? *hg_xpp : 1); if (*wg_ppx < 1) *wg_ppx = 1; }
  *wg_first = c->cfg.wf_first_xpoint / r;
}

/* FFT2_WATERFALL_LINE, fft2.c:707-815: short y = 1000*log10(powersum*yfac), clamp +-32767 */
static void fft2_waterfall_line(lro_ctx *c, lrh_ptrs *p, const float *ps)
{
  int hx, hp, wx, wp, wfirst; wf_geometry(c, &hx, &hp, &wx, &wp, &wfirst);
  int npix = c->cfg.wf_xpixels, siz = c->N2;
  int16_t *line = c->wg_waterf + p->wg_waterf_ptr;
  const float *yf = c->wg_waterf_yfac;
  double *pre = c->wf_pre ? c->wf_pre + p->wg_waterf_ptr : NULL, v_;      /* float64 build: the values before the truncation to short */
#define WF_PRE(k_, val_) do { if (pre && (k_) < npix) pre[k_] = (val_); } while (0)
  if (!ps) goto advance;                      /* coupled channels: the line is written by lro_fft2_xy_finish */
  float a2 = 1, a3; int y, itab, i;
  if (wx > 0) a2 = wx; else a2 = 1. / wp;
  a3 = wfirst + 0.5 * a2;
  if (hx == 1 || hp == 1) {
    i = c->cfg.wf_first_xpoint;
    for (int ix = 0; ix < npix; ix++) {
      itab = a3; a3 += a2;
      v_ = 1000. * log10(ps[i] * yf[itab]); y = v_; WF_PRE(ix, v_);
      if (y < -32767) y = -32767; if (y > 32767) y = 32767;
      line[ix] = y; i++;
    }
  } else if (hx == 0) {                       /* interpolate, fft2.c:739-783 */
    float yval, r1, der;
    i = c->cfg.wf_first_xpoint; itab = a3; a3 += a2;
    v_ = 1000. * log10(ps[i] * yf[itab]); y = v_; yval = y; WF_PRE(0, v_);
    if (y < -32767) y = -32767; if (y > 32767) y = 32767;
    line[0] = y; i++;
    int mlim = npix - hp, ix, k;
    for (ix = 0; ix < mlim; ix += hp) {
      itab = a3; a3 += a2;
      r1 = 1000. * log10(ps[i] * yf[itab]); der = (r1 - yval) / hp;
      for (k = ix + 1; k <= ix + hp; k++) { yval = yval + der; y = yval; WF_PRE(k, yval); if (y < -32767) y = -32767; if (y > 32767) y = 32767; line[k] = y; }
      yval = r1; i++;
    }
    if (i < siz) {
      itab = a3;
      r1 = 1000. * log10(ps[i] * yf[itab]); der = (r1 - yval) / hp;
      for (k = ix + 1; k <= ix + hp; k++) { yval = yval + der; y = yval; WF_PRE(k, yval); if (y < -32767) y = -32767; if (y > 32767) y = 32767; if (k < npix) line[k] = y; }
    }
  } else {                                    /* max over a group, fft2.c:784-811 */
    int ia = c->cfg.wf_first_xpoint, ib = ia + hx;
    for (int ix = 0; ix < npix; ix++) {
      float r2 = 0;
      for (i = ia; i < ib; i++) { float r1 = ps[i]; if (r1 > r2) r2 = r1; }
      itab = a3; a3 += a2;
      float a1 = yf[itab];
      v_ = 1000. * log10(a1 * r2); y = v_; WF_PRE(ix, v_);
      if (y < -32767) y = -32767; if (y > 32767) y = 32767;
      line[ix] = y;
      ia = ib; ib += hx; if (ib >= siz) ib = siz;
    }
  }
advance:
  /* update_wg_waterf, fft1.c:104-113 */
  p->wg_waterf_ptr -= npix; if (p->wg_waterf_ptr < 0) p->wg_waterf_ptr += c->cfg.wf_lines * npix;
  p->wg_waterf_sum_counter = 0;
  p->fft2_liminfo_cnt++;
}



/* make_fft2 mode 15 until FFT2_COMPLETE: fft2.c:86-141 (load, window, big_fftforward), 647-705 (power),
   707-815 (waterfall), 1831-1845 (pointers) */
int lro_make_fft2(lro_ctx *c, lrh_ptrs *p, int batch)
{
  int N = c->N2, n = c->cfg.fft2_n, mask = c->timf2_mask;
  for (int b = 0; b < batch; b++) {
    float *z = c->fft2_float + (size_t)2 * p->fft2_na * N;
    int p0 = p->timf2_px;
    const float *tf = c->timf2_float;
    if (c->cfg.fft2_sinpow != 0)
      for (int i = 0; i < N; i++) { z[2 * i] = c->fft2_window[i] * (tf[p0] + tf[p0 + 2]); z[2 * i + 1] = c->fft2_window[i] * (tf[p0 + 1] + tf[p0 + 3]); p0 = (p0 + 4) & mask; }
    else
      for (int i = 0; i < N; i++) { z[2 * i] = tf[p0] + tf[p0 + 2]; z[2 * i + 1] = tf[p0 + 1] + tf[p0 + 3]; p0 = (p0 + 4) & mask; }
    dif_stages(N, n, z, c->fft2tab, +1, 2); bitrev_inplace(N, n, z, 2);
    if (c->spurs) lro_spur_hook(c, p->fft2_na);                     /* FFT2_ELIMINATE_SPURS, fft2.c:647-652 */
    float *pwra = c->fft2_power + (size_t)p->fft2_na * N;
    if (p->wg_waterf_sum_counter == 0) for (int i = 0; i < N; i++) { pwra[i] = z[2 * i] * z[2 * i] + z[2 * i + 1] * z[2 * i + 1]; c->fft2_powersum[i] = pwra[i]; }
    else                               for (int i = 0; i < N; i++) { pwra[i] = z[2 * i] * z[2 * i] + z[2 * i + 1] * z[2 * i + 1]; c->fft2_powersum[i] += pwra[i]; }
    p->wg_waterf_sum_counter++;
    spur_search_row(c, pwra);                                        /* fft2.c:673-699 (genparm[MAX_NO_OF_SPURS] > 0) */
    if (p->wg_waterf_sum_counter >= c->cfg.waterfall_avgnum) fft2_waterfall_line(c, p, c->cfg.blanker_channels == 2 ? NULL : c->fft2_powersum);
    p->timf2_px = (p->timf2_px + 4 * c->M2) & mask;
    p->fft2_na = (p->fft2_na + 1) & c->fft2n_mask;
    p->fft2_pa = 2 * p->fft2_na * N;
    p->fft2_nb = (p->fft2_nb + 1) & c->fft2n_mask;
    if (p->fft2_nm != c->fft2n_mask) p->fft2_nm++;
  }
  return LRH_OK;
}

/* Two coupled channels, see include/linrad_hip.h: the own new transforms go to slot timf1_channel_index of LRH_X_BINS ... */
int lro_fft2_xy_begin(lro_ctx *c, const lrh_ptrs *at, int batch, size_t *count)
{
  if (c->cfg.blanker_channels != 2) return LRH_ESTATE;
  if (batch < 1 || batch > c->cfg.max_fft2n || !count) return LRH_EINVAL;
  const size_t per = (size_t)2 * c->N2;
  float *slot = c->xbins + (size_t)(c->cfg.timf1_channel_index & 1) * batch * per;
  for (int b = 0; b < batch; b++)
    memcpy(slot + b * per, c->fft2_float + (size_t)((at->fft2_na + b) & c->fft2n_mask) * per, 4 * per);
  *count = batch * per;
  return LRH_OK;
}
/* ... and, once the partner's slot has arrived, the cross products and sums of fft2.c:1622-1640 and the waterfall
   line of fft2.c:1700-1815, whose power is (x2+y2) + 2 (re_xy^2 + im_xy^2 - x2 y2) / (x2+y2) of the sums */
int lro_fft2_xy_finish(lro_ctx *c, const lrh_ptrs *at, int batch)
{
  if (c->cfg.blanker_channels != 2) return LRH_ESTATE;
  if (batch < 1 || batch > c->cfg.max_fft2n) return LRH_EINVAL;
  const int N = c->N2;
  lrh_ptrs q = *at;
  float *r = c->tmp;
  for (int b = 0; b < batch; b++) {
    const float *x = c->xbins + (size_t)b * 2 * N, *y = c->xbins + ((size_t)batch + b) * 2 * N;
    float *ya = c->fft2_xypower + (size_t)4 * q.fft2_na * N, *s = c->fft2_xysum;
    for (int i = 0; i < N; i++) {
      ya[4 * i] = x[2 * i] * x[2 * i] + x[2 * i + 1] * x[2 * i + 1];
      ya[4 * i + 1] = y[2 * i] * y[2 * i] + y[2 * i + 1] * y[2 * i + 1];
      ya[4 * i + 2] = -x[2 * i] * y[2 * i + 1] + x[2 * i + 1] * y[2 * i];
      ya[4 * i + 3] = x[2 * i] * y[2 * i] + x[2 * i + 1] * y[2 * i + 1];
      if (q.wg_waterf_sum_counter == 0) for (int k = 0; k < 4; k++) s[4 * i + k] = ya[4 * i + k];
      else for (int k = 0; k < 4; k++) s[4 * i + k] += ya[4 * i + k];
    }
    q.wg_waterf_sum_counter++;
    if (q.wg_waterf_sum_counter >= c->cfg.waterfall_avgnum) {
      for (int i = 0; i < N; i++) {
        float t1 = s[4 * i] + s[4 * i + 1];
        r[i] = t1 + 2 * (s[4 * i + 3] * s[4 * i + 3] + s[4 * i + 2] * s[4 * i + 2] - s[4 * i] * s[4 * i + 1]) / t1;
      }
      fft2_waterfall_line(c, &q, r);
    }
    q.fft2_na = (q.fft2_na + 1) & c->fft2n_mask;
  }
  return LRH_OK;
}

/* ------------------------------------------------------------------ mix1 */

int lro_set_mix1_selfreq(lro_ctx *c, double fq) { c->ms.mix1_selfreq = fq; return LRH_OK; }
int lro_get_mix1_state(lro_ctx *c, lrh_mix1_state *st) { *st = c->ms; return LRH_OK; }

/* set_mix1_phases, mix1.c:781-861 (float branch) */
static int set_mix1_phases(lro_ctx *c, f32 fq)     /* host bookkeeping in float32 whatever the build: these scalars DEFINE what the mixer does */
{
  lrh_mix1_state *s = &c->ms;
  if (fq < c->cfg.mix1_lowest_fq || fq > c->cfg.mix1_highest_fq) return LRH_ERANGE;
  int size = c->Nm;
  f32 t1 = fq * c->cfg.fftx_points_per_hz, t2;
  int pnt = t1 + 0.5;
  int k = pnt % size;
  t2 = size * (pnt / size);
  t2 = t1 - t2 - k;
  t2 = t2 - (int)(t2);
  s->mix1_phase_rot = t2 * 2 * PI_L / size;
  k = (k * c->Mm) % size;
  s->mix1_old_phase = s->mix1_phase;
  s->mix1_phase += s->mix1_phase_step;
  s->mix1_phase_step = k * 2 * PI_L / size;
  s->mix1_old_point = (s->mix1_point != -1) ? s->mix1_point : pnt;
  s->mix1_point = pnt;
  if (s->mix1_phase > PI_L) s->mix1_phase -= 2 * PI_L;
  if (s->mix1_phase < PI_L) s->mix1_phase += 2 * PI_L;    /* reference quirk: keeps phase in [pi,3pi) */
  return LRH_OK;
}

/* gather of fft2_mix1_fixed (mix1.c:955-983) / fft1_mix1_fixed (mix1.c:1015-1030) + do_mix1 (mix1.c:55-195; dfq forced
   to 0 at mix1.c:103).  z: first float of the source transform, lim: float index limit mm*nn*fft1_last_point. */
static int mix1_block(lro_ctx *c, lrh_ptrs *p, const float *zbase, int lim, float fq)
{
  int Nm = c->Nm, n = Nm, n2 = 2 * Nm, block = 2 * c->Mm;
  lrh_mix1_state *s = &c->ms;
  float *tmp = c->tmp, *t3 = c->timf3_float;
  if (s->mix1_selfreq >= 0) {
    int rc = set_mix1_phases(c, fq); if (rc) return rc;
    int k = s->mix1_point * 2;
    int ib = n; if (ib > lim - k) ib = lim - k; if (ib < 0) ib = 0;
    const float *z = zbase + k;
    for (int i = 0; i < ib; i++) tmp[i] = z[i];
    for (int i = ib; i < n; i++) tmp[i] = 0;
    k -= n2; ib = n; if (ib < -k) ib = -k; if (ib > n2) ib = n2;
    for (int i = n; i < ib; i++) tmp[i] = 0;
    z = zbase + k;
    for (int i = ib; i < n2; i++) tmp[i] = z[i];
    /* frequency-domain window, mix1.c:113-135 */
    int i = 0, j = Nm - 1, w = Nm / 2 - 1;
    float t1 = c->mix1_fqwin[w]; tmp[0] *= t1; tmp[1] *= t1; i++;
    while (j > i) { t1 = c->mix1_fqwin[w]; tmp[2 * i] *= t1; tmp[2 * i + 1] *= t1; tmp[2 * j] *= t1; tmp[2 * j + 1] *= t1; j--; w--; i++; }
    t1 = c->mix1_fqwin[w]; tmp[2 * i] *= t1; tmp[2 * i + 1] *= t1;
    /* fftback, fft0.c:481-533 */
    dif_stages(Nm, c->mix1_n, tmp, c->mix1tab, -1, 2); bitrev_inplace(Nm, c->mix1_n, tmp, 2);
    int pa = p->timf3_pa;
    float t2 = s->mix1_phase_rot; t1 = s->mix1_phase;
    if (c->Im == 0) {                        /* mix1.c:141-155 */
      for (i = 0; i < 2 * Nm; i += 2) {
        float sn = sin(t1), cs = cos(t1);
        t3[pa + i] = cs * tmp[i] - sn * tmp[i + 1]; t3[pa + i + 1] = cs * tmp[i + 1] + sn * tmp[i];
        t1 += t2;
      }
      s->mix1_phase = t1;
    } else if (c->Im != c->Mm) {             /* other windows: crossover functions, mix1.c:196-262 (dfq = 0) */
      const float *win = c->mix1_window, *w1t = c->mix1_sin2win, *w2t = c->mix1_cos2win;
      int p0 = p->timf3_pa, X = c->Xm;
      float r1 = s->mix1_old_phase;
      float r2 = t2 - 2 * (s->mix1_old_point - s->mix1_point) * PI_L / Nm;
      int k = 2 * c->Im / 2;                 /* mm*interleave_points/2, mm = 2 */
      k -= 2 * (X / 2);
      int j = k / 2 + X;
      int ia = 2 * X;
      for (i = 0; i < ia; i += 2) {
        float sn = sin(t1), cs = cos(t1), rs = sin(r1), rc2 = cos(r1);
        float w1 = w1t[i >> 1], w2 = w2t[i >> 1];
        float a1 = w2 * t3[p0], a2 = w2 * t3[p0 + 1];
        t3[p0] = rc2 * a1 - rs * a2 + (cs * tmp[i + k] - sn * tmp[i + k + 1]) * w1;
        t3[p0 + 1] = rc2 * a2 + rs * a1 + (cs * tmp[i + k + 1] + sn * tmp[i + k]) * w1;
        t1 += t2; r1 += r2;
        p0 = (p0 + 2) & c->timf3_mask;
      }
      int ib = c->Mm + 2 + 2 * (X / 2);
      for (i = ia; i < ib; i += 2) {
        float sn = sin(t1), cs = cos(t1), rw = win[j];
        t3[p0] = (cs * tmp[i + k] - sn * tmp[i + k + 1]) * rw;
        t3[p0 + 1] = (cs * tmp[i + k + 1] + sn * tmp[i + k]) * rw;
        p0 = (p0 + 2) & c->timf3_mask;
        t1 += t2; j++;
      }
      j--;
      int ic = 2 * c->Mm;
      for (i = ib; i < ic; i += 2) {
        j--;
        float sn = sin(t1), cs = cos(t1), rw = win[j];
        t3[p0] = (cs * tmp[i + k] - sn * tmp[i + k + 1]) * rw;
        t3[p0 + 1] = (cs * tmp[i + k + 1] + sn * tmp[i + k]) * rw;
        t1 += t2;
        p0 = (p0 + 2) & c->timf3_mask;
      }
      s->mix1_phase = t1;
      int id = 2 * (X + c->Mm);
      for (i = ic; i < id; i += 2) {
        t3[p0] = tmp[i + k]; t3[p0 + 1] = tmp[i + k + 1];
        p0 = (p0 + 2) & c->timf3_mask;
      }
    } else {                                 /* sin^2, 50 % overlap: mix1.c:161-195 */
      float r1 = s->mix1_old_phase;
      float r2 = t2 - 2 * (s->mix1_old_point - s->mix1_point) * PI_L / Nm;
      for (i = 0; i < Nm; i += 2) {
        float sn = sin(t1), cs = cos(t1), rs = sin(r1), rc2 = cos(r1);
        float a1 = t3[pa + i], a2 = t3[pa + i + 1];
        t3[pa + i] = rc2 * a1 - rs * a2 + cs * tmp[i] - sn * tmp[i + 1];
        t3[pa + i + 1] = rc2 * a2 + rs * a1 + cs * tmp[i + 1] + sn * tmp[i];
        r1 += r2; t1 += t2;
      }
      s->mix1_phase = t1;
      pa = ((p->timf3_pa + block) & c->timf3_mask) - block;
      for (i = Nm; i < 2 * Nm; i++) t3[pa + i] = tmp[i];
    }
  } else {                                   /* mix1_clear, mix1.c:766-779 */
    for (int i = 0; i < block; i++) t3[p->timf3_pa + i] = 0;
  }
  p->timf3_pa = (p->timf3_pa + block) & c->timf3_mask;
  return LRH_OK;
}

int lro_fft2_mix1_fixed(lro_ctx *c, lrh_ptrs *p, int batch)
{
  if (!c->cfg.second_fft_enable) return LRH_ESTATE;
  int ratio = c->N2 / c->N1; if (ratio < 1) ratio = 1;
  int lim = 2 * ratio * (c->N1 - 1); if (lim > 2 * c->N2) lim = 2 * c->N2;   /* clamp only bites when N2 < N1 */
  for (int b = 0; b < batch; b++) {
    int rc = mix1_block(c, p, c->fft2_float + (size_t)2 * p->fft2_nx * c->N2, lim, (float)c->ms.mix1_selfreq); if (rc) return rc;
    p->fft2_nx = (p->fft2_nx + 1) & c->fft2n_mask;
  }
  return LRH_OK;
}

/* do_mix1_afc, mix1.c:648-768, up to the call of do_mix1: bookkeeping on the per-transform frequency tables */
#define BWFAC 0.03
static void afc_tables(lro_ctx *c, lrh_afc *a, int nx, int na, int mask)
{
  f32 *fq = a->mix1_fq_mid, *dfq = a->mix1_fq_slope, *d2fq = a->mix1_fq_curv, *fqs = a->mix1_fq_start;      /* the caller's tables: float32 in either build */
  int ka = (nx + mask) & mask, kb = (nx + 1) & mask;
  float t1 = fq[nx] + dfq[ka], t2 = fq[kb], t3;
  if (fabs(t2 - t1) < BWFAC * a->baseband_bw_hz) {
    dfq[nx] = fq[kb] - fq[nx];
    d2fq[nx] = dfq[nx] - dfq[ka];
  } else {
    float error = t2 - t1, curv = BWFAC * a->baseband_bw_hz;
    if (error < 0) curv = -curv;
    t3 = fabs(error) / 2;
    int kk = nx, k = 0, ia = ka, ib = kb;
    while (fabs(error) > t3 && kk != na) {
      d2fq[kk] = curv; dfq[kk] = dfq[ia] + curv;
      t1 = fq[kk] + dfq[kk];
      error = fq[ib] - t1;
      if (t1 < c->cfg.mix1_lowest_fq) t1 = c->cfg.mix1_lowest_fq;
      if (t1 > c->cfg.mix1_highest_fq) t1 = c->cfg.mix1_highest_fq;
      fq[ib] = t1;
      ia = (ia + 1) & mask; kk = (kk + 1) & mask; ib = (ib + 1) & mask; k++;
    }
    t3 = error; curv = -curv;
    while (k > 0 && kk != na && t3 * error > 0) {
      d2fq[kk] = curv; dfq[kk] = dfq[ia] + curv;
      t1 = fq[kk] + dfq[kk];
      error = fq[ib] - t1;
      fq[ib] = t1;
      ia = (ia + 1) & mask; kk = (kk + 1) & mask; ib = (ib + 1) & mask; k--;
    }
  }
  fqs[kb] = fq[nx] + 0.5 * dfq[nx] + 0.25 * d2fq[nx];
}

/* fft2_mix1_afc, mix1.c:863-932 (fft1_first_point = 0: the gather equals the fixed variant's) */
int lro_fft2_mix1_afc(lro_ctx *c, lrh_ptrs *p, int batch, lrh_afc *afc)
{
  if (!c->cfg.second_fft_enable || !afc || c->ms.mix1_selfreq < 0) return LRH_ESTATE;
  int ratio = c->N2 / c->N1; if (ratio < 1) ratio = 1;
  int lim = 2 * ratio * (c->N1 - 1); if (lim > 2 * c->N2) lim = 2 * c->N2;
  for (int b = 0; b < batch; b++) {
    int rc = mix1_block(c, p, c->fft2_float + (size_t)2 * p->fft2_nx * c->N2, lim, afc->mix1_fq_mid[p->fft2_nx]); if (rc) return rc;
    afc_tables(c, afc, p->fft2_nx, p->fft2_na, c->fft2n_mask);
    p->fft2_nx = (p->fft2_nx + 1) & c->fft2n_mask;
  }
  return LRH_OK;
}

/* fft1_mix1_afc, mix1.c:1044-1097 */
int lro_fft1_mix1_afc(lro_ctx *c, lrh_ptrs *p, int batch, lrh_afc *afc)
{
  if (c->cfg.second_fft_enable || !afc || c->ms.mix1_selfreq < 0) return LRH_ESTATE;
  for (int b = 0; b < batch; b++) {
    int rc = mix1_block(c, p, c->fft1_float + p->fft1_px, 2 * (c->N1 - 1), afc->mix1_fq_mid[p->fft1_nx]); if (rc) return rc;
    afc_tables(c, afc, p->fft1_nx, p->fft1_nb, c->fft1n_mask);
    p->fft1_nx = (p->fft1_nx + 1) & c->fft1n_mask;
    p->fft1_px = (p->fft1_px + 2 * c->N1) & c->fft1_mask;
  }
  return LRH_OK;
}

/* fft1_mix1_fixed, mix1.c:995-1042 */
int lro_fft1_mix1_fixed(lro_ctx *c, lrh_ptrs *p, int batch)
{
  if (c->cfg.second_fft_enable) return LRH_ESTATE;
  for (int b = 0; b < batch; b++) {
    int rc = mix1_block(c, p, c->fft1_float + p->fft1_px, 2 * (c->N1 - 1), (float)c->ms.mix1_selfreq); if (rc) return rc;
    p->fft1_nx = (p->fft1_nx + 1) & c->fft1n_mask;
    p->fft1_px = (p->fft1_px + 2 * c->N1) & c->fft1_mask;
  }
  return LRH_OK;
}

/* compute_timf2_powersum, wcw.c:80-138 (1 channel, float) */
int lro_compute_timf2_powersum(lro_ctx *c, lrh_ptrs *p)
{
  int blk = c->cfg.timf2_blockpower_block;
  if (blk <= 0) return LRH_ESTATE;
  while (((p->timf2_pn2 - p->timf2_pb + 4 * c->cfg.timf2pow_size) & c->timf2_mask) > blk) {
    int i = p->timf2_pb;
    p->timf2_pb = (p->timf2_pb + blk) & c->timf2_mask;
    float t1 = 0;
    while (i != p->timf2_pb) { t1 += c->timf2_float[i] * c->timf2_float[i] + c->timf2_float[i + 1] * c->timf2_float[i + 1]; i = (i + 4) & c->timf2_mask; }
    c->timf2_blockpower[p->timf2_blockpower_pa] = t1;
    p->timf2_blockpower_pa = (p->timf2_blockpower_pa + 1) & (c->cfg.timf2_blockpower_size - 1);
  }
  return LRH_OK;
}

/* ------------------------------------------------------------------ fft3 / mix2 */

int lro_set_basebraw_fir(lro_ctx *c, const f32 *fir, int pts)
{
  if (!c->N3 || c->pol_set) return LRH_ESTATE;
  free(c->basebraw_fir); c->basebraw_fir = NULL; c->basebraw_fir_pts = 0;
  if (!fir) return LRH_OK;
  if (pts < 1 || !(pts & 1) || pts + pts / 2 > c->I3 + c->N3 / c->Nm2 + 1) return LRH_EINVAL;   /* the first FIR of a transform must not reach behind its samples */
  c->basebraw_fir = malloc(sizeof(float) * pts); in_f32(c->basebraw_fir, fir, pts); c->basebraw_fir_pts = pts;
  return LRH_OK;
}
int lro_set_bg_filterfunc(lro_ctx *c, const f32 *f) { if (!c->N3) return LRH_ESTATE; in_f32(c->bg_filterfunc, f, c->N3); return LRH_OK; }

/* make_fft3_all, 1 channel, transform part (fft3.c:240-283): window (mode-1 storage), e^{+j} radix-2 DIF without the
   conjugation fft1 applies, permute with half swap (DC at N/2); pointers fft3.c:784, 797 */
int lro_make_fft3_all(lro_ctx *c, lrh_ptrs *p, int batch)
{
  if (!c->N3) return LRH_ESTATE;
  int N = c->N3, n = c->cfg.fft3_n, nn = N / 2, m = c->timf3_mask;
  for (int b = 0; b < batch; b++) {
    float *z = c->tmp, *out = c->fft3 + p->fft3_pa;
    int pa = p->timf3_px, pb = (pa + N) & m;
    int win = c->cfg.fft3_sinpow != 0;
    for (int ia = 0; ia < nn; ia++) {
      float wa = win ? c->fft3_window[2 * ia] : 1.0f, wb = win ? c->fft3_window[2 * ia + 1] : 1.0f;
      z[2 * ia] = c->timf3_float[pa] * wa; z[2 * ia + 1] = c->timf3_float[pa + 1] * wa;
      z[2 * (ia + nn)] = c->timf3_float[pb] * wb; z[2 * (ia + nn) + 1] = c->timf3_float[pb + 1] * wb;
      pa = (pa + 2) & m; pb = (pb + 2) & m;
    }
    dif_stages(N, n, z, c->fft3tab, +1, 2);
    for (unsigned i = 0; i < (unsigned)N; i++) { unsigned k = (bitrev(i, n) + nn) & (N - 1); out[2 * k] = z[2 * i]; out[2 * k + 1] = z[2 * i + 1]; }
    p->timf3_px = (p->timf3_px + 2 * c->M3) & m;
    p->fft3_pa = (p->fft3_pa + 2 * N) & (c->cfg.max_fft3n * 2 * N - 1);
  }
  return LRH_OK;
}

/* two coupled channels, see include/linrad_hip.h: the own channel's share of A = c1 X + (c2 - j c3) Y and
   B = c1 Y - (c2 + j c3) X (mix2.c:340-343, 377-380) for mix2.size bins around fft3_size/2, bin j = fft3_size/2 - size/2 + j */
int lro_set_pol(lro_ctx *c, f32 c1, f32 c2, f32 c3)
{
  LRO_F32_ONLY(c);
  if (c->cfg.blanker_channels != 2) return LRH_ESTATE;
  if ((c->cfg.timf1_channel_index & 1) == 0) { c->pol_w[0] = c1; c->pol_w[1] = 0; c->pol_w[2] = -c2; c->pol_w[3] = -c3; }
  else { c->pol_w[0] = c2; c->pol_w[1] = -c3; c->pol_w[2] = c1; c->pol_w[3] = 0; }
  c->pol_set = 1; return LRH_OK;
}
int lro_set_combine_weights(lro_ctx *c, f32 wa_re, f32 wa_im, f32 wb_re, f32 wb_im)
{
  LRO_F32_ONLY(c);
  if (!c->xpol) return LRH_ESTATE;
  c->pol_w[0] = wa_re; c->pol_w[1] = wa_im; c->pol_w[2] = wb_re; c->pol_w[3] = wb_im; c->pol_set = 1; return LRH_OK;
}
int lro_mix2_pol_begin(lro_ctx *c, const lrh_ptrs *p, int batch, size_t *count)
{
  if (!c->N3 || !c->pol_set) return LRH_ESTATE;
  if (batch < 1 || batch > c->cfg.max_fft3n || !count) return LRH_EINVAL;
  const int N = c->N3, size = c->Nm2;
  const float war = c->pol_w[0], wai = c->pol_w[1], wbr = c->pol_w[2], wbi = c->pol_w[3];
  float *A = c->xpol, *B = c->xpol + (size_t)batch * 2 * size;
  for (int b = 0; b < batch; b++) {
    const float *f3 = c->fft3 + ((p->fft3_px + b * 2 * N) & (c->cfg.max_fft3n * 2 * N - 1));
    for (int j = 0; j < size; j++) {
      const float re = f3[2 * (N / 2 - size / 2 + j)], im = f3[2 * (N / 2 - size / 2 + j) + 1];
      float *a = A + ((size_t)b * size + j) * 2, *o = B + ((size_t)b * size + j) * 2;
      a[0] = war * re - wai * im; a[1] = war * im + wai * re;
      o[0] = wbr * re - wbi * im; o[1] = wbr * im + wbi * re;
    }
  }
  c->pol_batch = batch;
  *count = (size_t)4 * batch * size;
  return LRH_OK;
}

/* fft3_mix2, mixer_mode 1 (mix2.c:145-176) + pointers (mix2.c:1079, 2057-2059); pinned by the compiled reference run up to
   its thread-command check (mix2.c:749), golden n10_n12_fft3 and the two-channel chain goldens */
int lro_fft3_mix2(lro_ctx *c, lrh_ptrs *p, int batch)
{
  if (!c->N3) return LRH_ESTATE;
  int N = c->N3, size = c->Nm2, sizhalf = size / 2, nn = 2 * (size - 1), bmask = c->cfg.baseband_size - 1;
  const int pol = c->pol_set;
  if (pol && c->pol_batch != batch) return LRH_ESTATE;      /* lro_mix2_pol_begin + all-reduce come first */
  c->pol_batch = 0;
  if (c->basebraw_fir) {                                 /* bg.mixer_mode == 2: FIR on timf3, decimating by fft3_size / mix2.size (mix2.c:217-246) */
    const int pts = c->basebraw_fir_pts, resamp = N / size, m3 = c->timf3_mask, tsize = c->cfg.timf3_size;
    const float *t3 = c->timf3_float, *fir = c->basebraw_fir;
    for (int b = 0; b < batch; b++) {
      int p0 = p->baseb_pa;
      for (int k = 1; k <= c->Mm2; k++) {
        int pa = (p->timf3_py + 2 * (1 - pts + N - c->M3 + k * resamp) + tsize) & m3, mm = pa;
        float t1 = t3[pa] * fir[pts / 2], t2 = t3[pa + 1] * fir[pts / 2];
        for (int i = pts / 2 - 1; i >= 0; i--) {
          pa = (pa + 2) & m3; mm = (mm - 2 + tsize) & m3;
          t1 += (t3[pa] + t3[mm]) * fir[i]; t2 += (t3[pa + 1] + t3[mm + 1]) * fir[i];
        }
        c->baseb_raw[2 * p0] = t1; c->baseb_raw[2 * p0 + 1] = t2;
        p0 = (p0 + 1) & bmask;
      }
      p->baseb_pa = (p->baseb_pa + c->Mm2) & bmask;
      p->fft3_px = (p->fft3_px + 2 * N) & (c->cfg.max_fft3n * 2 * N - 1);
      p->timf3_py = (p->timf3_py + 2 * c->M3) & m3;                      /* mix2.c:2060 */
    }
    return LRH_OK;
  }
  for (int b = 0; b < batch; b++) {
    float *tmp = c->tmp;
    const float *f3 = c->fft3;
    int p0 = p->fft3_px + N, k = N / 2;
    if (pol) {        /* the summed A (channel 0) or B (channel 1) stands in for the spectrum; addressed like fft3 around its centre */
      f3 = c->xpol + ((size_t)(c->cfg.blanker_channels == 2 ? (c->cfg.timf1_channel_index & 1) : 0) * batch + b) * 2 * size;
      p0 = size;
    }
    for (int i = 0; i < sizhalf; i++) { tmp[2 * i] = f3[p0 + 2 * i] * c->bg_filterfunc[k + i]; tmp[2 * i + 1] = f3[p0 + 2 * i + 1] * c->bg_filterfunc[k + i]; }
    for (int i = 0; i < sizhalf; i++) { tmp[nn - 2 * i] = f3[p0 - 2 * i - 2] * c->bg_filterfunc[k - i - 1]; tmp[nn - 2 * i + 1] = f3[p0 - 2 * i - 1] * c->bg_filterfunc[k - i - 1]; }
    dif_stages(size, c->cfg.mix2_n, tmp, c->mix2tab, -1, 2); bitrev_inplace(size, c->cfg.mix2_n, tmp, 2);
    float *br = c->baseb_raw;
    if (c->cfg.fft3_sinpow == 2) {                       /* THIRD_FFT_SINPOW == 2: 50 % overlap-add, mix2.c:158-176 */
      for (int i = 0; i < sizhalf; i++) { br[2 * p->baseb_pa + 2 * i] += tmp[2 * i]; br[2 * p->baseb_pa + 2 * i + 1] += tmp[2 * i + 1]; }
      int q = (p->baseb_pa + sizhalf) & bmask;
      for (int i = 0; i < size; i++) br[2 * q + i] = tmp[size + i];
    } else if (c->Im2 == 0) {                            /* no window: transforms simply follow each other */
      for (int i = 0; i < 2 * size; i++) br[2 * p->baseb_pa + i] = tmp[i];
    } else {                                             /* other windows: crossover functions, mix2.c:177-216 */
      const int X = c->Xm2;
      int p0 = p->baseb_pa, k = c->Im2 - 2 * (X / 2), j = k / 2 + X, i;
      const int ia = 2 * X, ib = c->Mm2 + 2 + 2 * (X / 2), ic = 2 * c->Mm2, id = 2 * (X + c->Mm2);
      for (i = 0; i < ia; i += 2) {                      /* the raw tail the previous transform parked here, blended with this one's start */
        br[2 * p0] = br[2 * p0] * c->mix2_cos2win[i >> 1] + tmp[i + k] * c->mix2_sin2win[i >> 1];
        br[2 * p0 + 1] = br[2 * p0 + 1] * c->mix2_cos2win[i >> 1] + tmp[i + k + 1] * c->mix2_sin2win[i >> 1];
        p0 = (p0 + 1) & bmask;
      }
      for (i = ia; i < ib; i += 2) { br[2 * p0] = tmp[i + k] * c->mix2_window[j]; br[2 * p0 + 1] = tmp[i + k + 1] * c->mix2_window[j]; p0 = (p0 + 1) & bmask; j++; }
      j--;
      for (i = ib; i < ic; i += 2) { j--; br[2 * p0] = tmp[i + k] * c->mix2_window[j]; br[2 * p0 + 1] = tmp[i + k + 1] * c->mix2_window[j]; p0 = (p0 + 1) & bmask; }
      for (i = ic; i < id; i += 2) { br[2 * p0] = tmp[i + k]; br[2 * p0 + 1] = tmp[i + k + 1]; p0 = (p0 + 1) & bmask; }
    }
    p->baseb_pa = (p->baseb_pa + c->Mm2) & bmask;
    p->fft3_px = (p->fft3_px + 2 * N) & (c->cfg.max_fft3n * 2 * N - 1);
    p->timf3_py = (p->timf3_py + 2 * c->M3) & c->timf3_mask;              /* mix2.c:2060 */
  }
  return LRH_OK;
}

/* ------------------------------------------------------------------ orchestration */

/* single-CPU branch of wideband_dsp, wcw.c:1036-1118, batched */
int lro_set_exchange(lro_ctx *c, lrh_exchange_fn fn, void *user)
{
  LRO_F32_ONLY(c);
  if (!c || c->cfg.blanker_channels != 2) return LRH_ESTATE;
  c->xfn = fn; c->xuser = user;
  return LRH_OK;
}
static int lro_exchange(lro_ctx *c, int which, int op, size_t count)
{
  void *ptr = NULL;
  if (!count) return LRH_OK;
  int rc = lro_exchange_ptr(c, which, &ptr);
  if (rc) return rc;
  return c->xfn(c->xuser, which, op, ptr, count, NULL, NULL) ? LRH_EDEVICE : LRH_OK;   /* the oracle always fills its own slot */
}
/* two coupled channels through one call: the stage order of include/linrad_hip.h (lrh_set_exchange) with the exchanges made by the
   registered function on host memory */
static int lro_dsp_coupled(lro_ctx *c, lrh_ptrs *p, int nblocks, int batch)
{
  int rc;
  const int C = c->cfg.timf1_frame_channels > 1 ? c->cfg.timf1_frame_channels : 1;
  while (nblocks > 0) {
    const int B = nblocks < batch ? nblocks : batch;
    if ((rc = lro_fft1_b(c, 0, p->timf1p_px, p->fft1_pa, B))) return rc;
    p->timf1p_px = (p->timf1p_px + B * c->M1 * (c->cfg.timf1_dword_input ? 8 : 4) * C) & c->timf1_bytemask;
    p->fft1_pa = (p->fft1_pa + B * 2 * c->N1) & c->fft1_mask;
    p->fft1_na = p->fft1_pa / (2 * c->N1);
    for (int i = 0; i < B; i++) if (p->fft1_nm != c->fft1n_mask) p->fft1_nm++;
    const lrh_ptrs at1 = *p;
    if ((rc = lro_fft1_c(c, p, B))) return rc;
    if (c->corr_on) { size_t ns = 0; if ((rc = lro_fft1_corr_begin(c, &at1, B, &ns)) || (rc = lro_exchange(c, LRH_X_SPEC, LRH_XOP_GATHER, ns)) || (rc = lro_fft1_corr_finish(c, &at1, B))) return rc; }
    if ((rc = lro_make_timf2(c, p, B))) return rc;
    int cnt = 0;
    if ((rc = lro_blanker_begin(c, p, &cnt))) return rc;
    if (cnt > 0) {
      size_t nw = 0;
      if ((rc = lro_exchange(c, LRH_X_PWR, LRH_XOP_SUM, (size_t)cnt)) || (rc = lro_blanker_weak_span(c, &nw))) return rc;
      if (nw && (rc = lro_exchange(c, LRH_X_WEAK, LRH_XOP_GATHER, nw))) return rc;
    }
    if ((rc = lro_first_noise_blanker(c, p))) return rc;
    if (cnt > 0 && ((rc = lro_exchange(c, LRH_X_STAT, LRH_XOP_SUM, 2)) || (rc = lro_blanker_finish(c, p)))) return rc;
    const int avail = ((p->timf2_pn2 - p->timf2_px + 4 * c->cfg.timf2pow_size) & c->timf2_mask);
    int k = avail >= 4 * c->N2 ? 1 + (avail - 4 * c->N2) / (4 * c->M2) : 0;
    while (k > 0) {
      const int kb = k < c->cfg.max_fft2n ? k : c->cfg.max_fft2n;
      const lrh_ptrs at = *p;
      size_t n = 0;
      if ((rc = lro_make_fft2(c, p, kb)) || (rc = lro_fft2_xy_begin(c, &at, kb, &n)) || (rc = lro_exchange(c, LRH_X_BINS, LRH_XOP_GATHER, n)) ||
          (rc = lro_fft2_xy_finish(c, &at, kb)) || (rc = lro_fft2_mix1_fixed(c, p, kb))) return rc;
      if (c->N3 && c->ms.mix1_selfreq >= 0) {
        const int have = (p->timf3_pa - p->timf3_px + c->cfg.timf3_size) & (c->cfg.timf3_size - 1);
        int k3 = have < 2 * c->N3 ? 0 : 1 + (have - 2 * c->N3) / (2 * c->M3);
        const int cap = c->cfg.max_fft3n / 2 > 0 ? c->cfg.max_fft3n / 2 : 1;
        while (k3 > 0) {
          const int k3b = k3 < cap ? k3 : cap;
          size_t np = 0;
          if ((rc = lro_make_fft3_all(c, p, k3b))) return rc;
          if (c->pol_set && ((rc = lro_mix2_pol_begin(c, p, k3b, &np)) || (rc = lro_exchange(c, LRH_X_POL, LRH_XOP_SUM, np)))) return rc;
          if ((rc = lro_fft3_mix2(c, p, k3b))) return rc;
          k3 -= k3b;
        }
      }
      k -= kb;
    }
    nblocks -= B;
  }
  return LRH_OK;
}

int lro_wideband_dsp(lro_ctx *c, lrh_ptrs *p, int nblocks, int batch)
{
  int rc;
  if (c && c->cfg.blanker_channels == 2) return c->xfn ? lro_dsp_coupled(c, p, nblocks, batch) : LRH_ESTATE;
  while (nblocks > 0) {
    int B = nblocks < batch ? nblocks : batch;
    if ((rc = lro_fft1_b(c, 0, p->timf1p_px, p->fft1_pa, B))) return rc;
    p->timf1p_px = (p->timf1p_px + B * c->M1 * (c->cfg.timf1_dword_input ? 8 : 4)) & c->timf1_bytemask;
    p->fft1_pa = (p->fft1_pa + B * 2 * c->N1) & c->fft1_mask;
    p->fft1_na = p->fft1_pa / (2 * c->N1);
    for (int i = 0; i < B; i++) if (p->fft1_nm != c->fft1n_mask) p->fft1_nm++;
    if ((rc = lro_fft1_c(c, p, B))) return rc;
    if (!c->cfg.second_fft_enable) {            /* wcw.c:1049-1081 + narrowband loop: fft1_mix1_fixed per transform */
      if ((rc = lro_fft1_mix1_fixed(c, p, B))) return rc;
      nblocks -= B;
      continue;
    }
    if ((rc = lro_make_timf2(c, p, B))) return rc;
    if ((rc = lro_first_noise_blanker(c, p))) return rc;
    int avail = ((p->timf2_pn2 - p->timf2_px + 4 * c->cfg.timf2pow_size) & c->timf2_mask);
    int k = 0;
    if (avail >= 4 * c->N2) k = 1 + (avail - 4 * c->N2) / (4 * c->M2);
    while (k > 0) {
      int kb = k < c->cfg.max_fft2n ? k : c->cfg.max_fft2n;
      if ((rc = lro_make_fft2(c, p, kb))) return rc;
      if ((rc = lro_fft2_mix1_fixed(c, p, kb))) return rc;
      if (c->N3 && !c->pol_set && c->ms.mix1_selfreq >= 0) {       /* do_fft3 / do_mix2 behind mix1 (fft3.c:35-60, mix2.c:41-80) */
        const int have = (p->timf3_pa - p->timf3_px + c->cfg.timf3_size) & (c->cfg.timf3_size - 1);
        int k3 = have < 2 * c->N3 ? 0 : 1 + (have - 2 * c->N3) / (2 * c->M3);
        const int cap = c->cfg.max_fft3n / 2 > 0 ? c->cfg.max_fft3n / 2 : 1;
        while (k3 > 0) {
          const int k3b = k3 < cap ? k3 : cap;
          if ((rc = lro_make_fft3_all(c, p, k3b))) return rc;
          if ((rc = lro_fft3_mix2(c, p, k3b))) return rc;
          k3 -= k3b;
        }
      }
      k -= kb;
    }
    if (c->wl_on) {                             /* wcw.c:1124-1133 */
      if (p->fft1_liminfo_cnt != c->wl_cnt1) { if ((rc = lro_fft1_update_liminfo(c, p, &c->wl_par))) return rc; c->wl_cnt1 = p->fft1_liminfo_cnt; }
      if (c->wl_fft2 && p->fft2_liminfo_cnt != c->wl_cnt2) { if ((rc = lro_fft2_update_liminfo(c, p, &c->wl_par))) return rc; c->wl_cnt2 = p->fft2_liminfo_cnt; }
    }
    nblocks -= B;
  }
  return LRH_OK;
}

size_t lro_sizeof(int which)
{
  static const size_t sz[] = { sizeof(lrh_config), sizeof(lrh_ptrs), sizeof(lrh_blanker_state), sizeof(lrh_blanker_tables), sizeof(lrh_mix1_state),
                               sizeof(lrh_sellim), sizeof(lrh_spur), sizeof(lrh_afc), sizeof(lrh_synth) };
  return which >= 0 && which < (int)(sizeof sz / sizeof sz[0]) ? sz[which] : 0;
}

int lro_wideband_limiter(lro_ctx *c, const lrh_sellim *par, int fft2_too)
{
  LRO_F32_ONLY(c);
  if (!c) return LRH_EINVAL;
  c->wl_on = 0;
  if (!par) return LRH_OK;
  if (par->struct_size != (int)sizeof *par) return LRH_EINVAL;
  c->wl_par = *par; c->wl_on = 1; c->wl_fft2 = fft2_too != 0; c->wl_cnt1 = 0; c->wl_cnt2 = 0;
  return LRH_OK;
}

int lro_export(lro_ctx *c, lrh_ring ring, void *dst, size_t off, size_t cnt)
{
  const void *src; size_t esz = 4, total;
  switch (ring) {
    case LRH_RING_TIMF1: src = c->timf1; esz = 2; total = c->cfg.timf1_bytes / 2; break;
    case LRH_RING_FFT1_FLOAT: src = c->fft1_float; total = (size_t)c->cfg.max_fft1n * 2 * c->N1; break;
    case LRH_RING_FFT1_SUMSQ: src = c->fft1_sumsq; total = c->cfg.fft1_sumsq_bufsize; break;
    case LRH_RING_FFT1_SLOWSUM: src = c->fft1_slowsum; total = c->N1; break;
    case LRH_RING_TIMF2_FLOAT: src = c->timf2_float; total = 4 * (size_t)c->cfg.timf2pow_size; break;
    case LRH_RING_TIMF2_PWR: src = c->timf2_pwr; total = c->cfg.timf2pow_size; break;
    case LRH_RING_FFT2_FLOAT: src = c->fft2_float; total = (size_t)c->cfg.max_fft2n * 2 * c->N2; break;
    case LRH_RING_FFT2_POWER: src = c->fft2_power; total = (size_t)c->cfg.max_fft2n * c->N2; break;
    case LRH_RING_FFT2_POWERSUM: src = c->fft2_powersum; total = c->N2; break;
    case LRH_RING_WG_WATERF: src = c->wg_waterf; esz = 2; total = (size_t)c->cfg.wf_lines * c->cfg.wf_xpixels; break;
    case LRH_RING_TIMF3_FLOAT: src = c->timf3_float; total = c->cfg.timf3_size; break;
    case LRH_RING_TIMF2_BLOCKPOWER: src = c->timf2_blockpower; total = c->cfg.timf2_blockpower_size; break;
    case LRH_RING_FFT3: src = c->fft3; total = (size_t)c->cfg.max_fft3n * 2 * c->N3; break;
    case LRH_RING_BASEB_RAW: src = c->baseb_raw; total = 2 * (size_t)c->cfg.baseband_size; break;
    case LRH_RING_FFT2_XYPOWER: if (!c->fft2_xypower) return LRH_ESTATE; src = c->fft2_xypower; total = (size_t)c->cfg.max_fft2n * 4 * c->N2; break;
    case LRH_RING_FFT2_XYSUM: if (!c->fft2_xysum) return LRH_ESTATE; src = c->fft2_xysum; total = 4 * (size_t)c->N2; break;
    case LRH_RING_FFT1_CORRSUM: if (!c->corr_on) return LRH_ESTATE; src = c->fft1_corrsum; total = 2 * (size_t)c->cfg.fft1_sumsq_bufsize; break;
    case LRH_RING_FFT1_SLOWCORR: if (!c->corr_on) return LRH_ESTATE; src = c->fft1_slowcorr; total = 2 * (size_t)c->N1; break;
    case LRH_RING_FFT1_SLOWCORR_TOT: if (!c->corr_on) return LRH_ESTATE; src = c->fft1_slowcorr_tot; total = 2 * (size_t)c->N1; esz = 8; break;
    default: return LRH_EINVAL;
  }
  if (off + cnt > total) return LRH_EINVAL;
  if (esz == 4) { out_f32((f32 *)dst, (const float *)src + off, cnt); return LRH_OK; }      /* the float rings: converted in the float64 build */
  memcpy(dst, (const char *)src + off * esz, cnt * esz);
  return LRH_OK;
}
#ifdef LRO_F64
int lro_export_wf_pre(lro_ctx *c, double *dst, size_t off, size_t cnt)
{
  if (!c || !dst || off + cnt > (size_t)c->cfg.wf_lines * c->cfg.wf_xpixels) return LRH_EINVAL;
  memcpy(dst, c->wf_pre + off, cnt * sizeof(double)); return LRH_OK;
}
/* the float64 build's rings unrounded (float rings only) */
int lro_export_f64(lro_ctx *c, lrh_ring ring, double *dst, size_t off, size_t cnt)
{
  const double *src; size_t total;
  switch (ring) {
    case LRH_RING_FFT1_FLOAT: src = c->fft1_float; total = (size_t)c->cfg.max_fft1n * 2 * c->N1; break;
    case LRH_RING_FFT1_SUMSQ: src = c->fft1_sumsq; total = c->cfg.fft1_sumsq_bufsize; break;
    case LRH_RING_FFT1_SLOWSUM: src = c->fft1_slowsum; total = c->N1; break;
    case LRH_RING_TIMF2_FLOAT: src = c->timf2_float; total = 4 * (size_t)c->cfg.timf2pow_size; break;
    case LRH_RING_TIMF2_PWR: src = c->timf2_pwr; total = c->cfg.timf2pow_size; break;
    case LRH_RING_FFT2_FLOAT: src = c->fft2_float; total = (size_t)c->cfg.max_fft2n * 2 * c->N2; break;
    case LRH_RING_FFT2_POWER: src = c->fft2_power; total = (size_t)c->cfg.max_fft2n * c->N2; break;
    case LRH_RING_FFT2_POWERSUM: src = c->fft2_powersum; total = c->N2; break;
    case LRH_RING_TIMF3_FLOAT: src = c->timf3_float; total = c->cfg.timf3_size; break;
    case LRH_RING_TIMF2_BLOCKPOWER: src = c->timf2_blockpower; total = c->cfg.timf2_blockpower_size; break;
    case LRH_RING_FFT3: src = c->fft3; total = (size_t)c->cfg.max_fft3n * 2 * c->N3; break;
    case LRH_RING_BASEB_RAW: src = c->baseb_raw; total = 2 * (size_t)c->cfg.baseband_size; break;
    default: return LRH_EINVAL;
  }
  if (off + cnt > total) return LRH_EINVAL;
  memcpy(dst, src + off, cnt * sizeof(double));
  return LRH_OK;
}
#endif

int lro_get_blanker_state(lro_ctx *c, lrh_blanker_state *st) { *st = c->bs; return LRH_OK; }

/* NET_RXOUT_TIMF2 payload, float form: rxin.c:949-956 with twice_rxchan = 2 */
int lro_export_timf2_net(lro_ctx *c, f32 *dst, int timf2_pt, int count, f32 gain, f32 strong)
{
  if (count < 0 || count > c->cfg.timf2pow_size || (timf2_pt & 3)) return LRH_EINVAL;
  int pt = timf2_pt & c->timf2_mask;
  for (int i = 0; i < count; i++) {
    const float *zb = c->timf2_float + pt;
    for (int nn = 0; nn < 2; nn++) dst[2 * i + nn] = gain * (zb[nn] + strong * zb[2 + nn]);
    pt = (pt + 4) & c->timf2_mask;
  }
  return LRH_OK;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Selective limiter: fft1_update_liminfo (sellim.c:738-1157) followed by selfreq_liminfo (sellim.c:38-157), float path,
 * one RF channel, uncalibrated amplitude (the calibrated liminfo_amplitude_factor belongs to the clever blanker).
 * State kept between calls like the reference's globals: liminfo[], old_liminfo[] (second half of the reference's
 * array), liminfo_wait[], fft1_sumsq_tot, sel_ia / sel_ib.
 * ------------------------------------------------------------------------------------------------------------------ */
#define LRO_BIGFLOAT 300000000000000000000000000000000000000.F
#define LRO_RELEASE_FACTOR 1.15
#define LRO_SFAC 2.
typedef struct { float *old; unsigned char *wait; float *tmp, *group_min; int sumsq_tot, sel_ia, sel_ib; float *ftmp;
                 float *reg_noise; int *reg_first, *reg_len; } lro_sellim_state;   /* reg_*: variant 1's region list, kept between calls like the reference's (buf.c:976-980) */
static lro_sellim_state *sellim_state(lro_ctx *c)
{
  if (!c->sellim) {                       /* allocated on first use, freed by lro_close */
    lro_sellim_state *s = calloc(1, sizeof *s);
    s->old = calloc(c->N1, 4); s->wait = calloc(c->N1, 1); s->group_min = calloc(c->N1 + 4, 4);
    s->tmp = (float *)calloc(c->N1 + 16, 4) + 8;      /* the reference's scans look two bins below and above their range */
    s->ftmp = (float *)calloc(c->N1 + 16, 4) + 8;     /* fftf_tmp of fft2_update_liminfo: zero outside what it fills (buf.c:972) */
    s->reg_noise = calloc(c->N1 + 8, 4); s->reg_first = calloc(c->N1 + 8, sizeof(int)); s->reg_len = calloc(c->N1 + 8, sizeof(int));
    c->sellim = s;
  }
  return (lro_sellim_state *)c->sellim;
}

/* minimum of a group as the mean of its three smallest values (sellim.c:886-915) */
static float three_smallest(const float *v, int ia, int ib)
{
  float t1 = LRO_BIGFLOAT, t2 = LRO_BIGFLOAT, t3 = LRO_BIGFLOAT;
  for (int i = ia; i < ib; i++) {
    const float x = v[i];
    if (x <= t3) {
      if (x <= t1) { t3 = t2; t2 = t1; t1 = x; }
      else if (x <= t2) { t3 = t2; t2 = x; }
      else t3 = x;
    }
  }
  return (float)(0.3333333 * (t1 + t2 + t3));
}

static void selfreq_liminfo(lro_ctx *c, lro_sellim_state *st, const lrh_sellim *q)
{
  float *lim = c->liminfo;
  const int N = c->N1;
  if (c->ms.mix1_selfreq >= 0) {
    int ia = (int)(c->ms.mix1_selfreq * c->cfg.fftx_points_per_hz);
    int k = (int)(q->baseband_bw_fftxpts * .7);
    if (q->sellim_par6 == 0) k += 3;
    if (c->cfg.second_fft_enable) {
      int ratio = c->N2 / c->N1; if (ratio < 1) ratio = 1;
      ia /= ratio; k /= ratio; if (k < 3) k = 3;
    }
    int ib = ia + k; ia -= k;
    if (ia < 0) ia = 0;
    if (ib >= N) ib = N - 1;
    st->sel_ia = ia; st->sel_ib = ib;
    if (q->ston_scale) { for (int i = ia; i <= ib; i++) lim[i] = -1; }
    else {
      float t1 = 0, t2 = 2;
      for (int i = ia; i <= ib; i++) { if (lim[i] < 0) t1 = 1; if (lim[i] > 0 && t2 > lim[i]) t2 = lim[i]; }
      int skip = 0;
      if (t2 > 1) { if (t1 == 0) skip = 1; t2 = 1; }
      if (!skip) {
        t1 = 1 / t2;
        t1 *= (float)sqrt((float)(c->N2 / c->N1));
        if (q->sellim_par5 == 2) t2 = -1;
        if (q->sellim_par5 == 1) { if (t1 < 0x7fff / q->sellim_maxlevel) t2 = 0; }
        if (q->sellim_par5 == 0) { if (t1 < 0x7ffff / q->sellim_maxlevel) t2 = 0; }
        for (int i = ia; i <= ib; i++) lim[i] = t2;
      }
    }
  }
  /* liminfo_amplitude_factor (sellim.c:108-155): what the strong bins take away from a pulse's amplitude */
  if (!c->cfg.second_fft_enable) c->amp_factor = 1;
  else if (q->fft1_desired) {
    float t1 = 0, tot = 0;
    for (int i = 0; i < N; i++) tot += q->fft1_desired[i] * q->fft1_desired[i];       /* fft1_desired_totsum (calibration set-up) */
    for (int i = q->fft1_first_point; i <= q->fft1_last_point; i++) if (lim[i] != 0) t1 += q->fft1_desired[i] * q->fft1_desired[i];
    c->amp_factor = tot / (tot - t1);
  } else {
    int k = 0;
    for (int i = q->fft1_first_inband; i <= q->fft1_last_inband; i++) if (lim[i] != 0) k++;
    const int n = q->fft1_last_inband - q->fft1_first_inband + 1;
    c->amp_factor = (float)(n) / (n - k);
  }
  if (c->amp_factor > 2) c->amp_factor = 0;
  for (int i = 0; i < N; i++) st->old[i] = lim[i];
}

int lro_get_liminfo_amplitude_factor(lro_ctx *c, f32 *f) { if (!c || !f) return LRH_EINVAL; *f = c->amp_factor; return LRH_OK; }
int lro_set_liminfo_amplitude_factor(lro_ctx *c, f32 f) { if (!c) return LRH_EINVAL; c->amp_factor = f; return LRH_OK; }

/* hold-off count of the second limiter (sellim.c:207-209, 284-286, 536-538) */
static unsigned sellim2_wait_n(const lro_ctx *c, const lrh_sellim *q)
{
  unsigned w = 1 + (1 + (q->fft2_blocktime * c->cfg.waterfall_avgnum)) / (c->cfg.fft_avg1num * q->fft1_blocktime);
  return w > 255 ? 255 : w;
}
static void mark_strong(float *lim, unsigned char *wait, int i, unsigned wait_n) { lim[i] = -1; wait[i] = (unsigned char)wait_n; }

/* hg.sellim_par1 = 0 (sellim.c:170-281): noise floor = the median of every fft2 bin's power (the reference sorts the lower half by
   selection; any exact selection gives the same value), band edges where the spectrum stays below 2 % of it, and every fft1 bin with an
   fft2 bin above blanker_ston_fft2 * median.  One channel (sw_onechan). */
static int cmp_float(const void *a, const void *b) { const float x = *(const float *)a, y = *(const float *)b; return x < y ? -1 : x > y; }
static void fft2_liminfo_median(lro_ctx *c, lro_sellim_state *st, const lrh_sellim *q)
{
  const int N = c->N1, N2 = c->N2, nn = N2 / N;
  float *lim = c->liminfo;
  float *f = malloc(sizeof(float) * (size_t)N2), *srt = malloc(sizeof(float) * (size_t)N2);
  for (int i = 0; i < N; i++) for (int j = nn * i; j < nn * i + nn; j++) f[j] = c->fft2_powersum[j] * c->wg_waterf_yfac[i];
  memcpy(srt, f, 4 * (size_t)N2);
  qsort(srt, N2, 4, cmp_float);
  const float median = srt[N2 / 2 - 1];
  const unsigned wait_n = sellim2_wait_n(c, q);
  const float edge = median * 0.02F;
  int ia = nn * q->fft1_first_point;
  while (ia < N2 - 1 && f[ia] < edge) ia++;                       /* (the reference's scan has no end: a spectrum entirely below the limit is not a case) */
  int first = (ia + nn / 2) / nn;
  int ib = (nn + 1) * q->fft1_last_point;                          /* as written (sellim.c:239) */
  if (ib > N2) ib = N2;
  while (ib > 1 && f[ib - 1] < edge) ib--;
  int last = (ib + nn / 2) / nn;
  if (first < 5) first = 5;
  if (last > N - 6) last = N - 6;
  for (int i = 0; i < first; i++) mark_strong(lim, st->wait, i, wait_n);
  for (int i = last; i < N; i++) mark_strong(lim, st->wait, i, wait_n);
  const float limit = q->blanker_ston_fft2 * median;
  for (int i = first; i < last; i++) {
    int k = 0;
    for (int j = nn * i; j < nn * i + nn; j++) if (f[j] > limit) k++;
    if (k > 0) mark_strong(lim, st->wait, i, wait_n);
  }
  free(f); free(srt);
}

/* hg.sellim_par1 = 1 (sellim.c:283-533): every stretch of six or more bins that are not attenuated (liminfo <= 0) between attenuated
   carriers is a weak-signal region with a noise floor of its own; the list of regions (noise, first point, length) lives between calls
   like the reference's arrays, and the two clean-up loops that run when the list is nearly full keep the reference's indexing
   (sellim.c:424: the length of entry i bounds the loop over i; sellim.c:447-460: the count drops every pass). */
static void region_list_drop(lro_sellim_state *st, int k, int n)
{
  for (int j = k + 1; j < n; j++) { st->reg_noise[j - 1] = st->reg_noise[j]; st->reg_first[j - 1] = st->reg_first[j]; st->reg_len[j - 1] = st->reg_len[j]; }
}
static float region_mean_noise(const lro_sellim_state *st, int n, int skip_negative)
{
  int k = 0; float t = 0;
  for (int i = 0; i < n; i++) { if (skip_negative && st->reg_noise[i] < 0) continue; k += st->reg_len[i]; t += st->reg_noise[i] * st->reg_len[i]; }
  return t / k;
}
static void fft2_liminfo_regions(lro_ctx *c, lro_sellim_state *st, const lrh_sellim *q)
{
  const int N = c->N1, nn = c->N2 / N, G = N / q->liminfo_group_points, last = q->fft1_last_point;
  const int in_lo = q->fft1_first_inband, in_hi = q->fft1_last_inband;
  float *lim = c->liminfo, *f = st->ftmp;
  const unsigned wait_n = sellim2_wait_n(c, q);
  const float ston = q->blanker_ston_fft2;
  int reg_no = 0, ia = q->fft1_first_point;
  for (;;) {
    while (lim[ia] > 0 && ia < last) ia++;
    if (ia == last) break;
    int ib = ia;
    while (lim[ib] <= 0 && ib < last) ib++;
    if (ib - ia < 6) { ia = ib; continue; }                         /* too short to tell anything (sellim.c:297-300: ib--, then ia = ib + 1) */
    ia++; ib--;                                                      /* the blanker's own noise next to the carriers stays out */
    float lowest = LRO_BIGFLOAT;
    for (int i = ia; i < ib; i++) {
      float t = 0;
      for (int j = nn * i; j < nn * i + nn; j++) t += c->fft2_powersum[j];
      t *= c->wg_waterf_yfac[i];
      f[i] = t;
      if (t < lowest && i >= in_lo && i <= in_hi) lowest = t;
    }
    f[ia - 1] = f[ia]; f[ib] = f[ib - 1];
    float limit = lowest;
    limit *= 2 * (1 + 2. / c->cfg.waterfall_avgnum);
    int ja = ia < in_lo ? in_lo : ia, jb = ib > in_hi ? in_hi + 1 : ib;
    float sum = 0; int cnt = 0;                                      /* not cleared when the limit is widened (sellim.c:356-371) */
    for (;;) {
      for (int i = ja; i < jb; i++) if (f[i] < limit) { cnt++; sum += f[i]; }
      if (cnt == 0 || cnt >= (jb - ja) / 4) break;
      limit *= 3;
    }
    if (cnt != 0) {
      const float floor_ = sum / cnt;
      st->reg_noise[reg_no] = floor_; st->reg_first[reg_no] = ia - 1;
      const float over = floor_ * ston;
      if (getenv("LRO_SELLIM_DEBUG")) { const int b_ = atoi(getenv("LRO_SELLIM_DEBUG")); if (b_ >= ia && b_ < ib)   /* (test diagnostics: how far a bin is from the decisions) */
        { float near_ = 1e9f; int at_ = -1; for (int i = ja; i < jb; i++) if (fabsf(f[i] / limit - 1) < near_) { near_ = fabsf(f[i] / limit - 1); at_ = i; }
          fprintf(stderr, "LRO_SELLIM_DEBUG region [%d,%d) lowest %.9g limit %.9g cnt %d floor %.9g over %.9g f[%d] %.9g ratio-1 %.3e; nearest to the floor's limit: bin %d at %.3e\n", ia, ib, lowest, limit, cnt, floor_, over, b_, f[b_], f[b_] / over - 1, at_, near_); } }
      int ja2 = -1, jb2 = 0;
      for (int i = ia; i < ib; i++) if (f[i] > over) { if (ja2 < 0) ja2 = i; jb2 = i; mark_strong(lim, st->wait, i, wait_n); }
      if (ja2 < 0) st->reg_len[reg_no++] = ib - ia + 1;
      else {                                                         /* the quiet parts below the first and above the last bin taken out */
        st->reg_len[reg_no++] = ja2 - ia + 1;
        if (ib - jb2 > 4) { st->reg_noise[reg_no] = floor_; st->reg_first[reg_no] = jb2 + 1; st->reg_len[reg_no] = ib - jb2; reg_no++; }
      }
      if (reg_no >= G - 2) {                                         /* the list is nearly full: sellim.c:406-469 */
        float t1 = region_mean_noise(st, reg_no, 0) * ston;
        for (int k = 0; k < reg_no; k++)
          if (st->reg_noise[k] > t1) {
            for (int i = 0; i < G + 8 && i < st->reg_len[i]; i++) {  /* entry i's own length ends the loop (sellim.c:424); the list has G entries (+8 here) */
              const int b = i + st->reg_first[k];
              if (b >= 0 && b < N) mark_strong(lim, st->wait, b, wait_n);
            }
            region_list_drop(st, k, reg_no);
            reg_no--;                                                /* and k moves on: the entry that slid into place is not looked at */
          }
        if (reg_no >= 3 * G / 4) {
          t1 /= ston;
          while (reg_no > 0) {                                       /* sellim.c:447-460: k stays at 0 and the count drops every pass */
            if (st->reg_noise[0] < t1) region_list_drop(st, 0, reg_no);
            reg_no--;
          }
        }
      }
    }
    ia = ib + 1;
  }
  if (reg_no == 0) return;
  float t1 = region_mean_noise(st, reg_no, 0) * ston;
  int dropped = 0;
  for (int k = 0; k < reg_no; k++)
    if (st->reg_noise[k] > t1) {                                     /* a whole region above the common floor */
      dropped = 1;
      for (int i = 0; i < st->reg_len[k]; i++) mark_strong(lim, st->wait, i + st->reg_first[k], wait_n);
      st->reg_noise[k] = -1;
    }
  if (dropped) {
    t1 = region_mean_noise(st, reg_no, 1);                           /* sum and count taken before the list is compacted: same numbers */
    for (int i = 0; i < reg_no; i++) if (st->reg_noise[i] < 0) { region_list_drop(st, i, reg_no); i--; reg_no--; }
    t1 *= ston;
  }
  if (getenv("LRO_SELLIM_DEBUG")) { const int b_ = atoi(getenv("LRO_SELLIM_DEBUG")); fprintf(stderr, "LRO_SELLIM_DEBUG common limit %.9g f[%d] %.9g ratio-1 %.3e regions %d dropped %d\n", t1, b_, f[b_], f[b_] / t1 - 1, reg_no, dropped); }
  for (int k = 0; k < reg_no; k++)
    for (int i = 0; i < st->reg_len[k]; i++) if (f[i + st->reg_first[k]] > t1) mark_strong(lim, st->wait, i + st->reg_first[k], wait_n);
}

/* fft2_update_liminfo, sellim.c:159-736: hg.sellim_par1 = 2 (535-731) below, 0 and 1 above */
int lro_fft2_update_liminfo(lro_ctx *c, lrh_ptrs *p, const lrh_sellim *q)
{
  LRO_F32_ONLY(c);
  if (!c || !p || !q || q->struct_size != (int)sizeof *q) return LRH_EINVAL;
  if (!c->cfg.second_fft_enable || c->cfg.blanker_channels == 2) return LRH_EINVAL;
  if (q->sellim_par1 < 0 || q->sellim_par1 > 2) return LRH_EINVAL;
  if (q->sellim_par1 != 2) {
    if (q->liminfo_group_points < 1 || c->N2 < c->N1) return LRH_EINVAL;
    lro_sellim_state *s = sellim_state(c);
    if (q->sellim_par1 == 0) fft2_liminfo_median(c, s, q); else fft2_liminfo_regions(c, s, q);
    selfreq_liminfo(c, s, q);
    return LRH_OK;
  }
  const int N = c->N1, nn = c->N2 / c->N1, gp = q->liminfo_group_points;
  if (nn < 1 || gp < 1 || N / gp < 1) return LRH_EINVAL;
  lro_sellim_state *st = sellim_state(c);
  float *lim = c->liminfo, *f = st->ftmp;
  const int groups = N / gp;
  float *reg_min = calloc(3 * (size_t)groups + 3, 4), *reg_ston = reg_min + groups + 1, *reg_noise = reg_ston + groups + 1;
  unsigned wait_n = 1 + (1 + (q->fft2_blocktime * c->cfg.waterfall_avgnum)) / (c->cfg.fft_avg1num * q->fft1_blocktime);
  if (wait_n > 255) wait_n = 255;
  int i, j, k, ia, ib;
  float t1, t2, t3, global_noise_floor;
  for (i = q->fft1_first_point; i < q->fft1_last_point; i++) {       /* mean fft2 power over the width of an fft1 bin */
    t1 = 0;
    for (j = nn * i; j < nn * i + nn; j++) t1 += c->fft2_powersum[j];
    f[i] = t1 * c->wg_waterf_yfac[i] / nn;
  }
  for (i = 0; i < groups; i++) {
    ia = i * gp; ib = ia + gp;
    t1 = 0;
    for (j = ia; j < ib; j++) t1 += f[j];
    t1 /= gp;
    k = 0; t3 = LRO_BIGFLOAT; t1 *= 0.001F; t2 = 0;
    for (j = ia; j < ib; j++) if (f[j] > t1) { k++; if (f[j] < t3) t3 = f[j]; if (f[j] > t2) t2 = f[j]; }
    reg_min[i] = k < 2 ? -1 : t3;
    reg_ston[i] = t2 / t3;
  }
  t1 = 0; k = 0;
  for (i = 0; i < groups; i++) if (reg_ston[i] < 2000.F) { t1 += reg_min[i]; k++; }
  if (k == 0) { free(reg_min); return LRH_OK; }                      /* sellim.c:604: returns without selfreq_liminfo */
  t1 /= k;
  global_noise_floor = 0; k = 0;
  for (i = 0; i < groups; i++) if (reg_min[i] > 0.03F * t1 && reg_min[i] < 30.F * t1) { global_noise_floor += reg_min[i]; k++; }
  if (k < 3) goto done;
  global_noise_floor /= k;
  t1 = 5 * global_noise_floor;
  for (i = 0; i < groups; i++) {
    ia = i * gp; ib = ia + gp; t2 = 0; k = 0;
    for (j = ia; j < ib; j++) if (f[j] < t1) { k++; t2 += f[j]; }
    reg_noise[i] = k > 2 ? t2 / k : -1;
  }
  t1 = 0; k = 0;
  for (i = 0; i < groups; i++) if (reg_noise[i] > 0) { k++; t1 += reg_noise[i]; }
  if (k < 3) goto done;
  t1 /= k;
  global_noise_floor = 0; k = 0;
  for (i = 0; i < groups; i++) if (reg_noise[i] > 0.1F * t1 && reg_noise[i] < 10.F * t1) { global_noise_floor += reg_noise[i]; k++; }
  if (k < 3) goto done;
  global_noise_floor /= k;
  t1 = 0.5 * q->blanker_ston_fft2 * global_noise_floor;
  ia = q->fft1_first_point; if (ia < 2) ia = 2;
  ib = q->fft1_last_point; if (ib < N - 2) ib = N - 2;              /* as written (sellim.c:668-669): at least N-2 */
  k = 0;
  for (i = ia; i < ib; i++) {                                        /* noise the stupid blanker leaves next to strong bins */
    if (lim[i - 1] == 0 && lim[i] != 0) { if (f[i - 1] > f[i - 2]) f[i - 1] = f[i - 2]; }
    if (lim[i + 1] == 0 && lim[i] != 0) { if (f[i + 1] > f[i + 2]) f[i + 1] = f[i + 2]; }
    if (lim[i] != 0) k++;
  }
  if (k > (ib - ia) / 4) {                                           /* a quarter of the band strong: thin the table */
    k = 0;
    for (i = ia; i < ib; i++) { if (lim[i] < 0 && f[i] < t1) { lim[i] = 0; st->wait[i] = 0; } if (lim[i] != 0) k++; }
    if (k > (ib - ia) / 4) {
      t2 = 10.F * t1;
      for (i = ia; i < ib; i++) if (lim[i] < 0 && f[i] < t2) { lim[i] = 0; st->wait[i] = 0; }
    }
  }
  for (i = ia; i < ib; i++)                                           /* the fifth term repeats i-2 in the reference (sellim.c:723) */
    if (LRO_SFAC * f[i - 2] > t1 || f[i - 1] > t1 || f[i] > t1 || f[i + 1] > t1 || LRO_SFAC * f[i - 2] > t1) {
      if (lim[i] == 0) lim[i] = -1;
      st->wait[i] = (unsigned char)wait_n;
    }
done:
  free(reg_min);
  selfreq_liminfo(c, st, q);
  return LRH_OK;
}

int lro_fft1_update_liminfo(lro_ctx *c, lrh_ptrs *p, const lrh_sellim *q)
{
  LRO_F32_ONLY(c);
  if (!c || !p || !q || q->struct_size != (int)sizeof *q) return LRH_EINVAL;
  lro_sellim_state *st = sellim_state(c);
  const int N = c->N1, avg1 = c->cfg.fft_avg1num;
  float *lim = c->liminfo, *old = st->old, *tmp = st->tmp, *gmin = st->group_min;
  /* the block at the *advanced* pointer (sellim.c:788, fft1.c:4519); in the middle of a period (batched rounds: the reference never
     looks then) that slot holds the unfinished sums of the period in progress, and the newest finished period is taken instead */
  const float *sumsq = c->fft1_sumsq + ((p->fft1_sumsq_pa - (p->fft1_sumsq_counter ? N : 0) + c->cfg.fft1_sumsq_bufsize) & c->fft1_sumsq_mask);
  const float *slow = c->fft1_slowsum, *yfac = c->wg_waterf_yfac;
  const int par7 = q->sellim_par7;
  st->sumsq_tot += avg1;
  if (st->sumsq_tot > q->spek_avgnum) st->sumsq_tot = q->spek_avgnum;
  int k = (int)(1 + 1 / (avg1 * q->fft1_blocktime));
  const unsigned int wait_n = k < 255 ? (unsigned)k : 255u;
  float t1 = (float)q->sellim_maxlevel, t2, limit;
  limit = t1 * t1 * avg1 * 1;
  limit *= N; limit /= c->N2;
  int ia = q->fft1_first_point; const int ix = ia, iy = q->fft1_last_point - 1;
  /* pass 1: strong narrow-band signals get a common attenuation over their whole width, with tapered skirts */
  do {
    if (sumsq[ia] > limit) {
      float maxval = sumsq[ia];
      int ib = ia + 1;
      while (sumsq[ib] > limit && ib <= iy) { if (sumsq[ib] > maxval) maxval = sumsq[ib]; ib++; }
      while (ia > ix && sumsq[ia - 1] / sumsq[ia] < 0.3) ia--;
      while (ib < iy && sumsq[ib + 1] / sumsq[ib] < 0.3) ib++;
      int ja = ia, jb = ib;
      t1 = lim[ja];
      for (int j = ja + 1; j <= jb; j++) if (lim[j] > 0 && lim[j] < t1) t1 = lim[j];
      t2 = (float)sqrt(limit / maxval);
      if (t1 / t2 > 0.1 && t1 / t2 < 10) t2 = (float)(0.8 * t1 + 0.2 * t2);
      if (ja > st->sel_ib || jb < st->sel_ia || par7 == 0)
        for (int j = ja; j <= jb; j++) if (j > st->sel_ib || j < st->sel_ia || par7 == 0) lim[j] = t2;
      t1 = t2;
      int j = 1 + (ib - ia) / 4;
      while (ia > ix && j > 0) {
        j--; ia--; ja = ia;
        t1 = (float)pow(t1, 0.9);
        if (lim[ja] <= 0 || lim[ja] > t1) { if (ja > st->sel_ib || ja < st->sel_ia || par7 == 0) lim[ja] = t1; }
        else break;
      }
      j = 1 + (ib - ia) / 4;
      while (ib < iy && j > 0) {
        j--; ib++; jb = ib;
        t2 = (float)pow(t2, 0.9);
        if (lim[jb] <= 0 || lim[jb] > t1) lim[jb] = t2;
        else break;
      }
      ia = ib;
    } else {
      if (ia > st->sel_ib || ia < st->sel_ia || par7 == 0) lim[ia] = 0;
    }
    ia++;
  } while (ia < iy);
  if (st->sumsq_tot >= q->spek_avgnum) {
    /* pass 2: everything that rises out of the noise floor of the slow average goes to the strong path (liminfo = -1) */
    const int gp = q->liminfo_group_points;
    int ja = q->fft1_first_inband / gp, jb, ib;
    if (q->sellim_par2 == 0) {
      jb = 1 + q->fft1_last_inband / gp;
      if ((jb - ja) * gp > N) jb--;
      ia = ja * gp; ib = ia + gp;
      for (int j = ja; j < jb; j++) {
        for (int i = ia; i < ib; i++) tmp[i] = yfac[i] * slow[i];
        gmin[j] = three_smallest(tmp, ia, ib);
        ia += gp; ib += gp;
      }
    } else {
      jb = ja + 1; ib = jb * gp;
      for (int i = 0; i < ib; i++) tmp[i] = yfac[i] * slow[i];
      gmin[ja] = three_smallest(tmp, q->fft1_first_inband, ib);
      do {
        ia = ib; ib += gp;
        if (ib > q->fft1_last_inband) ib = q->fft1_last_inband + 1;
        for (int i = ia; i < ib; i++) tmp[i] = yfac[i] * slow[i];
        gmin[jb] = three_smallest(tmp, ia, ib);
        jb++;
      } while (ib < q->fft1_last_inband);
      for (int i = ib; i < q->fft1_last_point; i++) tmp[i] = yfac[i] * slow[i];
    }
    t1 = 0;
    for (int j = ja; j < jb; j++) t1 += gmin[j];
    t1 /= jb - ja;
    k = 0;
    float noise_floor = 0;
    t1 *= (float)(2 * (1 + 2. / q->spek_avgnum));
    for (int j = ja; j < jb; j++) if (gmin[j] < t1) { noise_floor += gmin[j]; k++; }
    if (q->sellim_par3 == 1) {
      t2 = (float)(0.05 * noise_floor / k);
      int first_group = ja, last_group = jb;
      while (gmin[first_group] < t2) first_group++;
      while (gmin[last_group - 1] < t2) last_group--;
      if (first_group != ja || last_group != jb) {
        k = 0; noise_floor = 0;
        for (int j = first_group; j < last_group; j++) if (gmin[j] < t1) { noise_floor += gmin[j]; k++; }
      }
    }
    if (noise_floor < 0.0001) noise_floor = 0.0001f;
    if (k != 0) {
      noise_floor *= (float)((1 + 2. / q->spek_avgnum) / k);
      noise_floor *= q->blanker_ston_fft1;
      ia = 0;
      while (ia < q->fft1_first_point || ia < 2) { if (lim[ia] == 0) lim[ia] = -1; ia++; }
      while (tmp[ia] > noise_floor && ia < N) { if (lim[ia] == 0) lim[ia] = -1; ia++; }
      t1 = q->sellim_par4 == 0 ? 4.F : 3.F;
      while (t1 * tmp[ia + 1] < tmp[ia] && ia < N) { ia++; if (lim[ia] == 0) lim[ia] = -1; }
      for (;;) {
        while (tmp[ia] <= noise_floor && ia < q->fft1_last_point) ia++;
        if (ia >= q->fft1_last_point) break;
        ib = ia;
        if (lim[ia] == 0) lim[ia] = -1;
        while ((LRO_SFAC * tmp[ib - 1] < tmp[ib] || LRO_SFAC * LRO_SFAC * tmp[ib - 2] < tmp[ib]) && ib > q->fft1_first_point) {
          ib--; if (lim[ib] == 0) lim[ib] = -1;
        }
        while (tmp[ia + 1] > noise_floor && ia < q->fft1_last_point) { ia++; if (lim[ia] == 0) lim[ia] = -1; }
        if (ia != q->fft1_last_point) {
          while ((LRO_SFAC * tmp[ia + 1] < tmp[ia] || LRO_SFAC * LRO_SFAC * tmp[ia + 2] < tmp[ia]) && ia < q->fft1_last_point) {
            ia++; if (lim[ia] == 0) lim[ia] = -1;
          }
          ia++;
        }
        if (ia >= q->fft1_last_point) break;
      }
    }
    if (ia > N - 2) ia = N - 2;
    if (q->sellim_par8 == 0) { while (ia < N) { lim[ia] = -1; ia++; } }
    else { while (ia < N) { if (lim[ia] == 0) lim[ia] = -1; ia++; } }
    /* hold-off: a bin stays with the strong signals for about a second after it was last classified, and a gain is
       released slowly (sellim.c:1121-1147) */
    for (int i = 0; i < N; i++) {
      if (lim[i] != 0) st->wait[i] = (unsigned char)wait_n;
      else {
        if (st->wait[i] > 0) st->wait[i]--;
        if (st->wait[i] > 0) lim[i] = -1;
      }
      if (old[i] > 0) {
        t1 = (float)(old[i] * LRO_RELEASE_FACTOR);
        if (t1 < 1) { if (lim[i] > 0 && lim[i] > t1) lim[i] = t1; }
      }
    }
  }
  selfreq_liminfo(c, st, q);
  lim[0] = 0; lim[1] = 0; lim[N - 2] = 0; lim[N - 1] = 0;
  return LRH_OK;
}

int lro_get_liminfo(lro_ctx *c, f32 *dst) { if (!c || !dst) return LRH_EINVAL; out_f32(dst, c->liminfo, c->N1); return LRH_OK; }

/* ------------------------------------------------------------------------------------------------------------------
 * Spur subtraction: eliminate_spurs (spur.c:36-494) for locked spurs, one channel, float spectra, with
 * refine_pll_parameters (spur.c:634-680), spur_phase_parameters (spur.c:1427-1652), complex_lowpass / remove_phasejumps /
 * average_slope (spursub.c:942-1068) and shift_spur_table (spursub.c:1070-1246).  Acquisition and re-lock stay with the
 * control plane (lro_spur_set hands the loop state over; see include/linrad_hip.h).
 * ------------------------------------------------------------------------------------------------------------------ */
#define SPW LRH_SPUR_WIDTH
#define SPSZ 8
#define NSPEC 256
typedef struct {
  int max, n, speknum, avgnum, numsub;
  float freq_factor, max_d2, minston, weiold, weinew, linefit;
  float spectra[LRH_SPUR_SPECTRA];
  lrh_spur *sp; float *table, *signal; int *ind;
  float *sig, *der, *pha, *tmp;
  float sp_d0, sp_d1, sp_d2;
  /* the search for new spurs (lro_spur_search_config): make_fft2's sums over 3 spur_speknum power rows (fft2.c:673-699) and the cleaned
     search spectrum spursearch_spectrum_cleanup leaves (spursub.c:40-175) */
  int ss_first, ss_last, ss_counter, ss_completed; float *ss_sum, *ss_spec, *ss_min; float ss_threshold;
  /* the ring the spurs live in (fftx of spur.c): fft2_float with the second fft on, fft1_float with it off (fft1_c, fft1.c:4196-4244) */
  float *fftx; int nx, maxn;
} lro_spurs;

static void spur_complex_lowpass(const float *zin, float *zout, int nn, int siz)
{
  int avgnum = nn | 1;
  if (avgnum > nn && avgnum > siz / 4) avgnum -= 2;
  if (avgnum < 1) return;
  float t1 = 0, t2 = 0, t3 = (float)(1.0 / avgnum);
  for (int i = 0; i < avgnum; i++) { t1 += zin[2 * i]; t2 += zin[2 * i + 1]; }
  int j = 1 + avgnum / 2;
  const float r1 = t1 * t3, r2 = t2 * t3;
  for (int i = 0; i < j; i++) { zout[2 * i] = r1; zout[2 * i + 1] = r2; }
  int i = 0, k = avgnum;
  while (k < siz) {
    t1 += zin[2 * k] - zin[2 * i]; t2 += zin[2 * k + 1] - zin[2 * i + 1];
    zout[2 * j] = t3 * t1; zout[2 * j + 1] = t3 * t2;
    i++; j++; k++;
  }
  t1 *= t3; t2 *= t3;
  while (j < siz) { zout[2 * j] = t1; zout[2 * j + 1] = t2; j++; }
}
static void spur_remove_phasejumps(float *z, int siz)
{
  float t1 = 0;
  for (int i = 1; i < siz; i++) {
    z[i] += t1;
    if (z[i] - z[i - 1] > PI_L) { z[i] -= (float)(2 * PI_L); t1 -= (float)(2 * PI_L); }
    if (z[i] - z[i - 1] < -PI_L) { z[i] += (float)(2 * PI_L); t1 += (float)(2 * PI_L); }
  }
}
static float spur_average_slope(const float *z, int siz)
{
  const int k = siz / 2;
  float t2 = 0, t3 = 0;
  for (int i = 0; i < k; i++) { t2 += z[i]; t3 += z[k + i]; }
  return (t3 - t2) / (k * k);
}

/* spur_phase_parameters, spur.c:1427-1652: phase, frequency and drift corrections sp_d0 / sp_d1 / sp_d2 from the de-rotated
   history sp_sig, amplitude and noise of the spur */
static void spur_phase_parameters(lro_spurs *S, lrh_spur *q)
{
  float *sig = S->sig, *der = S->der, *pha = S->pha, *tmp = S->tmp;
  const int n = S->speknum, ns = S->numsub, av = S->avgnum;
  float t1, t2, t3, r1, r2, a1, a2, b1, b2, d1, d2;
  for (int i = 1; i < n; i++) {
    t1 = sig[2 * i] * sig[2 * i - 2] + sig[2 * i + 1] * sig[2 * i - 1];
    t2 = sig[2 * i + 1] * sig[2 * i - 2] - sig[2 * i] * sig[2 * i - 1];
    r1 = (float)sqrt(t1 * t1 + t2 * t2);
    if (r1 > 0.000000001) { der[2 * i - 2] = t1 / r1; der[2 * i - 1] = t2 / r1; }
    else { der[2 * i - 2] = 0; der[2 * i - 1] = 0; }
  }
  spur_complex_lowpass(der, tmp, av, ns);
  r1 = 0; r2 = 0;
  for (int i = 1 + av / 2; i < ns - av / 2; i++) {
    t1 = tmp[2 * i] * tmp[2 * i - 2] + tmp[2 * i + 1] * tmp[2 * i - 1];
    t2 = tmp[2 * i + 1] * tmp[2 * i - 2] - tmp[2 * i] * tmp[2 * i - 1];
    t3 = (float)sqrt(t1 * t1 + t2 * t2);
    if (t3 > 0.00001) { r1 += t1 / t3; r2 += t2 / t3; }
  }
  S->sp_d2 = (float)atan2(r2, r1);
  t1 = q->spur_d2pha + S->sp_d2;
  if (fabs(t1) > S->max_d2 && fabs(S->sp_d2) > S->max_d2) S->sp_d2 = -q->spur_d2pha / n;
  else {
    t1 = S->weiold * q->spur_avgd2 + S->weinew * t1;
    if (q->spur_noise > 0.000001 && fabs(q->spur_ampl) > 0.000001) { t2 = (float)(0.1 * fabs(q->spur_ampl) / q->spur_noise); t2 = 1 / (1 + t2); }
    else t2 = 1;
    S->sp_d2 = t2 * (t1 - q->spur_d2pha) + (1 - t2) * S->sp_d2;
  }
  for (int i = 0; i < ns; i++) pha[i] = (float)atan2(tmp[2 * i + 1], tmp[2 * i]);
  spur_remove_phasejumps(pha, ns);
  pha[ns] = 0;
  for (int i = ns; i > 0; i--) pha[i - 1] = pha[i] - pha[i - 1];
  t1 = (float)(S->sp_d2 * 0.5);
  for (int i = 2; i < n; i++) pha[ns - i] -= i * (i - 1) * t1;
  int na = n - av;
  if (na < 10) na = n - av / 2;
  if (na < 3) na = n;
  const int ia = n - na;
  S->sp_d1 = spur_average_slope(&pha[ia], na);
  b1 = (float)cos(S->sp_d1); b2 = (float)sin(S->sp_d1);
  a1 = b1; a2 = b2;
  d1 = (float)cos(S->sp_d2); d2 = (float)sin(S->sp_d2);
  t1 = 0; t2 = 0;
  for (int i = ns; i >= 0; i--) {
    r1 = a1 * sig[2 * i] + a2 * sig[2 * i + 1];
    r2 = a1 * sig[2 * i + 1] - a2 * sig[2 * i];
    tmp[2 * i] = r1; tmp[2 * i + 1] = r2;
    t3 = (float)sqrt(r1 * r1 + r2 * r2);
    if (t3 > 0) { t1 += r1 / t3; t2 += r2 / t3; r2 += t3 * t3; }
    r1 = a1 * b1 + a2 * b2; a2 = a2 * b1 - a1 * b2; a1 = r1;
    r1 = b1 * d1 + b2 * d2; b2 = b2 * d1 - b1 * d2; b1 = r1;
  }
  S->sp_d0 = (float)atan2(t2, t1);
  t3 = (float)sqrt(t1 * t1 + t2 * t2);
  t1 /= t3; t2 /= t3;
  a1 = 0; a2 = 0;
  d1 = (float)(-0.5 * ns);
  for (int i = 0; i < n; i++) {
    r1 = t1 * tmp[2 * i] + t2 * tmp[2 * i + 1];
    r2 = t1 * tmp[2 * i + 1] - t2 * tmp[2 * i];
    tmp[2 * i] = r1; tmp[2 * i + 1] = r2;
    a1 += r1;
    if (r1 > 0 && fabs(r2) < fabs(r1)) a2 += (float)(d1 * r2 / fabs(r1));
    else a2 += (float)(d1 * atan2(r2, r1));
    d1 += 1;
  }
  a1 /= n;
  q->spur_ampl = a1;
  a2 /= S->linefit;
  S->sp_d1 += a2;
  d2 = (float)(-0.5 * ns * a2);
  b1 = (float)cos(a2); b2 = (float)-sin(a2);
  a1 = (float)cos(d2); a2 = (float)sin(d2);
  t1 = 0;
  for (int i = 0; i < n; i++) {
    r1 = a1 * tmp[2 * i] - a2 * tmp[2 * i + 1];
    r2 = a1 * tmp[2 * i + 1] + a2 * tmp[2 * i];
    tmp[2 * i] = r1; tmp[2 * i + 1] = r2;
    t1 += r1;
    r1 = a1 * b1 + a2 * b2; a2 = a2 * b1 - a1 * b2; a1 = r1;
  }
  t1 /= n;
  q->spur_ampl = t1;
  t2 = 0;
  for (int i = 0; i < n; i++) t2 += (tmp[2 * i] - t1) * (tmp[2 * i] - t1) + tmp[2 * i + 1] * tmp[2 * i + 1];
  q->spur_noise = (float)sqrt(t2 / n);
}

/* refine_pll_parameters, spur.c:634-680 */
static void spur_refine_pll(lro_spurs *S, lrh_spur *q, const float *zsig, int na, int mask)
{
  float phase = q->spur_d0pha, phase_slope = q->spur_d1pha, phase_curv = q->spur_d2pha, r1;
  int ni = na;
  phase_slope += phase_curv; phase += phase_slope;
  float a1 = (float)cos(phase), a2 = (float)sin(phase), b1 = (float)cos(phase_slope), b2 = (float)sin(phase_slope);
  float d1 = (float)cos(phase_curv), d2 = (float)sin(phase_curv);
  for (int i = S->speknum - 1; i >= 0; i--) {
    S->sig[2 * i] = a1 * zsig[2 * ni] + a2 * zsig[2 * ni + 1];
    S->sig[2 * i + 1] = a1 * zsig[2 * ni + 1] - a2 * zsig[2 * ni];
    r1 = a1 * b1 + a2 * b2; a2 = a2 * b1 - a1 * b2; a1 = r1;
    r1 = b1 * d1 + b2 * d2; b2 = b2 * d1 - b1 * d2; b1 = r1;
    ni = (ni + mask) & mask;
  }
  spur_phase_parameters(S, q);
  phase += S->sp_d0; phase_slope += S->sp_d1; phase_curv += S->sp_d2;
  phase -= phase_slope; phase_slope -= phase_curv;
  q->spur_d0pha = phase; q->spur_d1pha = phase_slope; q->spur_d2pha = phase_curv;
}

/* shift_spur_table, spursub.c:1070-1246 (one channel, float): the spur has drifted out of the centre of its SPUR_WIDTH bins */
static void spur_shift_table(lro_spurs *S, lrh_spur *q, float *tab, int j, int na, int mask, int n2)
{
  const int nj = (na + 1) & mask, shift = j < 0 ? -1 : 1;
  q->spur_location += shift;
  if (q->spur_location < SPW) { q->spur_flag = 1; q->spur_location = 2 * SPW; return; }
  if (q->spur_location > n2 - SPW) { q->spur_flag = 1; q->spur_location = n2 - 2 * SPW; return; }
  int ni = (na - S->speknum + mask + 1) & mask;
  while (ni != nj) {
    float *t = tab + ni * SPW * 2;
    if (shift == 1) { for (int i = 1; i < SPW; i++) { t[2 * i - 2] = t[2 * i]; t[2 * i - 1] = t[2 * i + 1]; } t[2 * SPW - 2] = 0; t[2 * SPW - 1] = 0; }
    else { for (int i = SPW - 1; i > 0; i--) { t[2 * i] = t[2 * i - 2]; t[2 * i + 1] = t[2 * i - 1]; } t[0] = 0; t[1] = 0; }
    ni = (ni + 1) & mask;
  }
}

static int spur_index(float freq, int loc, int *jout)        /* the reference line shape for this frequency; -1: out of the window */
{
  int j = (int)(freq) + 2 - loc - SPSZ / 2;
  *jout = j;
  if (j < 0 || j > 1) return -1;
  j = 1 - j;
  int ind = (int)(NSPEC * (freq - (int)(freq)));
  if (ind == NSPEC) ind = NSPEC - 1;
  *jout = j;
  return ind * SPSZ + j;
}

/* eliminate_spurs for the transform at ring slot na of fft2_float (spur.c:36-494) */
static void spur_eliminate(lro_spurs *S, float *fftx, int na, int n2, int maxn)
{
  const int mask = maxn - 1;
  const float ff = S->freq_factor;
  for (int s = 0; s < S->n; s++) {
    lrh_spur *q = &S->sp[s];
    float *tab = S->table + (size_t)s * maxn * SPW * 2, *spt = tab + na * SPW * 2;
    int *uind = S->ind + (size_t)s * maxn;
    float *zsig = S->signal + (size_t)s * maxn * 2;
    float *z;
    int i, j, k, ind;
    if (q->spur_flag == 1) {
      j = (int)(q->spur_freq) + 2 - q->spur_location - SPSZ / 2;
      if (j < 0 || j > 1) { if (j < -1) j = -1; if (j > 2) j = 2; spur_shift_table(S, q, tab, j, na, mask, n2); }
    }
    if (q->spur_flag != 0) {                 /* unlocked: keep the history, count; re-lock is the control plane's (spur.c:130-153) */
      z = &fftx[2 * ((size_t)na * n2 + q->spur_location)];
      for (i = 0; i < SPW; i++) { spt[2 * i] = z[2 * i]; spt[2 * i + 1] = z[2 * i + 1]; }
      q->spur_flag++;
      if (q->spur_flag > 1000000) q->spur_flag -= 2 * 3 * 5 * 7 * S->speknum;
      continue;
    }
    float phase_slope = q->spur_d1pha + q->spur_d2pha, phase_curv, phase, ampl;
    float rot = (float)(-0.5 * phase_slope / PI_L), freq, r1, r2, t1, t2;
    i = (int)(q->spur_freq * ff - rot + 0.5);
    rot += i;
    freq = rot / ff;
    q->spur_freq = freq;
    int lost = 0;
    for (;;) {
      ind = spur_index(freq, q->spur_location, &j);
      if (ind >= 0) break;
      if (j < -1 || j > 2) { q->spur_flag = 1; lost = 1; break; }
      spur_shift_table(S, q, tab, j, na, mask, n2);
      if (q->spur_flag) { lost = 1; break; }                      /* shifted against the band edge */
    }
    if (lost) continue;
    uind[na] = ind;
    z = &fftx[2 * ((size_t)na * n2 + q->spur_location)];
    r1 = 0; r2 = 0;
    for (i = 0; i < SPW; i++) {
      spt[2 * i] = z[2 * i]; r1 += z[2 * i] * S->spectra[ind + i];
      spt[2 * i + 1] = z[2 * i + 1]; r2 += z[2 * i + 1] * S->spectra[ind + i];
    }
    if ((j ^ (q->spur_location & 1)) == 1) { r1 = -r1; r2 = -r2; }
    zsig[2 * na] = r1; zsig[2 * na + 1] = r2;
    int iter = 0, diffind;
    freq = q->spur_freq;
    for (;;) {
      iter++;
      spur_refine_pll(S, q, zsig, na, mask);
      phase_slope = q->spur_d1pha; phase_curv = q->spur_d2pha;
      phase_slope += phase_curv;
      const int nx = (na - S->speknum + mask) & mask;
      diffind = 0;
      int ni = na, out = 0;
      while (ni != nx) {
        rot = (float)(-0.5 * phase_slope / PI_L);
        i = (int)(freq * ff - rot + 0.5);
        rot += i;
        freq = rot / ff;
        ind = spur_index(freq, q->spur_location, &j);
        if (ind < 0) { out = 1; break; }
        k = (uind[ni] - ind + NSPEC * SPSZ) & (NSPEC * SPSZ - 1);
        if (k > NSPEC * SPSZ / 2) k = NSPEC * SPSZ - k;
        if (k > diffind) diffind = k;
        uind[ni] = ind;
        if (k != 0) {
          const float *tt = tab + ni * SPW * 2;
          r1 = 0; r2 = 0;
          for (i = 0; i < SPW; i++) { r1 += tt[2 * i] * S->spectra[ind + i]; r2 += tt[2 * i + 1] * S->spectra[ind + i]; }
          if ((j ^ (q->spur_location & 1)) == 1) { r1 = -r1; r2 = -r2; }
          zsig[2 * ni] = r1; zsig[2 * ni + 1] = r2;
        }
        phase_slope -= phase_curv;
        ni = (ni + mask) & mask;
      }
      if (out) break;
      if (!(diffind > 2.5 * SPSZ && iter < 5)) break;
    }
    if (diffind != 0) spur_refine_pll(S, q, zsig, na, mask);
    if (fabs(q->spur_ampl) < S->minston * q->spur_noise) { q->spur_flag = 1; continue; }
    /* subtract the spur from the new transform */
    phase = q->spur_d0pha; phase_slope = q->spur_d1pha; phase_curv = q->spur_d2pha; ampl = q->spur_ampl;
    phase_slope += phase_curv; phase += phase_slope;
    q->spur_d0pha = phase; q->spur_d1pha = phase_slope;
    if (q->spur_d0pha > PI_L) q->spur_d0pha -= (float)(2 * PI_L);
    if (q->spur_d0pha < -PI_L) q->spur_d0pha += (float)(2 * PI_L);
    if (q->spur_d1pha > PI_L) q->spur_d1pha -= (float)(2 * PI_L);
    if (q->spur_d1pha < -PI_L) q->spur_d1pha += (float)(2 * PI_L);
    if (q->spur_d2pha > PI_L) q->spur_d2pha -= (float)(2 * PI_L);
    if (q->spur_d2pha < -PI_L) q->spur_d2pha += (float)(2 * PI_L);
    q->spur_avgd2 = S->weiold * q->spur_avgd2 + S->weinew * phase_curv;
    rot = (float)(-0.5 * phase_slope / PI_L);
    i = (int)(q->spur_freq * ff - rot + 0.5);
    rot += i;
    freq = rot / ff;
    q->spur_freq = freq;
    lost = 0;
    for (;;) {
      ind = spur_index(freq, q->spur_location, &j);
      if (ind >= 0) break;
      if (j < -1 || j > 2) { q->spur_flag = 1; lost = 1; break; }
      spur_shift_table(S, q, tab, j, na, mask, n2);
      if (q->spur_flag) { lost = 1; break; }
    }
    if (lost) continue;
    if ((j ^ (q->spur_location & 1)) == 1) { t1 = (float)(-cos(phase) * ampl); t2 = (float)(-sin(phase) * ampl); }
    else { t1 = (float)(cos(phase) * ampl); t2 = (float)(sin(phase) * ampl); }
    z = &fftx[2 * ((size_t)na * n2 + q->spur_location)];
    for (i = 0; i < SPW; i++) { z[2 * i] -= S->spectra[ind + i] * t1; z[2 * i + 1] -= S->spectra[ind + i] * t2; }
  }
}

int lro_spur_config(lro_ctx *c, int max_spurs, int speknum, const f32 *spectra)
{
  LRO_F32_ONLY(c);
  if (!c || max_spurs < 0 || (max_spurs && (!spectra || speknum < 4 || 4 * speknum > (c->cfg.second_fft_enable ? c->cfg.max_fft2n : c->cfg.max_fft1n)))) return LRH_EINVAL;
  lro_spurs *S = c->spurs;
  if (S) { free(S->sp); free(S->table); free(S->signal); free(S->ind); free(S->sig); free(S->der); free(S->pha); free(S->tmp); free(S->ss_sum); if (S->ss_spec) free(S->ss_spec - 8); free(S->ss_min); free(S); c->spurs = NULL; }
  if (!max_spurs) return LRH_OK;
  const int second = c->cfg.second_fft_enable != 0;
  const int maxn = second ? c->cfg.max_fft2n : c->cfg.max_fft1n;
  S = calloc(1, sizeof *S);
  S->fftx = second ? c->fft2_float : c->fft1_float; S->nx = second ? c->N2 : c->N1; S->maxn = maxn;
  S->max = max_spurs; S->n = 0; S->speknum = speknum; S->numsub = speknum - 1; S->avgnum = speknum / 3; if (S->avgnum > 10) S->avgnum = 10;
  S->freq_factor = second ? (float)c->M2 / c->N2 : (float)c->M1 / c->N1;   /* buf.c:480 / 1118: new points / size of the transform the spurs are taken from */
  S->max_d2 = (float)(PI_L * S->freq_factor / speknum);
  S->minston = (float)(1 / sqrt(0.5 * (float)(speknum)));
  { float t1 = (float)(0.5 * speknum); S->weiold = t1 / (1 + t1); S->weinew = 1 / (1 + t1);
    t1 = (float)(-0.5 * S->numsub); S->linefit = 0; for (int i = 0; i < speknum; i++) { S->linefit += t1 * t1; t1 += 1; } }
  memcpy(S->spectra, spectra, sizeof S->spectra);
  S->sp = calloc(max_spurs, sizeof(lrh_spur)); S->table = calloc((size_t)max_spurs * maxn * SPW * 2, 4); S->signal = calloc((size_t)max_spurs * maxn * 2, 4);
  S->ind = calloc((size_t)max_spurs * maxn, 4);
  S->sig = calloc(2 * (maxn + 8), 4); S->der = calloc(2 * (maxn + 8), 4); S->pha = calloc(2 * (maxn + 8), 4); S->tmp = calloc(2 * (maxn + 8), 4);
  c->spurs = S;
  return LRH_OK;
}
/* ---- the search for new spurs on the resident power rows ---- */
int lro_spur_search_config(lro_ctx *c, int first_point, int last_point)
{
  LRO_F32_ONLY(c);
  lro_spurs *S = c ? c->spurs : NULL;
  if (!S) return LRH_ESTATE;
  free(S->ss_sum); if (S->ss_spec) free(S->ss_spec - 8); free(S->ss_min); S->ss_sum = S->ss_spec = S->ss_min = NULL;
  S->ss_counter = 0; S->ss_completed = 0; S->ss_threshold = 0;
  if (first_point == 0 && last_point == 0) return LRH_OK;
  if (first_point < 0 || last_point >= S->nx || last_point - first_point < 64) return LRH_EINVAL;
  S->ss_first = first_point; S->ss_last = last_point;
  /* (the reference's walk reads up to three bins before the first and 31 behind the last point of the range, spursub.c:48, 160-167) */
  S->ss_sum = calloc(S->nx + 8, 4); S->ss_spec = (float *)calloc(S->nx + 72, 4) + 8; S->ss_min = calloc(S->nx / 32 + 8, 4);
  return LRH_OK;
}
/* parabolic_fit, llsq.c:113-153 */
static void parabolic_fit3(float *amp, float *pos, float y1, float y2, float y3)
{
  float t4 = y1 - y3, t3 = 2 * (y1 + y3 - 2 * y2);
  if (t3 < 0) { *amp = y2 - 0.5F * t4 * t4 / t3; t4 = t4 / t3; if (fabs(t4) > 1) t4 /= (float)fabs(t4); *pos = t4; }
  else if (y1 > y3) { *amp = y1; *pos = -1; }
  else { *amp = y3; *pos = 1; }
}
/* spursearch_spectrum_cleanup, spursub.c:40-175: noise floor from the minima of groups of 32 bins, floor subtracted, and every peak above
   the threshold that does not look like the reference line shape of a spur (it might be a wanted signal) wiped out of the search spectrum */
static void spur_search_cleanup(lro_ctx *c, lro_spurs *S)
{
  float *sp = S->ss_spec, *mn = S->ss_min;
  const int first = S->ss_first, last = S->ss_last;
  int k = 0, i, j, ia, ib, nn;
  float t1, noise, amp, pos, maxpow;
  for (i = first; i < last; i += 32) { mn[k] = 1e30f; for (j = 0; j < 32; j++) if (sp[i + j] < mn[k]) mn[k] = sp[i + j]; k++; }
  if (!k) return;
  t1 = 0; for (i = 0; i < k; i++) t1 += mn[i];
  t1 /= k;
  noise = 0; j = 0;
  for (i = 0; i < k; i++) if (mn[i] < t1) { noise += mn[i]; j++; }
  noise /= j;
  noise *= pow(10., 0.7 / sqrt((float)(3 * S->speknum)));
  for (i = first; i < last; i++) { sp[i] -= noise; if (sp[i] < 0) sp[i] = 0; }
  S->ss_threshold = noise * pow(10., 1.5 / sqrt((float)(3 * S->speknum)));
  const float thr = S->ss_threshold;
  ia = first;
  for (;;) {
    while (ia < last && sp[ia] < thr) ia++;
    if (ia == last) break;
    ib = ia + 1;
    while (ib < last && sp[ib] > thr) ib++;
    if (ib == last && ib - ia < SPSZ) break;
    maxpow = 0; k = ia;
    for (i = ia; i < ib; i++) if (sp[i] > maxpow) { maxpow = sp[i]; k = i; }
    parabolic_fit3(&amp, &pos, sp[k - 1], sp[k], sp[k + 1]);
    nn = k - SPSZ / 2 + 1;
    if (pos < 0) { pos += 1; nn--; }
    i = pos * NSPEC; if (i >= NSPEC) i = NSPEC - 1;
    const float *spk = &S->spectra[i * SPSZ];
    float refamp = fabs(spk[SPSZ / 2]);
    if (refamp < fabs(spk[SPSZ / 2 - 1])) refamp = fabs(spk[SPSZ / 2 - 1]);
    const float t2 = (float)sqrt(maxpow) / refamp;
    float tot = 0, rem = 0, edge = 0;
    int bad = 0;
    for (i = 0; i < SPSZ; i++) {
      if (sp[nn + i] < 0) { bad = 1; break; }
      tot += sp[nn + i];
      const float r1 = pow(sqrt(sp[nn + i]) - t2 * fabs(spk[i]), 2.0);
      rem += r1;
      if (edge < r1 && (i < 2 || i >= SPSZ - 2)) edge = r1;
    }
    if (bad || (edge / noise > 5 && tot / edge < 1000) || (edge / noise > 2 && tot / edge < 300) || rem / tot > 0.1) {
      while ((sp[ia] > sp[ia - 1] || sp[ia] > sp[ia - 2] || sp[ia] > sp[ia - 3]) && ia > first) ia--;
      while ((sp[ib] > sp[ib + 1] || sp[ib] > sp[ib + 2] || sp[ib] > sp[ib + 3]) && ib < last) ib++;
      for (i = ia; i < ib; i++) sp[i] = -0.00000001;
    }
    ia = ib;
  }
}
/* make_fft2's bookkeeping of the search spectrum for the new power row (fft2.c:673-699) */
static void spur_search_row(lro_ctx *c, const float *pwra)
{
  lro_spurs *S = c->spurs;
  if (!S || !S->ss_sum) return;
  const int a = S->ss_first, b = S->ss_last;
  if (S->ss_counter > 3 * S->speknum) {
    S->ss_counter = 0;
    for (int i = a; i <= b; i++) S->ss_spec[i] = S->ss_sum[i] + pwra[i];
    spur_search_cleanup(c, S);
    S->ss_completed++;
  } else {
    if (S->ss_counter == 0) for (int i = a; i <= b; i++) S->ss_sum[i] = pwra[i];
    else for (int i = a; i <= b; i++) S->ss_sum[i] += pwra[i];
    S->ss_counter++;
  }
}
int lro_spur_search_get(lro_ctx *c, f32 *spectrum, f32 *threshold, int *completed, int *sum_counter)
{
  lro_spurs *S = c ? c->spurs : NULL;
  if (!S || !S->ss_sum) return LRH_ESTATE;
  if (spectrum) out_f32(spectrum, S->ss_spec + S->ss_first, (size_t)(S->ss_last - S->ss_first + 1));
  if (threshold) *threshold = S->ss_threshold;
  if (completed) *completed = S->ss_completed;
  if (sum_counter) *sum_counter = S->ss_counter;
  return LRH_OK;
}

int lro_spur_set(lro_ctx *c, int n, const lrh_spur *sp, const f32 *table, const f32 *signal, const int *ind)
{
  LRO_F32_ONLY(c);
  lro_spurs *S = c ? c->spurs : NULL;
  if (!S || n < 0 || n > S->max || (n && (!sp || !table || !signal || !ind))) return LRH_EINVAL;
  const int maxn = S->maxn;
  S->n = n;
  if (n) { memcpy(S->sp, sp, n * sizeof *sp); memcpy(S->table, table, (size_t)n * maxn * SPW * 2 * 4); memcpy(S->signal, signal, (size_t)n * maxn * 2 * 4); memcpy(S->ind, ind, (size_t)n * maxn * 4); }
  return LRH_OK;
}
/* remove_spur / swap_spurs (spur.c:596-631, spursub.c:755): spur i becomes what was spur src[i] */
int lro_spur_permute(lro_ctx *c, int n, const int *src)
{
  lro_spurs *S = c ? c->spurs : NULL;
  if (!S || n < 0 || n > S->n || (n && !src)) return LRH_EINVAL;
  const size_t maxn = S->maxn;
  for (int i = 0; i < n; i++) if (src[i] < 0 || src[i] >= S->n) return LRH_EINVAL;
  if (n) {
    lrh_spur *sp = malloc(n * sizeof *sp); float *tab = malloc((size_t)n * maxn * SPW * 2 * 4), *sig = malloc((size_t)n * maxn * 2 * 4); int *ind = malloc((size_t)n * maxn * 4);
    for (int i = 0; i < n; i++) {
      sp[i] = S->sp[src[i]];
      memcpy(tab + (size_t)i * maxn * SPW * 2, S->table + (size_t)src[i] * maxn * SPW * 2, maxn * SPW * 2 * 4);
      memcpy(sig + (size_t)i * maxn * 2, S->signal + (size_t)src[i] * maxn * 2, maxn * 2 * 4);
      memcpy(ind + (size_t)i * maxn, S->ind + (size_t)src[i] * maxn, maxn * 4);
    }
    memcpy(S->sp, sp, n * sizeof *sp); memcpy(S->table, tab, (size_t)n * maxn * SPW * 2 * 4); memcpy(S->signal, sig, (size_t)n * maxn * 2 * 4); memcpy(S->ind, ind, (size_t)n * maxn * 4);
    free(sp); free(tab); free(sig); free(ind);
  }
  S->n = n;
  return LRH_OK;
}
int lro_spur_get(lro_ctx *c, int max, lrh_spur *sp, int *n)
{
  lro_spurs *S = c ? c->spurs : NULL;
  if (!S || !sp || !n) return LRH_EINVAL;
  *n = S->n < max ? S->n : max;
  memcpy(sp, S->sp, *n * sizeof *sp);
  return LRH_OK;
}

/* ---- acquisition of a new spur on the resident fft2 spectra: store_new_spur (spursub.c:619-751, one channel, float) with
   spurspek_norm (:472-493) and make_spur_freq (:592-616, parabolic_fit llsq.c:113-153), then spur_phase_lock (:1247-1426) with
   verify_spur_pll (:1428-1843).  ffts_na = p->fft2_na: the ring position behind the newest transform (wcw.c:288-289).
   *locked = 1: the spur joined the tracked ones (no_of_spurs++, spursub.c:309); 0: no lock (the reference returns without it). */
static void spur_parabolic_fit(float *amp, float *pos, float y1, float y2, float y3)
{
  float t4 = y1 - y3, t3 = 2 * (y1 + y3 - 2 * y2);
  if (t3 < 0) { *amp = y2 - 0.5F * t4 * t4 / t3; t4 = t4 / t3; if (fabs(t4) > 1) t4 /= (float)fabs(t4); *pos = t4; }
  else if (y1 > y3) { *amp = y1; *pos = -1; }
  else { *amp = y3; *pos = 1; }
}
int lro_spur_acquire(lro_ctx *c, const lrh_ptrs *p, int pnt, int *locked)
{
  LRO_F32_ONLY(c);
  lro_spurs *S = c ? c->spurs : NULL;
  if (!S || !p || !locked) return LRH_EINVAL;
  *locked = 0;
  const int maxn = S->maxn, mask = maxn - 1, n2 = S->nx, n = S->speknum, na = p->fft2_na & mask;      /* (p->fft2_na carries ffts_na: fft1_nb with the second fft off) */
  if (S->n >= S->max || pnt < 1 || pnt + SPW + 1 > n2) return LRH_EINVAL;
  const int s = S->n;
  lrh_spur *q = &S->sp[s];
  float *tab = S->table + (size_t)s * maxn * SPW * 2, *zsig = S->signal + (size_t)s * maxn * 2;
  int *uind = S->ind + (size_t)s * maxn;
  const float *fftx = S->fftx;
  float power[SPW];
  memset(q, 0, sizeof *q);
  q->spur_ampl = 1; q->spur_noise = 0.001f; q->spur_avgd2 = 0;               /* spursub.c:290-292 */
  /* store_new_spur: the last speknum transforms' bins into the table, their summed power */
  for (int i = 0; i < SPW; i++) power[i] = 0;
  for (int np = (na - n + maxn) & mask; np != na; np = (np + 1) & mask) {
    const float *z = &fftx[2 * ((size_t)np * n2 + pnt)];
    float *t = tab + (size_t)np * SPW * 2;
    for (int i = 0; i < SPW; i++) { t[2 * i] = z[2 * i]; t[2 * i + 1] = z[2 * i + 1]; power[i] += z[2 * i] * z[2 * i] + z[2 * i + 1] * z[2 * i + 1]; }
  }
  { float t1 = 0.5f * (power[0] + power[SPW - 1]), t2 = 0;                   /* spurspek_norm */
    for (int i = 0; i < SPW; i++) { power[i] -= t1; if (power[i] < 0) power[i] = 0; t2 += power[i]; }
    for (int i = 0; i < SPW; i++) power[i] /= t2; }
  q->spur_location = pnt;
  { float t2 = 0; int k = 0;                                                 /* make_spur_freq */
    for (int i = 0; i < SPW; i++) if (t2 < power[i]) { t2 = power[i]; k = i; }
    if (k == 0 || k == SPW - 1) return LRH_OK;
    float amp, pos;
    spur_parabolic_fit(&amp, &pos, (float)sqrt(power[k - 1]), (float)sqrt(t2), (float)sqrt(power[k + 1]));
    q->spur_freq = pnt + k + pos; }
  /* spur_phase_lock: history weighted with the power spectrum, alternating signs of neighbouring bins */
  { int izz = 0;
    for (int np = (na - n + maxn) & mask; np != na; np = (np + 1) & mask) {
      const float *t = tab + (size_t)np * SPW * 2;
      float t1 = 0, t2 = 0;
      for (int i = 0; i < SPW - 1; i += 2) { t1 += power[i] * t[2 * i] - power[i + 1] * t[2 * i + 2]; t2 += power[i] * t[2 * i + 1] - power[i + 1] * t[2 * i + 3]; }
      if ((q->spur_location & 1) == 1) { t1 = -t1; t2 = -t2; }
      zsig[2 * np] = t1; zsig[2 * np + 1] = t2;
      S->sig[2 * izz] = t1; S->sig[2 * izz + 1] = t2; izz++;
    } }
  spur_phase_parameters(S, q);
  if (q->spur_ampl < 3 * q->spur_noise / sqrt((float)(n))) return LRH_OK;
  /* verify_spur_pll */
  float average_rot = q->spur_freq * S->freq_factor, d0err = 0, d1err = 0, d2err = 0;
  for (int iter = 1; iter <= 5; iter++) {
    float a1 = (float)cos(q->spur_d0pha), a2 = (float)sin(q->spur_d0pha), b1 = (float)cos(q->spur_d1pha), b2 = (float)sin(q->spur_d1pha);
    float d1 = (float)cos(q->spur_d2pha), d2 = (float)sin(q->spur_d2pha), slope = q->spur_d1pha, r1, r2;
    const float curv = q->spur_d2pha;
    int ni = (na + mask) & mask;
    for (int izz = n - 1; izz >= 0; izz--) {
      float rot = (float)(-0.5 * slope / PI_L);
      int i = (int)(average_rot - rot + 0.5);
      rot += i;
      const float freq = rot / S->freq_factor;
      int j = (int)(freq) + 2 - q->spur_location - SPSZ / 2;
      if (j < 0) j = 0;
      j = 1 - j;
      if (j < 0) j = 0;
      int ind = (int)(NSPEC * (freq - (int)(freq)));
      if (ind == NSPEC) ind = NSPEC - 1;
      ind = ind * SPSZ + j;
      uind[ni] = ind;
      const float *t = tab + (size_t)ni * SPW * 2;
      r1 = 0; r2 = 0;
      for (i = 0; i < SPW; i++) { r1 += S->spectra[ind + i] * t[2 * i]; r2 += S->spectra[ind + i] * t[2 * i + 1]; }
      if ((j ^ (q->spur_location & 1)) == 1) { r1 = -r1; r2 = -r2; }
      zsig[2 * ni] = r1; zsig[2 * ni + 1] = r2;
      S->sig[2 * izz] = r1 * a1 + r2 * a2; S->sig[2 * izz + 1] = r2 * a1 - r1 * a2;
      ni = (ni + mask) & mask;
      r1 = a1 * b1 + a2 * b2; a2 = a2 * b1 - a1 * b2; a1 = r1;
      r2 = b1 * d1 + b2 * d2; b2 = b2 * d1 - b1 * d2; b1 = r2;
      slope -= curv;
    }
    spur_phase_parameters(S, q);
    if (q->spur_ampl < 3 * q->spur_noise / sqrt((float)(n))) return LRH_OK;
    q->spur_d0pha += S->sp_d0; q->spur_d1pha += S->sp_d1; q->spur_d2pha += S->sp_d2;
    while (q->spur_d0pha > PI_L) q->spur_d0pha -= (float)(2 * PI_L);
    while (q->spur_d0pha < -PI_L) q->spur_d0pha += (float)(2 * PI_L);
    while (q->spur_d1pha > PI_L) q->spur_d1pha -= (float)(2 * PI_L);
    while (q->spur_d1pha < -PI_L) q->spur_d1pha += (float)(2 * PI_L);
    while (q->spur_d2pha > PI_L) q->spur_d2pha -= (float)(2 * PI_L);
    while (q->spur_d2pha < -PI_L) q->spur_d2pha += (float)(2 * PI_L);
    { float rot = (float)(-0.5 * q->spur_d1pha / PI_L); const int i = (int)(average_rot - rot + 0.5); rot += i; q->spur_freq = rot / S->freq_factor; }
    if (iter > 1 && fabs(S->sp_d0) < 0.1 && fabs(S->sp_d1) < 0.01 && fabs(S->sp_d2) < 0.001 && fabs(d0err) < 0.3 && fabs(d1err) < 0.03 && fabs(d2err) < 0.003) {
      /* what would be left after the subtraction, over the window and the bin on either side: its spectrum must be flat */
      float errspek[SPW + 2];
      for (int i = 0; i < SPW + 2; i++) errspek[i] = 0;
      average_rot = q->spur_freq * S->freq_factor;
      a1 = q->spur_ampl * (float)cos(q->spur_d0pha); a2 = q->spur_ampl * (float)sin(q->spur_d0pha);
      b1 = (float)cos(q->spur_d1pha); b2 = (float)sin(q->spur_d1pha); d1 = (float)cos(q->spur_d2pha); d2 = (float)sin(q->spur_d2pha);
      slope = q->spur_d1pha;
      ni = (na + mask) & mask;
      for (int izz = n - 1; izz >= 0; izz--) {
        float rot = (float)(-0.5 * slope / PI_L);
        int i = (int)(average_rot - rot + 0.5);
        rot += i;
        const float freq = rot / S->freq_factor;
        int j = (int)(freq) + 2 - q->spur_location - SPSZ / 2;
        if (j < 0) j = 0;
        j = 1 - j;
        if (j < 0) j = 0;
        const int ind = uind[ni];
        float t1 = a1, t2 = a2;
        if ((j ^ (q->spur_location & 1)) == 1) { t1 = -a1; t2 = -a2; }
        const float *z = &fftx[2 * ((size_t)ni * n2 + q->spur_location)];
        for (i = 0; i < SPW; i++)
          errspek[i + 1] += (float)(pow(z[2 * i] - S->spectra[ind + i] * t1, 2.0) + pow(z[2 * i + 1] - S->spectra[ind + i] * t2, 2.0));
        errspek[0] += (float)(pow(z[-2], 2.0) + pow(z[-1], 2.0));
        errspek[SPW + 1] += (float)(pow(z[2 * (SPW + 1)], 2.0) + pow(z[2 * (SPW + 1) + 1], 2.0));
        ni = (ni + mask) & mask;
        r1 = a1 * b1 + a2 * b2; a2 = a2 * b1 - a1 * b2; a1 = r1;
        r2 = b1 * d1 + b2 * d2; b2 = b2 * d1 - b1 * d2; b1 = r2;
        slope -= q->spur_d2pha;
      }
      r1 = 0;
      for (int i = 0; i < SPW + 2; i++) { errspek[i] /= n; r1 += errspek[i]; errspek[i] = (float)sqrt(errspek[i]); }
      r1 = (float)sqrt(r1 / (SPW + 2));
      r2 = 0;
      for (int i = 0; i < SPW + 2; i++) r2 += (float)pow(errspek[i] - r1, 2.0);
      r2 = (float)sqrt(r2 / (SPW + 2));
      if (r2 > 0.5 * r1 / sqrt((int)(n)) + 0.02 * q->spur_ampl) return LRH_OK;
      q->spur_avgd2 = q->spur_d2pha;                                         /* spursub.c:1424 */
      S->n++; *locked = 1;
      /* initial_remove_spur, spursub.c:346-470 (init_spur_elimination calls it right behind the lock, :309): the carrier also leaves the
         spur_speknum transforms the loop was closed on -- the same backward walk once more, now writing */
      a1 = q->spur_ampl * (float)cos(q->spur_d0pha); a2 = q->spur_ampl * (float)sin(q->spur_d0pha);
      b1 = (float)cos(q->spur_d1pha); b2 = (float)sin(q->spur_d1pha); d1 = (float)cos(q->spur_d2pha); d2 = (float)sin(q->spur_d2pha);
      slope = q->spur_d1pha;
      ni = (na + mask) & mask;
      for (int izz = n - 1; izz >= 0; izz--) {
        float rot = (float)(-0.5 * slope / PI_L);
        int i = (int)(average_rot - rot + 0.5);
        rot += i;
        const float freq = rot / S->freq_factor;
        int j = (int)(freq) + 2 - q->spur_location - SPSZ / 2;
        if (j < 0) j = 0;
        j = 1 - j;
        if (j < 0) j = 0;
        const int ind = uind[ni];
        float t1 = a1, t2 = a2;
        if ((j ^ (q->spur_location & 1)) == 1) { t1 = -a1; t2 = -a2; }
        float *z = (float *)&fftx[2 * ((size_t)ni * n2 + q->spur_location)];
        for (i = 0; i < SPW; i++) { z[2 * i] -= S->spectra[ind + i] * t1; z[2 * i + 1] -= S->spectra[ind + i] * t2; }
        ni = (ni + mask) & mask;
        r1 = a1 * b1 + a2 * b2; a2 = a2 * b1 - a1 * b2; a1 = r1;
        r2 = b1 * d1 + b2 * d2; b2 = b2 * d1 - b1 * d2; b1 = r2;
        slope -= q->spur_d2pha;
      }
      return LRH_OK;
    }
    d0err = S->sp_d0; d1err = S->sp_d1; d2err = S->sp_d2;
  }
  return LRH_OK;
}


static void lro_spur_hook(lro_ctx *c, int na)
{
  lro_spurs *S = c->spurs;
  if (S->n > 0) spur_eliminate(S, S->fftx, na, S->nx, S->maxn);
}
