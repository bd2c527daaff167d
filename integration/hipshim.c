/*
 * hipshim.c -- Linrad-side glue for liblinrad_hip.so.  Our code, compiled inside the Linrad tree against Linrad's own headers.
 *
 * Linrad's stage functions are void f(void) over global rings and pointer variables (z_BUFFERS.txt).  With fft1 version 21 the
 * rings from fft1_float to timf3 live on the MI355X; each stand-in below
 *   hip_sync_in:   copies the pointer globals its stage reads into a local lrh_ptrs,
 *   calls the library stage (which advances exactly the fields the reference function advances),
 *   hip_sync_out:  copies those fields back into the globals,
 * and brings the small host-visible products back for the rest of Linrad: the newest fft1_sumsq block and fft1_slowsum
 * (wide graph, selective limiter), waterfall lines, blanker scalars, and timf3 -- from where fft3 / mix2 / the demodulators
 * run on the CPU as before.  Every stage thread copies only the fields its own stage owns, so the threads of wcw.c do not
 * disturb each other (the library itself may be called from any thread, include/linrad_hip.h "threading").
 */
#include <string.h>
#include <stdlib.h>
#include <pthread.h>
#include "globdef.h"
#include "uidef.h"
#include "fft1def.h"
#include "fft2def.h"
#include "fft3def.h"
#include "screendef.h"
#include "sigdef.h"
#include "seldef.h"
#include "blnkdef.h"
#include "thrdef.h"
#include "linrad_hip.h"
#include "hipshim.h"

/* One context per RF channel (ui.rx_rf_channels = 1 or 2): the library shards channels one per context, Linrad keeps both in every
   array -- frames {I0,Q0,I1,Q1} in timf1, {ch0, ch1} per bin / sample behind it, pointers counted in floats of the two-channel arrays.
   hip_rx is channel 0; HC contexts in hip_ctx[].  A pointer that counts floats is divided by HC on the way in and multiplied on the way
   out (HIP_IN / HIP_OUT); transform numbers, the blanker's sample pointers (timf2p_fit) and the power-sum pointers are per channel anyway.
   The sums Linrad forms over the channels -- fft1_sumsq, fft1_slowsum, the blanker's power and noise, the fft2 cross products -- are
   the exchanges of include/linrad_hip.h ("two coupled RF channels"), made here through host memory (lrh_exchange_read / _write). */
static lrh_ctx *hip_rx, *hip_ctx[2];
static int HC = 1;
#define HIP_IN(x) ((x) / HC)
#define HIP_OUT(x) ((x) * HC)
/* host scratch of the exchanges and of the channel interleaving: per thread -- Linrad calls the hooks from its stage threads and, for the
   network outputs, from the network thread (the buffers of threads that have ended are not reclaimed: a few hundred KB each) */
static __thread float *hip_xa, *hip_xb; static __thread size_t hip_xcap;
/* two REAL channels: Linrad's frames are {a_k, b_k}; the library takes one channel's real samples as adjacent pairs (one complex point per
   pair, fft1_reherm_dit_*'s packing), so the producer hook de-interleaves into one arena per channel -- half of timf1, same ring positions / 2 */
static char *hip_deint[2]; static int hip_real2;
static float hip_ch2_c1 = 1, hip_ch2_c2 = 0;                /* pg_ch2_c1 / pg_ch2_c2 as last handed to channel 1's context (pol_graph.c:160-170) */
static float *hip_scratch(size_t n) { if (n > hip_xcap) { free(hip_xa); free(hip_xb); hip_xa = malloc(n * sizeof(float)); hip_xb = malloc(n * sizeof(float)); hip_xcap = n; } return hip_xa; }
static float *hip_liminfo_sent;           /* the routing table the device holds (sellim.c updates liminfo[] on the host) */
/* liminfo[] / hip_liminfo_sent are touched by the wideband thread (the limiter hooks) and, with more than one CPU, by
   THREAD_TIMF2 (hip_make_timf2, wcw.c:419-425): one lock around the whole read-compare-upload / download-publish sequences */
static pthread_mutex_t hip_liminfo_lock = PTHREAD_MUTEX_INITIALIZER;
static int hip_n1, hip_n2, hip_max_batch;
/* -1: the fft2 spectrum ring is opened sparse (cfg.fft2_float_sparse: only the band fft2_mix1_fixed cuts out is stored, power sums and
   waterfall lines come out of the transform kernels) when nothing on the host reads whole fft2 spectra -- AFC off (make_afc reads power
   rows, fft2_mix1_afc whole spectra), no spur removal (acquisition and search read them), no NET_RXOUT_FFT2, one RF channel.
   0: always every bin (the test harness sets it: it compares the device ring itself).  fft1_float is left full: the stage calls
   run k_fft1 + k_timf2, which write and read whole spectra; the fused kernels that need no ring belong to lrh_wideband_dsp. */
int hip_sparse_rings = -1;
static int hip_clever_mode;               /* hg.clever_bln_mode for which the blanker tables on the device were installed */
static float *hip_afc_tmp;               /* scratch of hip_afc_rows */
static int hip_dev_spurs;                     /* length of the device's spur list as the glue left it (acquire: +1, permute: n) */
static int hip_spurs_on, hip_spur_pnt = -1;  /* spur removal served; the bins store_new_spur was last asked to take */
static lrh_spur *hip_sp; static int *hip_spsrc; static int hip_spcap;   /* scratch of the spur hooks: genparm[MAX_NO_OF_SPURS] + 1 entries (buf.c:1102 clamps the parameter
                                                                          to fftx_size / SPUR_WIDTH), allocated by hip_open; the hooks run on THREAD_SECOND_FFT / spur_removal only */
static pthread_mutex_t hip_phasing_lock = PTHREAD_MUTEX_INITIALIZER;    /* hip_ch2_c1 / _c2: up to six fft1_b workers compare and update them */
extern float spur_search_threshold;       /* spursub.c:38 */
static double hip_afc_selfreq = -2;       /* frequency around which the AFC's window of power spectra was last brought back */

/* HIPSHIM_PROF=1 (diagnostics): where hip_fft1_c's time goes, printed by hip_close */
#include <time.h>
#include <stdio.h>
static int hip_prof; static double hip_t[8]; static long hip_tn;
static double hip_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
#define HIP_T(i, stmt) do { if (hip_prof) { const double t0_ = hip_now(); stmt; hip_t[i] += hip_now() - t0_; } else { stmt; } } while (0)
static void *hip_reg[8]; static int hip_nreg;     /* host arrays page-locked for direct read-backs (hip_pin) */
static void hip_pin(void *p, size_t bytes) { if (p && bytes && hip_nreg < 8 && lrh_host_register(hip_ctx[0], p, bytes) == 0) hip_reg[hip_nreg++] = p; }
static int hip_xgather(int which, size_t count);
static void hip_spur_after_fft2(int na);
static void hip_spur_resync(void);
static void hip_open_failed(void);
/* Calls a stage thread leaves in flight on the device (lrh_stage_wait_lag): with 0 the thread alternates between enqueueing a call and waiting for it
   -- the device idles while the host enqueues and the host idles while the device works; with 1 it enqueues call n while call n-1 runs
   (round 6; Linrad's own stage threads only ever wait on the ring pointers, wcw.c:401-441, 250-304).  Measured (profiles/r06_glue_ab.txt): configs[1]
   gains 15 % at fft1_batch_n 4 with one call in flight, configs[2] loses as much -- the stage threads share one stream and one lock, and what the
   timf2 thread no longer waits for the narrowband thread then does.  Default 0 (as round 5); HIPSHIM_LAG sets it (0 .. 2). */
static int hip_lag = 0;
/* Where hip_fft1_c collects the previous call's read-backs (lag 0): HIPSHIM_LATE=1 at the END of the call, behind this call's kernels and the start of its own
   read-backs, when the copies have had a whole call to land (the library queues the sums of this call behind copies of the same rings that are still out,
   order_behind_readbacks).  Measured (round 6, two boxes x two runs): nothing beyond the spread -- the wait moves into the stage wait.  Default 0: at the start. */
static int hip_late = 0;
/* read-backs hip_fft1_c has started, per generation: the call that follows `hip_lag` later collects them (THREAD_TIMF2 / the wideband thread only: one caller) */
static int hip_ss_ticket[3][8], hip_ss_n[3], hip_ss_gen;
static int hip_wf_ticket[3][6], hip_wf_n[3], hip_wf_gen;     /* the same for hip_make_fft2 (THREAD_SECOND_FFT only) */
static void hip_ss_collect(void);
static void hip_wf_collect(void);
lrh_ctx *hip_context(void) { return hip_rx; }
lrh_ctx *hip_context_of(int ch) { return ch >= 0 && ch < HC ? hip_ctx[ch] : NULL; }

/* What versions 21 / 22 cannot serve is refused here, so that wideband_dsp ends with lirerr(1463) instead of running host code on
   rings that stay empty: the MMX / int16 back transform and second fft (their rings are short int), the correlation receiver
   (fft1_correlation_flag >= 2), spur removal with the second fft off (fft1_c would subtract from fft1_float on the host), the int16
   NET_RXOUT_TIMF2 payload, more than one mix1 channel, and two RF channels together with spur removal.
   Served: I/Q and real input, one or two RF channels (one context per channel, exchanges through host memory) with their correlation
   spectrum, spur removal with the second fft on, NET_RXOUT_FFT1 / TIMF2 (float) / FFT2. */
static int hip_unsupported(void)
{
  if (ui.rx_rf_channels != 1 && ui.rx_rf_channels != 2) return 1;
  if (ui.rx_rf_channels == 2 && genparm[MAX_NO_OF_SPURS] != 0) return 8;   /* two channels (I/Q or real): no spur removal */
  if ((ui.rx_input_mode & IQ_DATA) == 0 && fft_cntrl[FFT1_CURMODE].permute != 2) return 2;   /* real samples: version 22, whose permute field gives Linrad's
                                                                                                  filter table the real version's scaling (fft1.c:4659) */
  if (genparm[SECOND_FFT_ENABLE] != 0 && (fft_cntrl[FFT1_BCKCURMODE].mmx != 0 || fft_cntrl[FFT2_CURMODE].mmx != 0)) return 3;
  /* correlation spectrum: served for two I/Q channels (flag 1: fft1_corrsum / fft1_slowcorr / fft1_slowcorr_tot, fft1.c:4146-4150, 4584-4603);
     the correlation receiver (flag >= 2, its double-precision mixer) is not built */
  if (fft1_correlation_flag != 0 && !(fft1_correlation_flag == 1 && ui.rx_rf_channels == 2)) return 4;
  /* spur removal: eliminate_spurs inside make_fft2 with the second fft on, inside fft1_c with it off (fft1.c:4196-4244: the library takes the
     carriers out of the fft1 transforms then), acquisition through the hooks in spursub.c -- buf.c:836 zeroes MAX_NO_OF_SPURS itself when
     the AFC is off.  (Refusal 5 -- spurs on the fft1 side -- is gone: served at every fft1_size since round 5.) */
  if ((ui.network_flag & NET_RXOUT_TIMF2) != 0 && !swfloat) return 6;      /* the int16 payload is built from the MMX ring (rxin.c:968-990) */
  if (genparm[MIX1_NO_OF_CHANNELS] != 1) return 7;
  return 0;
}

int hip_open(void)
{
  lrh_config c;
  int rc;
  if ((rc = hip_unsupported()) != 0) return 100 + rc;
  lrh_config_defaults(&c, fft1_n, fft2_n);
  c.device = gpu.fft1_device;
  c.fft1_sinpow = genparm[FIRST_FFT_SINPOW]; c.fft1_gain = genparm[FIRST_FFT_GAIN]; c.fft1_direction = fft1_direction;
  c.fft_avg1num = wg.fft_avg1num; c.fft_avg2num = wg_fft_avg2num;
  c.timf1_bytes = timf1_bytes; c.max_fft1n = max_fft1n; c.fft1_sumsq_bufsize = fft1_sumsq_bufsize;
  c.wg_xpoints = wg.xpoints;
  c.slowsum_fresh_recalc = genparm[FIRST_FFT_BANDWIDTH] > 200 ? (wg_fft_avg2num < 100 ? 4 : 8) : 2;       /* fft1.c:4546-4567 */
  c.bckfft_att_n = genparm[FIRST_BCKFFT_ATT_N]; c.timf2pow_size = timf2pow_size;
  c.stupid_bln_mode = hg.stupid_bln_mode; c.stupid_bln_factor = hg.stupid_bln_factor;
  c.blnfit_range = blnfit_range; c.blanker_pulsewidth = blanker_pulsewidth;
  c.timf2_noise_floor_avgnum = timf2_noise_floor_avgnum; c.blanker_info_update_interval = blanker_info_update_interval;
  c.blanker_min_points = (int)(min_delay_time * ui.rx_ad_speed);                                           /* blank1.c:712 */
  c.timf2_noise_floor = timf2_noise_floor;
  c.fft2_sinpow = genparm[SECOND_FFT_SINPOW]; c.max_fft2n = max_fft2n;
  c.waterfall_avgnum = wg.waterfall_avgnum; c.wf_first_xpoint = hgwat_first_xpoint; c.wf_xpixels = wg_xpixels;
  c.wf_mode = hgwat_xpoints_per_pixel == 1 || hgwat_pixels_per_xpoint == 1 ? 1 :
              (hgwat_xpoints_per_pixel > 1 ? hgwat_xpoints_per_pixel : -hgwat_pixels_per_xpoint);        /* fft2.c:723-811 */
  c.wf_lines = wg_waterf_size / wg_xpixels;
  c.mix1_bandwidth_reduction_n = genparm[MIX1_BANDWIDTH_REDUCTION_N]; c.timf3_size = timf3_size;
  c.fftx_points_per_hz = fftx_points_per_hz; c.mix1_lowest_fq = mix1_lowest_fq; c.mix1_highest_fq = mix1_highest_fq;
  /* transforms per library call: fft1_b hands over gpu_fft1_batch_size at a time (buf.c:248-257); the stages behind it take whatever has
     accumulated when their thread comes round (hip_fft1_c), up to what the rings allow -- the library wants max_fft1n >= 2 max_batch and a
     batch of new points beside one transform in the timf2 ring */
  c.max_batch = gpu_fft1_batch_size > 0 ? gpu_fft1_batch_size : 1;
  { int cap = 64;                                           /* (HIPSHIM_MAX_BATCH: 128 measured no faster, profiles/r06_glue_ab.txt -- a call carries 27-40 blocks whatever the cap) */
    { const char *e = getenv("HIPSHIM_MAX_BATCH"); if (e && atoi(e) > 0) cap = atoi(e); }
    if (cap > max_fft1n / 2) cap = max_fft1n / 2;
    if (genparm[SECOND_FFT_ENABLE] != 0) while (cap > 1 && (long long)cap * fft1_new_points + fft1_size > (long long)timf2pow_size) cap >>= 1;
    if (ui.rx_rf_channels == 2) cap = c.max_batch;          /* two channels: the exchange buffers are sized by the batch; one hand-over at a time */
    if (cap > c.max_batch) c.max_batch = cap; }
  hip_max_batch = c.max_batch;
  c.second_fft_enable = genparm[SECOND_FFT_ENABLE];
  if (hip_sparse_rings != 0 && ui.rx_rf_channels == 1 && genparm[SECOND_FFT_ENABLE] != 0 && genparm[AFC_ENABLE] == 0 && genparm[MAX_NO_OF_SPURS] == 0 &&
      (ui.network_flag & NET_RXOUT_FFT2) == 0) c.fft2_float_sparse = 1;
  c.timf2_blockpower_block = timf2_blockpower_block; c.timf2_blockpower_size = timf2_blockpower_size;     /* compute_timf2_powersum, wcw.c:80 */
  c.timf1_dword_input = (ui.rx_input_mode & DWORD_INPUT) != 0; c.sample_shift = ui.sample_shift;
  c.timf1_real_input = (ui.rx_input_mode & IQ_DATA) == 0;          /* fft1_reherm_dit_one's job (fft1_re.c:32-131): 2 fft1_size reals per transform */
  HC = ui.rx_rf_channels;
  hip_real2 = HC == 2 && (ui.rx_input_mode & IQ_DATA) == 0;
  hip_n1 = fft1_size; hip_n2 = fft2_size; hip_afc_selfreq = -2;
  { const char *e = getenv("HIPSHIM_LATE"); hip_late = e ? atoi(e) != 0 : 0; }
  { const char *e = getenv("HIPSHIM_LAG"); hip_lag = e ? atoi(e) : 0; if (hip_lag < 0) hip_lag = 0; if (hip_lag > 2) hip_lag = 2; }
  hip_ss_gen = hip_wf_gen = 0;
  hip_prof = getenv("HIPSHIM_PROF") != NULL; memset(hip_t, 0, sizeof hip_t); hip_tn = 0;
  hip_ctx[0] = hip_ctx[1] = NULL;
  for (int ch = 0; ch < HC; ch++) {
    if (HC == 2) {                                                 /* one context per channel, coupled (include/linrad_hip.h) */
      c.blanker_channels = 2; c.timf1_frame_channels = 2; c.timf1_channel_index = ch;
      if (hip_real2) { c.timf1_frame_channels = 1; c.timf1_bytes = timf1_bytes / 2; }   /* its own de-interleaved arena */
      c.timf3_size = timf3_size / 2; c.timf2_blockpower_block = timf2_blockpower_block / 2;
    }
    if ((rc = lrh_open(&c, &hip_ctx[ch])) != 0) { hip_ctx[ch] = NULL; hip_rx = hip_ctx[0]; hip_open_failed(); return rc; }
    if (HC == 1) lrh_set_filtercorr(hip_ctx[ch], fft1_filtercorr);  /* the calibration Linrad loaded (fft1.c:4653-5386) */
    else {                                                         /* two channels: {ch0, ch1} per bin (fft1.c:4132-4145) */
      float *f = hip_scratch(2 * (size_t)fft1_size);
      for (int i = 0; i < fft1_size; i++) { f[2 * i] = fft1_filtercorr[4 * i + 2 * ch]; f[2 * i + 1] = fft1_filtercorr[4 * i + 2 * ch + 1]; }
      lrh_set_filtercorr(hip_ctx[ch], f);
    }
    lrh_set_liminfo(hip_ctx[ch], liminfo);
  }
  hip_rx = hip_ctx[0];
  hip_ch2_c1 = 1; hip_ch2_c2 = 0;
  hip_liminfo_sent = malloc(sizeof(float) * (size_t)fft1_size);
  memcpy(hip_liminfo_sent, liminfo, sizeof(float) * (size_t)fft1_size);
  hip_spurs_on = genparm[AFC_ENABLE] != 0 && genparm[MAX_NO_OF_SPURS] != 0;
  if (hip_spurs_on) {
    hip_spcap = genparm[MAX_NO_OF_SPURS] + 1;
    hip_sp = malloc(sizeof(lrh_spur) * (size_t)hip_spcap); hip_spsrc = malloc(sizeof(int) * (size_t)hip_spcap);
    if (!hip_sp || !hip_spsrc) { hip_open_failed(); return LRH_ENOMEM; }
    if ((rc = lrh_spur_config(hip_rx, genparm[MAX_NO_OF_SPURS], spur_speknum, spur_spectra)) != 0) { hip_open_failed(); return rc; }
    /* the search spectrum for new spurs is kept and cleaned on the device (fft2.c:673-699, spursearch_spectrum_cleanup) */
    if ((rc = lrh_spur_search_config(hip_rx, spur_search_first_point, spur_search_last_point)) != 0) { hip_open_failed(); return rc; }
  }
  if (HC == 2 && fft1_correlation_flag == 1)
    for (int ch = 0; ch < 2; ch++) if (lrh_set_correlation(hip_ctx[ch], 1) != 0) { hip_open_failed(); return LRH_EINVAL; }
  if (hip_real2) for (int ch = 0; ch < 2; ch++) { hip_deint[ch] = malloc((size_t)timf1_bytes / 2); if (!hip_deint[ch]) { hip_open_failed(); return LRH_ENOMEM; } lrh_host_register(hip_ctx[ch], hip_deint[ch], (size_t)timf1_bytes / 2); }
  else
  for (int ch = 0; ch < HC; ch++) lrh_host_register(hip_ctx[ch], timf1_char, (size_t)timf1_bytes); /* the timf1 arena, page-locked once; the shim never frees it (buf.c:2105) */
  /* ... and the host arrays the display data come back into (round 6): a read-back whose destination is page-locked is written by the copy engine
     itself -- no staging slot, no memcpy on the stage thread that asked for it (THREAD_TIMF2 spent more than half of hip_fft1_c in those).  A span the
     runtime refuses (pages shared with a neighbour that is registered already) simply stays pageable: the library then stages as before. */
  if (HC == 1) {
    hip_nreg = 0;
    hip_pin(fft1_sumsq, sizeof(float) * (size_t)fft1_sumsq_bufsize); hip_pin(fft1_slowsum, sizeof(float) * (size_t)fft1_size);
    if (genparm[SECOND_FFT_ENABLE] != 0) { hip_pin(wg_waterf, sizeof(short int) * (size_t)wg_waterf_size); hip_pin(fft2_powersum_float, sizeof(float) * (size_t)fft2_size); }
    hip_pin(timf3_float, sizeof(float) * (size_t)timf3_size);
  }
  return 0;
}

/* Everything hip_open has made so far goes: contexts (hip_rx may still be NULL when the failure came before it was set), page-locks, scratch.
   hip_close is the same with the pending read-backs collected first. */
static void hip_release(void)
{
  hip_clever_mode = 0;
  if (hip_ctx[0]) while (hip_nreg > 0) lrh_host_unregister(hip_ctx[0], hip_reg[--hip_nreg]);
  hip_nreg = 0;
  for (int ch = 0; ch < 2; ch++) if (hip_ctx[ch]) {
    lrh_timf1_write_wait(hip_ctx[ch]);
    lrh_host_unregister(hip_ctx[ch], hip_real2 && hip_deint[ch] ? (void *)hip_deint[ch] : (void *)timf1_char);    /* not registered yet: refused, harmless */
    lrh_close(hip_ctx[ch]); hip_ctx[ch] = NULL;
  }
  for (int ch = 0; ch < 2; ch++) { free(hip_deint[ch]); hip_deint[ch] = NULL; }
  hip_real2 = 0;
  hip_rx = NULL; HC = 1;
  free(hip_xa); free(hip_xb); hip_xa = hip_xb = NULL; hip_xcap = 0;
  free(hip_liminfo_sent); hip_liminfo_sent = NULL;
  free(hip_afc_tmp); hip_afc_tmp = NULL;
  free(hip_sp); free(hip_spsrc); hip_sp = NULL; hip_spsrc = NULL; hip_spcap = 0;
  hip_spurs_on = 0; hip_spur_pnt = -1; hip_dev_spurs = 0;
}
static void hip_open_failed(void) { memset(hip_ss_n, 0, sizeof hip_ss_n); memset(hip_wf_n, 0, sizeof hip_wf_n); hip_release(); }

void hip_close(void)
{
  if (!hip_rx) return;
  if (hip_prof && hip_tn) fprintf(stderr, "HIPSHIM_PROF hip_fft1_c: %ld calls, %.1f blocks each; per call: stage wait %.1f us, collect %.1f, lrh_fft1_c %.1f, sumsq fetches %.1f (%.1f of them), slowsum fetch %.1f\n", hip_tn, hip_t[5] / hip_tn, hip_t[0] / hip_tn, hip_t[1] / hip_tn, hip_t[2] / hip_tn, hip_t[3] / hip_tn, hip_t[6] / hip_tn, hip_t[4] / hip_tn);
  for (int g = 0; g < 3; g++) {                               /* every generation still out */
    for (int i = 0; i < hip_ss_n[g]; i++) lrh_export_end(hip_rx, hip_ss_ticket[g][i]);
    for (int i = 0; i < hip_wf_n[g]; i++) lrh_export_end(hip_rx, hip_wf_ticket[g][i]);
    hip_ss_n[g] = hip_wf_n[g] = 0;
  }
  hip_release();
}

void hip_timf1_new(int pa, int nbytes)
{
  if (hip_real2 && hip_rx) {                                /* {a_k, b_k} -> a_k | b_k, each into its context's arena at the same ring position / 2 */
    const int es = (ui.rx_input_mode & DWORD_INPUT) != 0 ? 4 : 2, n = nbytes / (2 * es);
    for (int ch = 0; ch < 2; ch++) {
      char *d = hip_deint[ch] + pa / 2;
      const char *sp = &timf1_char[pa] + ch * es;
      if (es == 2) for (int i = 0; i < n; i++) ((short *)d)[i] = *(const short *)(sp + 4 * i);
      else for (int i = 0; i < n; i++) ((int *)d)[i] = *(const int *)(sp + 8 * i);
      if (lrh_timf1_write_async(hip_ctx[ch], d, pa / 2, nbytes / 2) != 0) lirerr(1465);
    }
    return;
  }
  for (int ch = 0; ch < HC && hip_rx; ch++)               /* two channels: both contexts hold the interleaved frames and read their own channel */
    if (lrh_timf1_write_async(hip_ctx[ch], &timf1_char[pa], pa, nbytes) != 0) lirerr(1465);
}

int hip_fft1_b(int timf1p_ref, float *out, int gpu_handle_number)
{
  /* the dispatcher's workers carry gpu_handle_number 0..5 (wcw.c:500), the no-worker path passes 0 too (wcw.c:1036) */
  const int handle = no_of_fft1b > 0 ? gpu_handle_number + 1 : 0;
  int rc = 0;
  /* (two real channels: fft1_reherm_dit_two leaves through `goto fft_done` before the phasing step, fft1_re.c:133-231) */
  if (HC == 2 && !hip_real2) {                               /* phasing of channel 2, fft1.c:4064-4080 (pol_graph.c:160-170 sets it) */
    pthread_mutex_lock(&hip_phasing_lock);
    if (pg_ch2_c1 != hip_ch2_c1 || pg_ch2_c2 != hip_ch2_c2) {
      hip_ch2_c1 = pg_ch2_c1; hip_ch2_c2 = pg_ch2_c2;
      lrh_set_ch2_phasing(hip_ctx[1], hip_ch2_c1, hip_ch2_c2);
    }
    pthread_mutex_unlock(&hip_phasing_lock);
  }
  for (int ch = 0; ch < HC && !rc; ch++) rc = lrh_fft1_b(hip_ctx[ch], handle, hip_real2 ? timf1p_ref / 2 : timf1p_ref, HIP_IN((int)(out - fft1_float)), gpu_fft1_batch_size);
  return rc;
}

/* make_afc (afc_graph.c:362) stays host code: with ag.mode_control != 0 it searches the power spectra fftx_pwr[transform][bin] of
   the recent transforms in a window around the selected frequency (collect_initial_spectrum and friends, afcsub.c:60-90, 849;
   one RF channel: powers only).  Those spectra live on the device, so the window -- not the spectra -- comes back: the new
   transform's row when one has been made, every row of the ring when the operator has moved the frequency.  With
   ag.mode_control == 0 make_afc only fills mix1_fq_mid with the selected frequency and nothing is fetched. */
static void hip_afc_rows(int first_row, int rows)
{
  const int second = genparm[SECOND_FFT_ENABLE] != 0;
  const int size = second ? hip_n2 : hip_n1, nmask = second ? fft2n_mask : fft1n_mask;
  int centre, half, lo, hi, r, i;
  if (genparm[AFC_ENABLE] == 0 || genparm[AFC_LOCK_RANGE] == 0 || ag.mode_control == 0 || mix1_selfreq[0] < 0 || fftx_pwr == NULL) return;
  if (HC == 2) return;                                     /* two channels: make_afc searches TWOCHAN_POWER rows (afcsub.c): not fetched -- the AFC then holds the selected frequency */
  if (hip_afc_selfreq != mix1_selfreq[0]) { hip_afc_selfreq = mix1_selfreq[0]; first_row = 0; rows = nmask + 1; }
  centre = (int)(mix1_selfreq[0] * fftx_points_per_hz);
  half = 4 * max_afcf_points + 64;
  lo = centre - half; if (lo < 0) lo = 0;
  hi = centre + half; if (hi > size) hi = size;
  if (hi <= lo) return;
  for (r = 0; r < rows; r++) {
    const int row = (first_row + r) & nmask;
    if (second) { lrh_export(hip_rx, LRH_RING_FFT2_POWER, &fftx_pwr[(size_t)row * size + lo], (size_t)row * size + lo, (size_t)(hi - lo)); continue; }
    /* second fft off: fftx_pwr = fft1_power, |corrected fft1 bin|^2 as fft1_c forms it (fft1.c:4431-4440) */
    if (!hip_afc_tmp) hip_afc_tmp = malloc(sizeof(float) * 2 * (size_t)hip_n1);
    lrh_export(hip_rx, LRH_RING_FFT1_FLOAT, hip_afc_tmp, 2 * ((size_t)row * size + lo), 2 * (size_t)(hi - lo));
    for (i = 0; i < hi - lo; i++) fftx_pwr[(size_t)row * size + lo + i] = hip_afc_tmp[2 * i] * hip_afc_tmp[2 * i] + hip_afc_tmp[2 * i + 1] * hip_afc_tmp[2 * i + 1];
  }
}

static void hip_ss_collect(void)                             /* moves on to the oldest generation and brings it in; the caller's new read-backs go there */
{
  hip_ss_gen = (hip_ss_gen + 1) % (hip_lag + 1);
  for (int i = 0; i < hip_ss_n[hip_ss_gen]; i++) if (lrh_export_end(hip_rx, hip_ss_ticket[hip_ss_gen][i]) != 0) lirerr(1466);
  hip_ss_n[hip_ss_gen] = 0;
}
static void hip_ss_collect_late(void)                        /* two generations: this call's read-backs are in hip_ss_gen, the previous call's in the other one */
{
  const int prev = hip_ss_gen ^ 1;
  for (int i = 0; i < hip_ss_n[prev]; i++) if (lrh_export_end(hip_rx, hip_ss_ticket[prev][i]) != 0) lirerr(1466);
  hip_ss_n[prev] = 0;
  hip_ss_gen = prev;                                          /* the next call's go there */
}
static void hip_ss_fetch(lrh_ring ring, float *dst, size_t off, size_t cnt)
{
  int t = 0;
  if (hip_ss_n[hip_ss_gen] >= 8) { lrh_export(hip_rx, ring, dst, off, cnt); return; }
  if (lrh_export_begin(hip_rx, ring, dst, off, cnt, &t) != 0) { lirerr(1466); return; }
  if (t) hip_ss_ticket[hip_ss_gen][hip_ss_n[hip_ss_gen]++] = t;
}

void hip_fft1_c(void)
{
  lrh_ptrs q;
  int old_pa, n, room;
  memset(&q, 0, sizeof q);
  /* hip_sync_in: what fft1_c reads (fft1.c:4507-4523) */
  old_pa = fft1_sumsq_pa;
  /* back-pressure: not further ahead of the device than one call of this stage (lrh_stage_wait, include/linrad_hip.h) -- what arrives
     meanwhile goes into this call */
  HIP_T(0, for (int ch = 0; ch < HC; ch++) lrh_stage_wait_lag(hip_ctx[ch], LRH_STAGE_TIMF2, HC == 1 ? hip_lag : 0));   /* (two channels: the exchanges go through host memory, one call at a time) */
  hip_tn++;
  if (!(hip_late && hip_lag == 0 && HC == 1)) HIP_T(1, hip_ss_collect());   /* the previous call's spectra: on the host by now (HIPSHIM_LATE: at the end of this call) */
  /* Every transform fft1_b has delivered goes through in one call: both callers loop `while(fft1_na != fft1_nb){do_fft1_c();
     make_timf2();}` (wcw.c:421-425, 1096-1101), which then ends after one pass -- a call costs the device a fixed latency chain
     whatever its size (INTEGRATION.md "Call size"), and the limiter looks at fft1_liminfo_cnt only after that loop (wcw.c:1124). */
  n = (fft1_na - fft1_nb + max_fft1n) & fft1n_mask;
  /* what hip_make_timf2 can place (the callers test room for one, wcw.c:419); with the second fft off nothing follows in timf2 */
  room = genparm[SECOND_FFT_ENABLE] != 0 ? ((timf2_px - timf2_pa + timf2_mask + 1) & timf2_mask) / timf2_input_block - 1 : n;
  if (n > room) n = room;
  if (n > hip_max_batch) n = hip_max_batch;
  if (hip_spurs_on && genparm[SECOND_FFT_ENABLE] == 0) n = 1;   /* spurs tracked in the fft1 transforms: Linrad looks at the loop state after every one */
  { /* ... and the periods it completes, beside the wg_fft_avg2num the slow average reads, must fit the fft1_sumsq ring (lrh_fft1_c) */
    const int lim = wg.fft_avg1num * (fft1_sumsq_bufsize / fft1_size - wg_fft_avg2num - 1) - fft1_sumsq_counter;
    if (n > lim) n = lim; }
  if (n < 1) n = 1;
  q.fft1_nb = fft1_nb; q.fft1_pb = HIP_IN(fft1_pb); q.fft1_sumsq_pa = fft1_sumsq_pa; q.fft1_sumsq_counter = fft1_sumsq_counter;
  q.fft1_liminfo_cnt = fft1_liminfo_cnt; q.fft1_sumsq_recalc = fft1_sumsq_recalc;
  { lrh_ptrs q0 = q;
    if (hip_spurs_on && genparm[SECOND_FFT_ENABLE] == 0) hip_spur_resync();   /* (the spur loop runs inside lrh_fft1_c then) */
    { const double t0_ = hip_prof ? hip_now() : 0;
    for (int ch = 0; ch < HC; ch++) { q = q0; if (lrh_fft1_c(hip_ctx[ch], &q, n) != 0) { lirerr(1466); return; } }
    if (hip_prof) { hip_t[2] += hip_now() - t0_; hip_t[5] += n; } }
    if (HC == 2 && fft1_correlation_flag == 1) {       /* X conj(Y) needs both channels' bins: the all-gather of the batch's transforms, through host memory */
      size_t cnt[2] = { 0, 0 };
      if (correlation_reset_flag != fft1corr_reset_flag) {   /* the operator's reset (fft1.c:4586): host arrays by Linrad's own code, the device's by switching the mode on again */
        clear_fft1_correlation();
        for (int ch = 0; ch < 2; ch++) if (lrh_set_correlation(hip_ctx[ch], 1) != 0) { lirerr(1466); return; }
      }
      for (int ch = 0; ch < 2; ch++) if (lrh_fft1_corr_begin(hip_ctx[ch], &q0, n, &cnt[ch]) != 0) { lirerr(1466); return; }
      if (cnt[0] != cnt[1] || hip_xgather(LRH_X_SPEC, cnt[0]) != 0) { lirerr(1466); return; }
      for (int ch = 0; ch < 2; ch++) if (lrh_fft1_corr_finish(hip_ctx[ch], &q0, n) != 0) { lirerr(1466); return; }
    } }
  /* hip_sync_out */
  fft1_nb = q.fft1_nb; fft1_pb = HIP_OUT(q.fft1_pb); fft1_sumsq_pa = q.fft1_sumsq_pa; fft1_sumsq_counter = q.fft1_sumsq_counter;
  fft1_liminfo_cnt = q.fft1_liminfo_cnt; fft1_sumsq_recalc = q.fft1_sumsq_recalc;
  if (genparm[SECOND_FFT_ENABLE] == 0) hip_afc_rows((q.fft1_nb - n) & fft1n_mask, n);
  if (genparm[SECOND_FFT_ENABLE] == 0 && hip_spurs_on) {     /* what fft1_c's AFC branch leaves for spur_removal (fft1.c:4206-4207), then the spur display and the search walk */
    ffts_na = (q.fft1_nb - 1) & fft1n_mask; ffts_nm = fft1_nm;
    hip_spur_after_fft2(ffts_na);
  }
  if (q.fft1_sumsq_pa != old_pa) {            /* averaging periods completed: the wide graph and sellim.c read these on the host */
    int pa;
    if (HC == 1) {
      /* The completed periods lie one behind the other: as few read-backs as the ring's end and the slot size allow.  They are for the wide
         graph (the limiter works on the device-resident sums), so this thread does not wait for them: they are collected by its next call
         (hip_ss_collect) -- a wait here would drain the device once per call with the weak / strong split not even queued yet */
      const int slot = (1 << 17) / hip_n1 > 0 ? (1 << 17) / hip_n1 * hip_n1 : hip_n1;     /* floats per read-back (512 KiB slots, include/linrad_hip.h) */
      pa = old_pa;
      while (pa != q.fft1_sumsq_pa) {
        int k = ((q.fft1_sumsq_pa - pa) & fft1_sumsq_mask);
        if (pa + k > fft1_sumsq_bufsize) k = fft1_sumsq_bufsize - pa;
        if (k > slot) k = slot;
        HIP_T(3, hip_ss_fetch(LRH_RING_FFT1_SUMSQ, &fft1_sumsq[pa], (size_t)pa, (size_t)k)); hip_t[6] += 1;
        pa = (pa + k) & fft1_sumsq_mask;
      }
      HIP_T(4, hip_ss_fetch(LRH_RING_FFT1_SLOWSUM, fft1_slowsum, 0, (size_t)hip_n1));
      if (hip_late && hip_lag == 0) HIP_T(1, hip_ss_collect_late());
      return;
    } else
    for (pa = old_pa; pa != q.fft1_sumsq_pa; pa = (pa + hip_n1) & fft1_sumsq_mask) {
      lrh_export(hip_rx, LRH_RING_FFT1_SUMSQ, &fft1_sumsq[pa], (size_t)pa, (size_t)hip_n1);
      {                                                    /* |X0|^2 + |X1|^2 (fft1.c:4132-4145) */
        float *t = hip_scratch((size_t)hip_n1);
        lrh_export(hip_ctx[1], LRH_RING_FFT1_SUMSQ, t, (size_t)pa, (size_t)hip_n1);
        for (int i = 0; i < hip_n1; i++) fft1_sumsq[pa + i] += t[i];
      }
    }
    lrh_export(hip_rx, LRH_RING_FFT1_SLOWSUM, fft1_slowsum, 0, (size_t)hip_n1);
    if (HC == 2) {
      float *t = hip_scratch((size_t)hip_n1);
      lrh_export(hip_ctx[1], LRH_RING_FFT1_SLOWSUM, t, 0, (size_t)hip_n1);
      for (int i = 0; i < hip_n1; i++) fft1_slowsum[i] += t[i];
    }
    if (HC == 2 && fft1_correlation_flag == 1) {       /* both contexts hold the same rings: the wide graph's correlation display reads these (wide_graph.c) */
      for (pa = old_pa; pa != q.fft1_sumsq_pa; pa = (pa + hip_n1) & fft1_sumsq_mask)
        lrh_export(hip_rx, LRH_RING_FFT1_CORRSUM, &fft1_corrsum[2 * pa], (size_t)2 * pa, (size_t)2 * hip_n1);
      lrh_export(hip_rx, LRH_RING_FFT1_SLOWCORR, fft1_slowcorr, 0, (size_t)2 * hip_n1);
      lrh_export(hip_rx, LRH_RING_FFT1_SLOWCORR_TOT, fft1_slowcorr_tot, 0, (size_t)2 * hip_n1);
      lrh_get_slowcorr_tot_avgnum(hip_rx, &slowcorr_tot_avgnum);
    }
  }
}

void hip_make_timf2(void)
{
  lrh_ptrs q;
  int n;
  memset(&q, 0, sizeof q);
  /* host code has rewritten liminfo[] (a GUI reset): the device gets the new table.
     The limiter hooks below publish the device's own table under the same lock, so a half-published table is never taken for a change. */
  pthread_mutex_lock(&hip_liminfo_lock);
  if (memcmp(hip_liminfo_sent, liminfo, sizeof(float) * (size_t)hip_n1)) {
    memcpy(hip_liminfo_sent, liminfo, sizeof(float) * (size_t)hip_n1);
    for (int ch = 0; ch < HC; ch++) lrh_set_liminfo(hip_ctx[ch], hip_liminfo_sent);
  }
  pthread_mutex_unlock(&hip_liminfo_lock);
  q.fft1_px = HIP_IN(fft1_px); q.fft1_nx = fft1_nx; q.timf2_pa = HIP_IN(timf2_pa);
  q.fft1_lowlevel_points = fft1_lowlevel_points; q.fft1_lowlevel_fraction = fft1_lowlevel_fraction;
  /* all the transforms hip_fft1_c has just passed (it left no more than the timf2 ring has room for) */
  n = (fft1_nb - fft1_nx + max_fft1n) & fft1n_mask;
  if (n > hip_max_batch) n = hip_max_batch;
  if (n < 1) n = 1;
  { lrh_ptrs q0 = q;
    for (int ch = 0; ch < HC; ch++) { q = q0; if (lrh_make_timf2(hip_ctx[ch], &q, n) != 0) { lirerr(1467); return; } } }
  fft1_px = HIP_OUT(q.fft1_px); fft1_nx = q.fft1_nx; timf2_pa = HIP_OUT(q.timf2_pa);            /* timf2.c:127-128, 205-207 */
  fft1_lowlevel_points = q.fft1_lowlevel_points; fft1_lowlevel_fraction = q.fft1_lowlevel_fraction;
}

/* Two channels: the blanker decides on the channels' power SUM and averages both channels' noise (blank1.c:1017, 1236-1300, 1510-1570);
   with the tables of the linear blanker both contexts run the same pulse search on both channels' samples.  The sums and the gather
   are the caller's (include/linrad_hip.h): here through host memory.  Both contexts advance the same pointers. */
static int hip_xsum(int which, size_t count)
{
  float *a = hip_scratch(count), *b = hip_xb;
  if (lrh_exchange_read(hip_ctx[0], which, a, 0, count) != 0 || lrh_exchange_read(hip_ctx[1], which, b, 0, count) != 0) return 1;
  for (size_t i = 0; i < count; i++) a[i] += b[i];
  return lrh_exchange_write(hip_ctx[0], which, a, 0, count) != 0 || lrh_exchange_write(hip_ctx[1], which, a, 0, count) != 0;
}
static int hip_xgather(int which, size_t count)            /* slot ch of context ch to slot ch of the other one */
{
  float *a = hip_scratch(count);
  for (int ch = 0; ch < 2; ch++)
    if (lrh_exchange_read(hip_ctx[ch], which, a, (size_t)ch * count, count) != 0 || lrh_exchange_write(hip_ctx[1 - ch], which, a, (size_t)ch * count, count) != 0) return 1;
  return 0;
}
static int hip_coupled_blanker(lrh_ptrs *q)
{
  const lrh_ptrs q0 = *q;
  int n[2] = { 0, 0 };
  size_t nw = 0;
  for (int ch = 0; ch < 2; ch++) if (lrh_blanker_begin(hip_ctx[ch], &q0, &n[ch]) != 0) return 1;
  if (n[0] != n[1]) return 1;
  if (n[0] > 0 && hip_xsum(LRH_X_PWR, (size_t)n[0]) != 0) return 1;
  if (lrh_blanker_weak_span(hip_rx, &nw) != 0) return 1;
  if (nw > 0 && hip_xgather(LRH_X_WEAK, nw) != 0) return 1;
  for (int ch = 0; ch < 2; ch++) { *q = q0; if (lrh_first_noise_blanker(hip_ctx[ch], q) != 0) return 1; }
  if (n[0] > 0) {
    if (hip_xsum(LRH_X_STAT, 2) != 0) return 1;
    for (int ch = 0; ch < 2; ch++) { lrh_ptrs t = *q; if (lrh_blanker_finish(hip_ctx[ch], &t) != 0) return 1; if (ch == 1) *q = t; }
  }
  return 0;
}

/* the linear blanker's tables follow hg.clever_bln_mode (hires_graph.c:496-498 toggles it; init_blanker, buf.c:1771, built them) */
static void hip_blanker_tables(void)
{
  lrh_blanker_tables t;
  int i;
  if (hg.clever_bln_mode == hip_clever_mode) return;
  hip_clever_mode = hg.clever_bln_mode;
  if (hg.clever_bln_mode == 0 || refpul_size == 0) { for (int ch = 0; ch < HC; ch++) lrh_set_blanker_tables(hip_ctx[ch], NULL); return; }
  memset(&t, 0, sizeof t);
  t.clever_bln_mode = hg.clever_bln_mode; t.clever_bln_factor = hg.clever_bln_factor; t.clever_bln_limit = hg.clever_bln_limit;
  t.refpul_size = refpul_size; t.largest_blnfit = largest_blnfit; t.liminfo_amplitude_factor = liminfo_amplitude_factor;
  for (i = 0; i < BLN_INFO_SIZE && i < LRH_BLN_INFO_SIZE; i++) { t.bln[i].size = bln[i].size; t.bln[i].rest = bln[i].rest; t.bln[i].avgmax = bln[i].avgmax; }
  t.refpulse = blanker_refpulse; t.phasefunc = blanker_phasefunc; t.pulindex = blanker_pulindex;
  for (int ch = 0; ch < HC; ch++) if (lrh_set_blanker_tables(hip_ctx[ch], &t) != 0) lirerr(1472);
}

void hip_first_noise_blanker(void)
{
  lrh_ptrs q;
  memset(&q, 0, sizeof q);
  q.timf2p_fit = timf2p_fit; q.timf2_pn2 = HIP_IN(timf2_pn2); q.timf2_pa = HIP_IN(timf2_pa); q.timf2_blanker_points = timf2_blanker_points;
  q.blanker_info_update_counter = blanker_info_update_counter; q.fft1_lowlevel_fraction = fft1_lowlevel_fraction;
  hip_blanker_tables();
  if (HC == 1) { if (lrh_first_noise_blanker(hip_rx, &q) != 0) { lirerr(1468); return; } }
  else if (hip_coupled_blanker(&q) != 0) { lirerr(1468); return; }
  timf2p_fit = q.timf2p_fit; timf2_pn2 = HIP_OUT(q.timf2_pn2); timf2_blanker_points = q.timf2_blanker_points;     /* blank1.c:1458-1476 */
  if (q.blanker_info_update_counter == 0 && blanker_info_update_counter != 0) {   /* thresholds were updated (blank1.c:1550-1601) */
    lrh_blanker_state bs;
    if (lrh_get_blanker_state(hip_rx, &bs) == 0) {
      timf2_noise_floor = bs.timf2_noise_floor; hg.stupid_bln_limit = bs.stupid_bln_limit;
      stupid_blanker_rate = bs.stupid_blanker_rate; clever_blanker_rate = bs.clever_blanker_rate; hg.clever_bln_limit = bs.clever_bln_limit;
      timf2_despiked_pwr[0] = bs.timf2_despiked_pwr[0]; timf2_despiked_pwr[1] = bs.timf2_despiked_pwr[1];
      timf2_despiked_pwrinc[0] = bs.timf2_despiked_pwrinc[0]; timf2_despiked_pwrinc[1] = bs.timf2_despiked_pwrinc[1];
      timf2_cleared_points = bs.timf2_cleared_points; timf2_fitted_pulses = bs.timf2_fitted_pulses;
    }
  }
  blanker_info_update_counter = q.blanker_info_update_counter;
}

/* fft1_update_liminfo (sellim.c:738): the table is formed on the device from the resident fft1_sumsq / fft1_slowsum; the host's
   liminfo[] is refreshed for the high-resolution graph, which draws it */
static void hip_sellim_par(lrh_sellim *par, lrh_ptrs *q)
{
  memset(par, 0, sizeof *par); memset(q, 0, sizeof *q);
  par->struct_size = (int)sizeof *par;
  par->sellim_maxlevel = genparm[SELLIM_MAXLEVEL]; par->spek_avgnum = wg.spek_avgnum;
  par->fft1_blocktime = fft1_blocktime; par->blanker_ston_fft1 = hg.blanker_ston_fft1;
  par->sellim_par2 = hg.sellim_par2; par->sellim_par3 = hg.sellim_par3; par->sellim_par4 = hg.sellim_par4; par->sellim_par5 = hg.sellim_par5;
  par->sellim_par6 = hg.sellim_par6; par->sellim_par7 = hg.sellim_par7; par->sellim_par8 = hg.sellim_par8;
  par->liminfo_group_points = liminfo_group_points;
  par->fft1_first_point = fft1_first_point; par->fft1_last_point = fft1_last_point;
  par->fft1_first_inband = fft1_first_inband; par->fft1_last_inband = fft1_last_inband;
  par->baseband_bw_fftxpts = baseband_bw_fftxpts; par->ston_scale = mg.scale_type == MG_SCALE_STON;
  par->exact_stats = 0;
  par->blanker_ston_fft2 = hg.blanker_ston_fft2; par->fft2_blocktime = fft2_blocktime; par->sellim_par1 = hg.sellim_par1;
  par->fft1_desired = (fft1_calibrate_flag & CALAMP) == CALAMP ? fft1_desired : NULL;     /* sellim.c:134-143 */
  /* in the middle of an averaging period (a batch of gpu_fft1_batch_size transforms need not end on one) the slot at fft1_sumsq_pa holds
     unfinished sums: the counter tells the library to read the newest finished period instead (include/linrad_hip.h, lrh_sellim) */
  q->fft1_sumsq_pa = fft1_sumsq_pa; q->fft1_sumsq_counter = fft1_sumsq_counter;
  lrh_set_mix1_selfreq(hip_rx, mix1_selfreq[0]);             /* selfreq_liminfo protects the selected passband (sellim.c:38) */
}
static void hip_liminfo_back(void)
{
  /* the host's liminfo[] for the high-resolution graph; nothing to upload later.  liminfo_amplitude_factor stays on the device,
     where the linear blanker reads it (the host copy serves the GUI's "skip the smart blanker" test, sellim.c:156) */
  pthread_mutex_lock(&hip_liminfo_lock);
  if (lrh_get_liminfo(hip_rx, hip_liminfo_sent) == 0) memcpy(liminfo, hip_liminfo_sent, sizeof(float) * (size_t)hip_n1);
  pthread_mutex_unlock(&hip_liminfo_lock);
  lrh_get_liminfo_amplitude_factor(hip_rx, &liminfo_amplitude_factor);
}
int hip_fft1_update_liminfo(void)
{
  lrh_sellim par;
  lrh_ptrs q;
  /* two channels: the reference's limiter works on the channels' summed spectra with its limit scaled by rx_rf_channels (sellim.c:786);
     hip_fft1_c has brought fft1_sumsq / fft1_slowsum (sums over both) to the host, Linrad's own code runs on them and hip_make_timf2
     uploads the table it leaves in liminfo[] to both contexts */
  if (HC == 2) return 0;
  hip_sellim_par(&par, &q);
  if (lrh_fft1_update_liminfo(hip_rx, &q, &par) != 0) { lirerr(1471); return 1; }
  hip_liminfo_back();
  return 1;
}
/* fft2_update_liminfo (sellim.c:159; all three settings of hg.sellim_par1 run on the device-resident fft2 power sums) */
int hip_fft2_update_liminfo(void)
{
  lrh_sellim par;
  lrh_ptrs q;
  if (hg.sellim_par1 < 0 || hg.sellim_par1 > 2) return 0;
  if (HC == 2) return 1;                                    /* two channels: not served (the fft2 sums live on the device; the first limiter carries on) */
  hip_sellim_par(&par, &q);
  if (lrh_fft2_update_liminfo(hip_rx, &q, &par) != 0) { lirerr(1473); return 1; }
  hip_liminfo_back();
  return 1;
}

/* ---- spur removal (genparm[MAX_NO_OF_SPURS] != 0, AFC on, second fft on).  Tracking and subtraction -- eliminate_spurs, spur.c:36-494 -- run
   on the device inside lrh_make_fft2.  The control plane stays Linrad's: spur_removal (wcw.c:204-247) and init_spur_elimination
   (spursub.c:181-343) decide where to look, using spursearch_spectrum, the power summed over 3 spur_speknum transforms that make_fft2
   keeps (fft2.c:673-699) -- hip_spur_after_fft2 keeps it from the new transform's power row; the two functions it calls to take a
   carrier on, store_new_spur (spursub.c:619: history of the seven bins from fftx) and spur_phase_lock (:1247: closes the loop on that
   history), are one device call, lrh_spur_acquire, on the resident spectra; remove_spur / swap_spurs (spur.c:596, spursub.c:755), with
   which it drops a weaker neighbour and keeps the list in order of frequency, become lrh_spur_permute.  The loop state comes back after
   every transform for the spur display and those decisions.  initial_remove_spur (spursub.c:346: the carrier also leaves the
   spur_speknum transforms the loop was closed on) is part of lrh_spur_acquire; the patch returns from Linrad's own at once. ---- */
static void hip_spur_state_back(void)
{
  lrh_spur *sp = hip_sp;
  int n = 0, i;
  if (lrh_spur_get(hip_rx, hip_spcap, sp, &n) != 0) { lirerr(1480); return; }
  for (i = 0; i < n && i < genparm[MAX_NO_OF_SPURS]; i++) {
    spur_location[i] = sp[i].spur_location; spur_flag[i] = sp[i].spur_flag; spur_freq[i] = sp[i].spur_freq;
    spur_d0pha[i] = sp[i].spur_d0pha; spur_d1pha[i] = sp[i].spur_d1pha; spur_d2pha[i] = sp[i].spur_d2pha;
    spur_ampl[i] = sp[i].spur_ampl; spur_noise[i] = sp[i].spur_noise; spur_avgd2[i] = sp[i].spur_avgd2;
  }
}
/* Linrad drops the LAST spur of its list by counting no_of_spurs down and nothing else (init_spur_elimination, spursub.c:331-335: remove_spur --
   the hooked function -- is called only when the dropped one is not the last).  The device's list is therefore cut back to Linrad's count
   before every acquisition and after every transform; otherwise it would keep subtracting the dropped carrier and the next acquisition
   would land one slot past the one Linrad reads. */
static void hip_spur_resync(void)
{
  int i;
  if (hip_dev_spurs <= no_of_spurs) return;
  for (i = 0; i < no_of_spurs; i++) hip_spsrc[i] = i;
  if (lrh_spur_permute(hip_rx, no_of_spurs, hip_spsrc) != 0) lirerr(1481);
  hip_dev_spurs = no_of_spurs;
}
int hip_store_new_spur(int pnt) { hip_spur_pnt = pnt; return 0; }      /* the history is taken on the device, by hip_spur_phase_lock */
int hip_spur_phase_lock(int nx)
{
  lrh_ptrs q;
  int locked = 0;
  if (hip_spur_pnt < 0) return 1;
  hip_spur_resync();
  memset(&q, 0, sizeof q);
  q.fft2_na = nx;                                                      /* ffts_na: the ring position behind the newest transform (wcw.c:288-289) */
  if (lrh_spur_acquire(hip_rx, &q, hip_spur_pnt, &locked) != 0) locked = 0;
  hip_spur_pnt = -1;
  if (!locked) return 1;
  hip_dev_spurs++;
  hip_spur_state_back();                                               /* spurno == no_of_spurs: the new spur's loop state for init_spur_elimination's ordering */
  return 0;
}
void hip_remove_spur(int ia)                                           /* remove_spur(ia): the last spur (number no_of_spurs, already counted down) takes slot ia */
{
  int *src = hip_spsrc, i;
  if (no_of_spurs >= hip_spcap) { lirerr(1481); return; }
  for (i = 0; i < no_of_spurs; i++) src[i] = i;
  if (ia >= 0 && ia < no_of_spurs) src[ia] = no_of_spurs;
  if (lrh_spur_permute(hip_rx, no_of_spurs, src) != 0) lirerr(1481);
  hip_dev_spurs = no_of_spurs;
  hip_spur_state_back();
}
void hip_swap_spurs(int ia, int ib)
{
  int *src = hip_spsrc, i, n = 0;
  if (lrh_spur_get(hip_rx, hip_spcap, hip_sp, &n) != 0 || ia >= n || ib >= n) return;
  for (i = 0; i < n; i++) src[i] = i;
  src[ia] = ib; src[ib] = ia;
  if (lrh_spur_permute(hip_rx, n, src) != 0) lirerr(1481);
  hip_spur_state_back();
}
/* after every transform: the loop state for the display; the search spectrum of make_fft2 (fft2.c:673-699) from the transform's power row;
   a spur the device reports unlocked for spur_speknum transforms is dropped when the search is automatic (spur.c:141-151: spur_relock has
   failed there; here the next pass of the search may take the carrier on again) */
static void hip_spur_after_fft2(int na)
{
  int i;
  hip_spur_resync();
  if (no_of_spurs > 0) {
    int n = 0, *src = hip_spsrc;
    hip_spur_state_back();
    if (genparm[AFC_ENABLE] == 2) {
      for (i = 0; i < no_of_spurs && i < hip_spcap; i++) if (spur_flag[i] < spur_speknum) src[n++] = i;
      if (n != no_of_spurs) { if (lrh_spur_permute(hip_rx, n, src) != 0) lirerr(1481); no_of_spurs = n; hip_dev_spurs = n; hip_spur_state_back(); }
    }
  }
  if (spursearch_spectrum == NULL) return;
  /* make_fft2's counter walk (fft2.c:675-699); the library sums the power rows and cleans the finished spectrum itself: what comes back,
     once per 3 spur_speknum + 2 transforms, is the search spectrum init_spur_elimination walks, its threshold, and the tail of
     spursearch_spectrum_cleanup that re-arms the walk (spursub.c:173-174) */
  if (spursearch_sum_counter > 3 * spur_speknum) {
    const int lo = spur_search_first_point;
    int done = 0, cnt = -1;
    spursearch_sum_counter = 0;
    if (lrh_spur_search_get(hip_rx, &spursearch_spectrum[lo], &spur_search_threshold, &done, &cnt) != 0 || cnt != 0) { lirerr(1482); return; }
    if (autospur_point >= spur_search_last_point - SPUR_WIDTH / 2) autospur_point = spur_search_first_point + SPUR_WIDTH / 2 + 1;
  } else spursearch_sum_counter++;
  (void)na;
}

static void hip_wf_collect(void)
{
  hip_wf_gen = (hip_wf_gen + 1) % (hip_lag + 1);
  for (int i = 0; i < hip_wf_n[hip_wf_gen]; i++) if (lrh_export_end(hip_rx, hip_wf_ticket[hip_wf_gen][i]) != 0) lirerr(1469);
  hip_wf_n[hip_wf_gen] = 0;
}
static void hip_wf_fetch(int lag, lrh_ring ring, void *dst, size_t off, size_t cnt)
{
  int t = 0;
  if (!lag || hip_wf_n[hip_wf_gen] >= 6) { lrh_export(hip_rx, ring, dst, off, cnt); return; }
  if (lrh_export_begin(hip_rx, ring, dst, off, cnt, &t) != 0) { lirerr(1469); return; }
  if (t) hip_wf_ticket[hip_wf_gen][hip_wf_n[hip_wf_gen]++] = t;
}

void hip_make_fft2(void)
{
  lrh_ptrs q;
  int old_ptr, nf = 1;
  memset(&q, 0, sizeof q);
  q.timf2_px = HIP_IN(timf2_px); q.fft2_na = fft2_na; q.fft2_pa = HIP_IN(fft2_pa); q.fft2_nb = fft2_nb; q.fft2_nm = fft2_nm;
  q.wg_waterf_sum_counter = wg_waterf_sum_counter; q.wg_waterf_ptr = wg_waterf_ptr; q.fft2_liminfo_cnt = fft2_liminfo_cnt;
  old_ptr = wg_waterf_ptr;
  if (HC == 1) {
    /* every transform the blanker has released samples for goes through in one call (second_fft comes back for each one it is owed:
       wcw.c:265-285 -- its test then fails after one pass), as far as the fft2 ring has room; not with spurs being tracked, whose loop
       state Linrad looks at after every transform */
    lrh_stage_wait_lag(hip_rx, LRH_STAGE_FFT2, hip_lag);
    hip_wf_collect();
    nf = 1 + (((timf2_pn2 - timf2_px + timf2_size) & timf2_mask) - 4 * fft2_size) / timf2_output_block;
    { const int room = max_fft2n - 1 - ((fft2_na - fft2_nx + max_fft2n) & fft2n_mask); if (nf > room) nf = room; }
    if (nf > max_fft2n / 2) nf = max_fft2n / 2;
    if (hip_spurs_on || nf < 1) nf = 1;
    if (hip_spurs_on) hip_spur_resync();                     /* a spur Linrad dropped since the last transform must not be subtracted from this one */
    q.timf2_px = HIP_IN(timf2_px); q.fft2_na = fft2_na;
    if (lrh_make_fft2(hip_rx, &q, nf) != 0) { lirerr(1469); return; } }
  else {
    /* two channels: each context transforms its own; the cross products TWOCHAN_POWER, their sums over the waterfall group and the
       polarisation-independent waterfall line (fft2.c:1622-1640, 1700-1815) need both channels' bins: all-gather of LRH_X_BINS */
    const lrh_ptrs at = q;
    size_t cnt[2] = { 0, 0 };
    for (int ch = 0; ch < 2; ch++) {
      q = at;
      if (lrh_make_fft2(hip_ctx[ch], &q, 1) != 0 || lrh_fft2_xy_begin(hip_ctx[ch], &at, 1, &cnt[ch]) != 0) { lirerr(1469); return; }
    }
    if (cnt[0] != cnt[1] || hip_xgather(LRH_X_BINS, cnt[0]) != 0) { lirerr(1469); return; }
    for (int ch = 0; ch < 2; ch++) if (lrh_fft2_xy_finish(hip_ctx[ch], &at, 1) != 0) { lirerr(1469); return; }
  }
  timf2_px = HIP_OUT(q.timf2_px); fft2_na = q.fft2_na; fft2_pa = HIP_OUT(q.fft2_pa); fft2_nb = q.fft2_nb; fft2_nm = q.fft2_nm;   /* fft2.c:1831-1845 */
  wg_waterf_sum_counter = q.wg_waterf_sum_counter; wg_waterf_ptr = q.wg_waterf_ptr; fft2_liminfo_cnt = q.fft2_liminfo_cnt;
  if (q.wg_waterf_ptr != old_ptr) {            /* waterfall lines completed (fft2.c:703-815; the pointer walks down, :813-815): the screen thread draws them */
    int pt;
    /* a call that found more than one transform owed is behind the device: the lines and the averaged spectrum (display data) are then
       collected by this thread's next call instead of being waited for; a call of one transform brings them back at once */
    const int lag = HC == 1 && nf > 1;
    for (pt = old_ptr; pt != q.wg_waterf_ptr; pt = pt - wg_xpixels < 0 ? pt - wg_xpixels + wg_waterf_size : pt - wg_xpixels)
      hip_wf_fetch(lag, LRH_RING_WG_WATERF, &wg_waterf[pt], (size_t)pt, (size_t)wg_xpixels);
    if (HC == 1) {                             /* the averaged spectrum of the newest line (read-backs of up to 512 KiB wait without the context's lock) */
      int at = 0;
      while (at < hip_n2) { const int k = hip_n2 - at > (1 << 17) ? (1 << 17) : hip_n2 - at; hip_wf_fetch(lag, LRH_RING_FFT2_POWERSUM, &fft2_powersum_float[at], (size_t)at, (size_t)k); at += k; }
    }
    else lrh_export(hip_rx, LRH_RING_FFT2_XYSUM, fft2_xysum, 0, (size_t)4 * hip_n2);
  }
  hip_afc_rows((q.fft2_na - nf) & fft2n_mask, nf);
  if (hip_spurs_on) hip_spur_after_fft2((q.fft2_na + fft2n_mask) & fft2n_mask);
  make_fft2_status = FFT2_COMPLETE;           /* second_fft loops until this (wcw.c:280-285) */
}

/* timf3 is where the device hands back to the CPU: fft3, mix2 and the demodulators carry on from the host ring (audio rate).
   The mixer's phase bookkeeping (set_mix1_phases, mix1.c:781-861) is host state of the library; the globals follow it because
   make_afc and the baseband graph read them. */
static void hip_mix1_back_n(int old_pa, int blocks)
{
  lrh_mix1_state m;
  int n = blocks * timf3_block, first = n;
  if (HC == 2) {                                           /* {ch0, ch1} per sample: each context's block, interleaved (timf3_block counts both) */
    const int per = timf3_block / 2, size1 = timf3_size / 2;
    int pa1 = old_pa / 2;
    float *t = hip_scratch((size_t)per);
    for (int ch = 0; ch < 2; ch++) {
      int done = 0, p1 = pa1;
      while (done < per) {
        int k = per - done; if (p1 + k > size1) k = size1 - p1;
        lrh_export(hip_ctx[ch], LRH_RING_TIMF3_FLOAT, t + done, (size_t)p1, (size_t)k);
        done += k; p1 = (p1 + k) & (size1 - 1);
      }
      for (int i = 0; i < per / 2; i++) {
        const int at = (old_pa + 4 * i) & timf3_mask;
        timf3_float[at + 2 * ch] = t[2 * i]; timf3_float[at + 2 * ch + 1] = t[2 * i + 1];
      }
    }
    if (lrh_get_mix1_state(hip_rx, &m) == 0 && m.mix1_selfreq >= 0) {
      mix1_point[0] = m.mix1_point; mix1_old_point[0] = m.mix1_old_point; mix1_phase[0] = m.mix1_phase;
      mix1_phase_step[0] = m.mix1_phase_step; mix1_phase_rot[0] = m.mix1_phase_rot; mix1_old_phase[0] = m.mix1_old_phase;
    }
    return;
  }
  if (old_pa + n > timf3_size) first = timf3_size - old_pa;
  lrh_export(hip_rx, LRH_RING_TIMF3_FLOAT, &timf3_float[old_pa], (size_t)old_pa, (size_t)first);
  if (first < n) lrh_export(hip_rx, LRH_RING_TIMF3_FLOAT, timf3_float, 0, (size_t)(n - first));
  if (lrh_get_mix1_state(hip_rx, &m) == 0 && m.mix1_selfreq >= 0) {
    mix1_point[0] = m.mix1_point; mix1_old_point[0] = m.mix1_old_point; mix1_phase[0] = m.mix1_phase;
    mix1_phase_step[0] = m.mix1_phase_step; mix1_phase_rot[0] = m.mix1_phase_rot; mix1_old_phase[0] = m.mix1_old_phase;
  }
}

static void hip_mix1_back(int old_pa) { hip_mix1_back_n(old_pa, 1); }

void hip_fft2_mix1_fixed(void)
{
  lrh_ptrs q;
  int old_pa, n = 1;
  memset(&q, 0, sizeof q);
  for (int ch = 0; ch < HC; ch++) lrh_set_mix1_selfreq(hip_ctx[ch], mix1_selfreq[0]);
  if (HC == 1) {
    /* every transform second_fft has finished goes through in one call, as far as timf3 has room (the narrowband thread comes back while
       `fft2_na != fft2_nx`, wcw.c:1737-1745: that test then fails after one pass).  The read-back below waits for the device, so this
       thread is never ahead of it: a call's size follows the load */
    const int room = ((timf3_px - timf3_pa + timf3_mask) & timf3_mask) / timf3_block;
    n = (fft2_na - fft2_nx + max_fft2n) & fft2n_mask;
    if (n > room) n = room;
    if (n > max_fft2n / 2) n = max_fft2n / 2;
    if (n < 1) n = 1;
  }
  q.fft2_nx = fft2_nx; q.timf3_pa = HIP_IN(timf3_pa); q.fft2_na = fft2_na;
  old_pa = timf3_pa;
  { const lrh_ptrs q0 = q;
    for (int ch = 0; ch < HC; ch++) { q = q0; if (lrh_fft2_mix1_fixed(hip_ctx[ch], &q, n) != 0) { lirerr(1470); return; } } }
  fft2_nx = q.fft2_nx; timf3_pa = HIP_OUT(q.timf3_pa);                          /* mix1.c:991-992 */
  hip_mix1_back_n(old_pa, n);
}

/* fft1_mix1_fixed (mix1.c:995-1042): the second fft is off -- Linrad's default in every rx mode (uivar.c:371-392, column 8) -- and
   the narrowband thread cuts the baseband straight out of the fft1 spectra (call site wcw.c:1712) */
void hip_fft1_mix1_fixed(void)
{
  lrh_ptrs q;
  int old_pa;
  memset(&q, 0, sizeof q);
  for (int ch = 0; ch < HC; ch++) lrh_set_mix1_selfreq(hip_ctx[ch], mix1_selfreq[0]);
  q.fft1_nx = fft1_nx; q.fft1_px = HIP_IN(fft1_px); q.fft1_nb = fft1_nb; q.timf3_pa = HIP_IN(timf3_pa);
  old_pa = timf3_pa;
  { const lrh_ptrs q0 = q;
    for (int ch = 0; ch < HC; ch++) { q = q0; if (lrh_fft1_mix1_fixed(hip_ctx[ch], &q, 1) != 0) { lirerr(1474); return; } } }
  fft1_nx = q.fft1_nx; fft1_px = HIP_OUT(q.fft1_px); timf3_pa = HIP_OUT(q.timf3_pa);             /* mix1.c:1039-1041 */
  hip_mix1_back(old_pa);
}

/* AFC variants (mix1.c:863-932, 1044-1097; call sites wcw.c:1700, 1737; AFC_ENABLE defaults to 1 for weak-signal CW): the
   per-transform frequency tables are Linrad's own globals, filled by make_afc on the host and kept by do_mix1_afc's bookkeeping,
   which the library restates (the caller's arrays are read and written in place) */
static void hip_afc_tables(lrh_afc *a)
{
  a->mix1_fq_mid = mix1_fq_mid; a->mix1_fq_slope = mix1_fq_slope; a->mix1_fq_curv = mix1_fq_curv; a->mix1_fq_start = mix1_fq_start;
  a->baseband_bw_hz = baseband_bw_hz;
}
void hip_fft2_mix1_afc(void)
{
  lrh_ptrs q;
  lrh_afc a;
  int old_pa;
  if (mix1_selfreq[0] < 0) { hip_fft2_mix1_fixed(); return; }                   /* nothing selected: mix1_clear either way (mix1.c:924-927) */
  memset(&q, 0, sizeof q);
  hip_afc_tables(&a);
  for (int ch = 0; ch < HC; ch++) lrh_set_mix1_selfreq(hip_ctx[ch], mix1_selfreq[0]);
  q.fft2_nx = fft2_nx; q.timf3_pa = HIP_IN(timf3_pa); q.fft2_na = fft2_na;
  old_pa = timf3_pa;
  { const lrh_ptrs q0 = q;                                 /* (the second context finds the tables as the first has left them: the same numbers once more) */
    for (int ch = 0; ch < HC; ch++) { q = q0; if (lrh_fft2_mix1_afc(hip_ctx[ch], &q, 1, &a) != 0) { lirerr(1475); return; } } }
  fft2_nx = q.fft2_nx; timf3_pa = HIP_OUT(q.timf3_pa);                          /* mix1.c:930-931 */
  hip_mix1_back(old_pa);
}
void hip_fft1_mix1_afc(void)
{
  lrh_ptrs q;
  lrh_afc a;
  int old_pa;
  if (mix1_selfreq[0] < 0) { hip_fft1_mix1_fixed(); return; }
  memset(&q, 0, sizeof q);
  hip_afc_tables(&a);
  for (int ch = 0; ch < HC; ch++) lrh_set_mix1_selfreq(hip_ctx[ch], mix1_selfreq[0]);
  q.fft1_nx = fft1_nx; q.fft1_px = HIP_IN(fft1_px); q.fft1_nb = fft1_nb; q.timf3_pa = HIP_IN(timf3_pa);
  old_pa = timf3_pa;
  { const lrh_ptrs q0 = q;
    for (int ch = 0; ch < HC; ch++) { q = q0; if (lrh_fft1_mix1_afc(hip_ctx[ch], &q, 1, &a) != 0) { lirerr(1476); return; } } }
  fft1_nx = q.fft1_nx; fft1_px = HIP_OUT(q.fft1_px); timf3_pa = HIP_OUT(q.timf3_pa);             /* mix1.c:1094-1096 */
  hip_mix1_back(old_pa);
}

/* compute_timf2_powersum (wcw.c:80-138; S/N meter, mg.scale_type == MG_SCALE_STON): the block powers are formed on the device
   from the resident timf2 ring and the new ones come back for the meter graph */
void hip_compute_timf2_powersum(void)
{
  lrh_ptrs q;
  int old_pa, n;
  memset(&q, 0, sizeof q);
  q.timf2_pn2 = HIP_IN(timf2_pn2); q.timf2_pb = HIP_IN(timf2_pb); q.timf2_blockpower_pa = timf2_blockpower_pa;
  old_pa = timf2_blockpower_pa;
  { const lrh_ptrs q0 = q;
    for (int ch = 0; ch < HC; ch++) { q = q0; if (lrh_compute_timf2_powersum(hip_ctx[ch], &q) != 0) { lirerr(1477); return; } } }
  timf2_pb = HIP_OUT(q.timf2_pb); timf2_blockpower_pa = q.timf2_blockpower_pa;
  if (HC == 2) {                                           /* {ch0, ch1} per block (wcw.c:112-134) */
    n = (q.timf2_blockpower_pa - old_pa) & timf2_blockpower_mask;
    while (n > 0) {
      float t[2];
      for (int ch = 0; ch < 2; ch++) lrh_export(hip_ctx[ch], LRH_RING_TIMF2_BLOCKPOWER, &t[ch], (size_t)old_pa, 1);
      timf2_blockpower[2 * old_pa] = t[0]; timf2_blockpower[2 * old_pa + 1] = t[1];
      old_pa = (old_pa + 1) & timf2_blockpower_mask; n--;
    }
    return;
  }
  n = (q.timf2_blockpower_pa - old_pa) & timf2_blockpower_mask;
  while (n > 0) {
    int k = n;
    if (old_pa + k > timf2_blockpower_mask + 1) k = timf2_blockpower_mask + 1 - old_pa;
    lrh_export(hip_rx, LRH_RING_TIMF2_BLOCKPOWER, &timf2_blockpower[old_pa], (size_t)old_pa, (size_t)k);
    old_pa = (old_pa + k) & timf2_blockpower_mask; n -= k;
  }
}

/* ---- network output of device-resident stages.  Linrad's senders read the host rings (the dispatcher's memcpy of a retired batch,
   wcw.c:1024-1043; the network thread's walks from timf2_pt / fft2_pt, rxin.c:919-1060); with version 21 those rings stay empty, so each
   hook fetches exactly the span the sender is about to read into its place in the host ring and the reference's packet code runs on
   unchanged.  (lrh_export_timf2_net would halve the PCIe bytes of the timf2 payload by combining weak and strong on the device; that
   needs the sender's loop itself replaced.) ---- */
static void hip_ring_span(lrh_ring ring, float *host, int pt, int count, int size)
{
  while (count > 0) {
    int k = count;
    if (pt + k > size) k = size - pt;
    if (lrh_export(hip_rx, ring, &host[pt], (size_t)pt, (size_t)k) != 0) { lirerr(1478); return; }
    pt = (pt + k) & (size - 1); count -= k;
  }
}
/* The same for two channels: a frame of the host ring holds P complex values of channel 0 and of channel 1, value by value ({ch0, ch1} per
   bin of fft1_float / fft2_float: P = 1; {weak ch0, weak ch1, strong ch0, strong ch1} per sample of timf2_float: P = 2), a context's ring the P
   values of its own channel.  pt, count, size in floats of the HOST ring. */
static void hip_ring_span2(lrh_ring ring, float *host, int pt, int count, int size, int P)
{
  const int hf = 4 * P, cf = 2 * P;
  int f = pt / hf, nfr = (pt % hf + count + hf - 1) / hf;
  const int ring_frames = size / hf;
  while (nfr > 0) {
    int k = nfr;
    if (f + k > ring_frames) k = ring_frames - f;
    float *t = hip_scratch((size_t)k * cf);
    for (int ch = 0; ch < 2; ch++) {
      if (lrh_export(hip_ctx[ch], ring, t, (size_t)f * cf, (size_t)k * cf) != 0) { lirerr(1478); return; }
      for (int i = 0; i < k; i++)
        for (int p = 0; p < P; p++) {
          float *h = &host[(size_t)(f + i) * hf + 2 * (2 * p + ch)];
          h[0] = t[(size_t)i * cf + 2 * p]; h[1] = t[(size_t)i * cf + 2 * p + 1];
        }
    }
    f = (f + k) % ring_frames; nfr -= k;
  }
}
/* NET_RXOUT_FFT1: the batch of transforms as fft1_b leaves them -- before fft1_c's filter correction (network.c:383-388) -- for the blocks
   that start at timf1p_ref, into fft1_float at fft1_pa where the memcpy expects them */
void hip_net_fft1(int timf1p_ref, int pa)
{
  const int nb = gpu_fft1_batch_size > 0 ? gpu_fft1_batch_size : 1;
  if (!hip_rx) return;
  if (HC == 2) {                                            /* {ch0, ch1} per bin (fft1.c:2041) */
    float *t = hip_scratch((size_t)nb * 2 * hip_n1);
    for (int ch = 0; ch < 2; ch++) {
      if (lrh_export_fft1_net(hip_ctx[ch], t, hip_real2 ? timf1p_ref / 2 : timf1p_ref, nb) != 0) { lirerr(1479); return; }
      for (size_t i = 0; i < (size_t)nb * hip_n1; i++) { fft1_float[pa + 4 * i + 2 * ch] = t[2 * i]; fft1_float[pa + 4 * i + 2 * ch + 1] = t[2 * i + 1]; }
    }
    return;
  }
  if (lrh_export_fft1_net(hip_rx, &fft1_float[pa], timf1p_ref, nb) != 0) lirerr(1479);
}
/* NET_RXOUT_TIMF2, float format: `mm` floats of {weak, strong} samples from timf2_pt on (rxin.c:944-966 consumes 2*twice_rxchan floats per
   twice_rxchan floats it sends, mm*2 bytes in all) */
void hip_net_timf2(int pt, int mm)
{
  if (hip_rx && mm > 0 && HC == 2) hip_ring_span2(LRH_RING_TIMF2_FLOAT, timf2_float, pt, mm, timf2_size, 2);
  else
  if (hip_rx && mm > 0) hip_ring_span(LRH_RING_TIMF2_FLOAT, timf2_float, pt & ~3, (mm + 3) & ~3, timf2_size);
}
/* NET_RXOUT_FFT2: `count` floats of fft2_float from fft2_pt on (rxin.c:1026-1035) */
void hip_net_fft2(int pt, int count)
{
  if (hip_rx && count > 0 && HC == 2) hip_ring_span2(LRH_RING_FFT2_FLOAT, fft2_float, pt, count, max_fft2n * 4 * hip_n2, 1);
  else
  if (hip_rx && count > 0) hip_ring_span(LRH_RING_FFT2_FLOAT, fft2_float, pt, count, max_fft2n * 2 * hip_n2);
}
