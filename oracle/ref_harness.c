/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Head-less driver for the *compiled reference* (fventuri/linrad C sources under
 * /root/reference, compiled where they lie by oracle/Makefile into oracle/_ref/).
 * It owns the rings and the ~80 globals the reference hot path reads, feeds
 * recorded/synthetic int16 IQ through the reference's own functions in the
 * order the single-CPU branch of wideband_dsp uses (wcw.c:1036-1118):
 *
 *   fft1_b -> fft1_c -> make_timf2 -> first_noise_blanker -> make_fft2* -> fft2_mix1_fixed
 *
 * and dumps every stage ring + scalar state into one container file that
 * tests/golden/make_golden.py turns into the committed fixtures.
 *
 * Nothing here is reference source: the reference is linked, not copied.
 * Stubs below stand in for GUI/thread entry points of the same program
 * (lirerr, lir_sched_yield, awake_screen, ...), as SURVEY.md Appendix C found.
 *
 * usage: ref_harness key=value ... (see parse section); writes out=<file>.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include <time.h>
#include <pthread.h>
#include <unistd.h>

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
#include "graphcal.h"
#ifdef SHIM_HARNESS
/* oracle/_ref/shim_harness (oracle/build_shim_harness.sh): the same driver over the PATCHED reference objects, fft1 version 21;
   integration/hipshim.c is linked over the oracle's C ABI (oracle/shim_alias.h), so the stage functions called below are the
   reference's own entry points handing over to the glue.  The dump then holds what Linrad sees on the host (sums, lines,
   pointers, scalars, timf3) plus, fetched at the end, the device-resident rings. */
#include "linrad_hip.h"
#include "hipshim.h"
#endif

/* ---- stand-ins for GUI / OS entry points of the reference program ---- */
int harness_err = 0;
void lirerr(int e) { fprintf(stderr, "lirerr(%d)\n", e); harness_err = e; }
/* the reference's block allocator (modesub.c:1886-1960) as plain callocs: init_blanker (buf.c:1771) registers its tables through it */
void init_memalloc(MEM_INF *mm, size_t max) { (void)mm; (void)max; }
void mem(int num, void *pointer, size_t size, int scratch_size) { (void)num; *(void **)pointer = (char *)calloc(1, size + scratch_size + 64) + ((scratch_size + 15) & ~15); }
size_t memalloc(size_t **hh, char *s) { (void)s; *hh = (size_t *)calloc(1, 64); return 64; }
void memcheck(int callno, MEM_INF *mm, size_t **hh) { (void)callno; (void)mm; (void)hh; }
/* fft3_mix2 runs on into the demodulators; its filter / decimate / polarisation part ends at the thread-command check of
   mix2.c:749.  The back transform in front of that check yields (fft0.c:495, yieldflag_ndsp_mix2), and the yield is this
   OS stub: armed, it withdraws the thread's command the way the GUI thread would, so the function returns there. */
static int mix2_trip = 0;
void lir_sched_yield(void) { if (mix2_trip) thread_command_flag[THREAD_MIX2] = THRFLAG_IDLE; }
void awake_screen(void) {}
void lir_pixwrite(int x, int y, char *s) { (void)x; (void)y; (void)s; }
void lir_text(int x, int y, char *s) { (void)x; (void)y; (void)s; }
void settextcolor(unsigned char c) { (void)c; }
void lir_mutex_lock(int n) { (void)n; }
void lir_mutex_unlock(int n) { (void)n; }
void lir_sleep(int us) { (void)us; }

/* prototypes of reference functions not in the headers we include */
void fft1_b(int timf1p_ref, float *out, float *tmp, int gpu_handle_number);
void prepare_mixer(MIXER_VARIABLES *m, int nn);
void fft1_c(void);
void make_timf2(void);
void first_noise_blanker(void);
void make_fft2(void);
void fft2_mix1_fixed(void);
void fft1_mix1_fixed(void);
void fft2_mix1_afc(void);
void fft1_mix1_afc(void);
void compute_timf2_powersum(void);
void make_fft3_all(void);
void fft3_mix2(void);
void clear_fft1_filtercorr(void);
void make_permute(int mo, int nz, int sz, unsigned short int *perm);
void make_bigpermute(int mo, int nz, int sz, unsigned int *perm);
void make_sincos(int mo, int sz, COSIN_TABLE *tab);
void init_fft(int mo, int nz, int sz, COSIN_TABLE *tab, unsigned short int *perm);
void set_fft1_endpoints(void);
void fft1_update_liminfo(void);
int store_new_spur(int pnt);
void initial_remove_spur(void);
int spur_phase_lock(int nx);
void init_spur_spectra(void);
void eliminate_spurs(void);
extern float spur_search_threshold;

/* ---- container writer ---- */
static FILE *fo;
static void put(const char *name, const char *dtype, const void *p, size_t count, size_t esz)
{
  char nm[32]; char dt[4];
  memset(nm, 0, 32); strncpy(nm, name, 31);
  memset(dt, 0, 4); strncpy(dt, dtype, 3);
  uint64_t c = count;
  fwrite(nm, 1, 32, fo); fwrite(dt, 1, 4, fo); fwrite(&c, 8, 1, fo);
  fwrite(p, esz, count, fo);
}
#define PUTF(n, p, c) put(n, "f4", p, c, 4)
#define PUTI(n, p, c) put(n, "i4", p, c, 4)

static void *zalloc(size_t n) { void *p = calloc(n + 64, 1); if (!p) { fprintf(stderr, "oom\n"); exit(2);} return p; }

static const char *arg(int argc, char **argv, const char *k, const char *def)
{
  size_t l = strlen(k);
  for (int i = 1; i < argc; i++) if (!strncmp(argv[i], k, l) && argv[i][l] == '=') return argv[i] + l + 1;
  return def;
}
#define AI(k, d) atoi(arg(argc, argv, k, #d))
#define AF(k, d) atof(arg(argc, argv, k, #d))

/* ---- timing=1 threads=T: the reference's own thread topology (wcw.c:604-648) around its unchanged stage functions.
   THREAD_WIDEBAND_DSP dispatches every block to one of up to six THREAD_FFT1Bk workers and retires them in order (wcw.c:969-1047);
   THREAD_TIMF2 runs fft1_c / make_timf2 for every finished transform and then the blanker (wcw.c:401-441); THREAD_SECOND_FFT runs
   make_fft2 while a transform's worth of samples is released (wcw.c:250-304); the narrowband thread runs fft2_mix1_fixed and
   what follows (wcw.c:1240-1405).  Hand-offs: binary auto-reset condition events like lxsys.c:415-447. ---- */
/* the oracle (oracle/_ref/shim_harness: hipshim.c over lro_*, SHIM_ALIAS_H) is not thread safe -- the HIP library is, and that is what
   shim_harness_hip puts under test -- so over the oracle the free-running stage threads take one lock around each stage call */
#ifdef SHIM_ALIAS_H
static pthread_mutex_t stage_big_lock = PTHREAD_MUTEX_INITIALIZER;
#define SLOCK(call) do { pthread_mutex_lock(&stage_big_lock); call; pthread_mutex_unlock(&stage_big_lock); } while (0)
#else
#define SLOCK(call) do { call; } while (0)
#endif
/* timing=1 over the glue (shim_harness_hip, bench.py "glue"): samples enter through the producer hook block by block, and every stage call's
   wall time is summed per stage (where the drop-in's time goes: the stage threads are Linrad's, the calls are the patched entry points) */
enum { SG_INGEST, SG_FFT1B, SG_FFT1C, SG_TIMF2, SG_BLANK, SG_FFT2, SG_MIX1, SG_FFT3, SG_WAIT_DISP, SG_WAIT_IN, SG_N };
static const char *sg_name[SG_N] = { "finish_rx_read_hook", "fft1_b", "fft1_c", "make_timf2", "first_noise_blanker", "make_fft2", "fft2_mix1_fixed", "fft3_mix2_host", "dispatcher_wait_room", "dispatcher_wait_input" };
static volatile double sg_sec[SG_N]; static volatile long sg_calls[SG_N]; static int sg_on = 0, glue_ingest = 0;
static pthread_mutex_t sg_m = PTHREAD_MUTEX_INITIALIZER;
static inline double sg_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define STAGE(k, call) do { if (sg_on) { const double t0_ = sg_now(); SLOCK(call); const double d_ = sg_now() - t0_; pthread_mutex_lock(&sg_m); sg_sec[k] += d_; sg_calls[k]++; pthread_mutex_unlock(&sg_m); } else SLOCK(call); } while (0)
enum { TEV_TIMF2, TEV_FFT2, TEV_READY, TEV_SPACE, TEV_DONE, TEV_DO, TNEV = TEV_DO + 6 };
static pthread_mutex_t tev_m[TNEV]; static pthread_cond_t tev_c[TNEV]; static volatile int tev_f[TNEV];
static void tev_set(int n) { pthread_mutex_lock(&tev_m[n]); tev_f[n] = 1; pthread_cond_signal(&tev_c[n]); pthread_mutex_unlock(&tev_m[n]); }
static void tev_await(int n) { pthread_mutex_lock(&tev_m[n]); while (!tev_f[n]) pthread_cond_wait(&tev_c[n], &tev_m[n]); tev_f[n] = 0; pthread_mutex_unlock(&tev_m[n]); }
static struct { int nblk, workers, C, n3, mix2on, N2, fq_ok; volatile int wide_done, timf2_done, fft2_done; volatile int nfft2; float *tmp[6];
                struct { volatile int inptr, out, busy; } job[6]; } TH;
static void *th_fft1b(void *arg)
{
  const int k = (int)(long)arg;
  for (;;) {
    tev_await(TEV_DO + k);
    if (TH.job[k].busy < 0) return NULL;
    STAGE(SG_FFT1B, fft1_b(TH.job[k].inptr, &fft1_float[TH.job[k].out], TH.tmp[k], k));
    TH.job[k].busy = 2;
    tev_set(TEV_DONE);
  }
}
static void *th_timf2(void *arg)
{
  for (;;) {
    tev_await(TEV_TIMF2);
    while (fft1_na != fft1_nb) {
      while (fft1_na != fft1_nb) { STAGE(SG_FFT1C, fft1_c()); STAGE(SG_TIMF2, make_timf2()); }
      STAGE(SG_BLANK, first_noise_blanker());
      if (((timf2_pn2 - timf2_px + timf2_size) & timf2_mask) >= 4 * TH.C * fft2_size) tev_set(TEV_FFT2);
      tev_set(TEV_SPACE);
    }
    if (TH.wide_done && fft1_na == fft1_nb) break;
  }
  TH.timf2_done = 1; tev_set(TEV_FFT2);
  return NULL;
}
static void *th_fft2(void *arg)
{
  for (;;) {
    tev_await(TEV_FFT2);
    while (((timf2_pn2 - timf2_px + timf2_size) & timf2_mask) >= 4 * TH.C * fft2_size && ((fft2_na - fft2_nx + max_fft2n) & fft2n_mask) < max_fft2n - 1) {
      make_fft2_status = FFT2_NOT_ACTIVE;
      while (make_fft2_status != FFT2_COMPLETE) STAGE(SG_FFT2, make_fft2());
      tev_set(TEV_READY); tev_set(TEV_SPACE);
    }
    if (TH.timf2_done && ((timf2_pn2 - timf2_px + timf2_size) & timf2_mask) < 4 * TH.C * fft2_size) break;
    if (((fft2_na - fft2_nx + max_fft2n) & fft2n_mask) >= max_fft2n - 1) { tev_set(TEV_READY); usleep(50); tev_set(TEV_FFT2); }
  }
  TH.fft2_done = 1; tev_set(TEV_READY);
  return NULL;
}
static void *th_narrow(void *arg)
{
  for (;;) {
    tev_await(TEV_READY);
    while (fft2_nx != fft2_na) {
      { const int nx0 = fft2_nx;
        if (TH.fq_ok) STAGE(SG_MIX1, fft2_mix1_fixed()); else fft2_nx = (fft2_nx + 1) & fft2n_mask;
        TH.nfft2 += (fft2_nx - nx0 + max_fft2n) & fft2n_mask; }           /* (the glue takes every finished transform in one call) */
      if (TH.n3 <= 0) timf3_px = timf3_pa;                              /* no fft3 configured: the consumer of timf3 keeps up */
      if (TH.n3 > 0) while (((timf3_pa - timf3_px + timf3_size) & timf3_mask) >= 2 * TH.C * fft3_size &&
                            ((fft3_pa - fft3_px + fft3_totsiz) & fft3_mask) < fft3_totsiz - 2 * fft3_block) {
        const double t3_ = sg_on ? sg_now() : 0;
        make_fft3_all();
        if (TH.mix2on) { thread_command_flag[THREAD_MIX2] = THRFLAG_ACTIVE; mix2_trip = 1; fft3_mix2(); mix2_trip = 0; baseb_pa = (baseb_pa + mix2.new_points) & baseband_mask; timf3_py = (timf3_py + 2 * TH.C * fft3_new_points) & timf3_mask; }
        fft3_px = (fft3_px + fft3_block) & fft3_mask;
        if (sg_on) { sg_sec[SG_FFT3] += sg_now() - t3_; sg_calls[SG_FFT3]++; }     /* this thread only */
      }
      tev_set(TEV_FFT2); tev_set(TEV_SPACE);
    }
    if (TH.fft2_done && fft2_nx == fft2_na) break;
  }
  return NULL;
}
#ifdef SHIM_HARNESS
/* the rx input thread (rxin.c: the sound-card / SDR reader): completes blocks in timf1 and hands each to finish_rx_read, whose patched tail
   (rxin.c:1423) passes it to the device before EVENT_TIMF1 wakes the wideband thread -- here: one fft1_b dispatch's worth at a time, never more
   than half the ring ahead of the dispatcher */
static struct { pthread_mutex_t m; pthread_cond_t c; volatile long in_done, disp_done; long total, ring_blocks; int pa; } IN;
static void *th_input(void *arg)
{
  for (long b = 0; b < IN.total; b++) {
    pthread_mutex_lock(&IN.m);
    while (IN.in_done - IN.disp_done >= IN.ring_blocks / 2) pthread_cond_wait(&IN.c, &IN.m);
    pthread_mutex_unlock(&IN.m);
    STAGE(SG_INGEST, hip_timf1_new(IN.pa, timf1_blockbytes));
    IN.pa = (IN.pa + timf1_blockbytes) & timf1_bytemask;
    pthread_mutex_lock(&IN.m); IN.in_done++; pthread_cond_broadcast(&IN.c); pthread_mutex_unlock(&IN.m);
  }
  return NULL;
}
#endif
static void th_retire(int k)
{
  while (TH.job[k].busy != 2) tev_await(TEV_DONE);
  TH.job[k].busy = 0;
  fft1_pa = (fft1_pa + fft1_mulblock) & fft1_mask;
  fft1_na = fft1_pa / fft1_block;
  if (fft1_nm != fft1n_mask) fft1_nm++;
  tev_set(TEV_TIMF2);
}
static void run_reference_threads(void)        /* the dispatcher = this thread (THREAD_WIDEBAND_DSP) */
{
  pthread_t th[3 + 6];
  int n = 0, next = 0, oldest = 0, inflight = 0, out = fft1_pa;
  for (int i = 0; i < TNEV; i++) { pthread_mutex_init(&tev_m[i], NULL); pthread_cond_init(&tev_c[i], NULL); tev_f[i] = 0; }
  TH.wide_done = TH.timf2_done = TH.fft2_done = 0;                 /* (the function may run again on the state the previous run left: warm-up, then the timed run) */
  for (int k = 0; k < 6; k++) TH.job[k].busy = 0;
#ifdef SHIM_HARNESS
  pthread_t thin; 
  if (glue_ingest) {
    pthread_mutex_init(&IN.m, NULL); pthread_cond_init(&IN.c, NULL);
    IN.in_done = IN.disp_done = 0; IN.total = TH.nblk; IN.ring_blocks = timf1_bytes / timf1_blockbytes; IN.pa = timf1p_px;
    pthread_create(&thin, NULL, th_input, NULL);
  }
#endif
  pthread_create(&th[n++], NULL, th_narrow, NULL); pthread_create(&th[n++], NULL, th_fft2, NULL); pthread_create(&th[n++], NULL, th_timf2, NULL);
  for (int k = 0; k < TH.workers; k++) pthread_create(&th[n++], NULL, th_fft1b, (void *)(long)k);
  for (int b = 0; b < TH.nblk; b++) {
    /* room in the rings (Linrad counts overruns instead, wcw.c:770-785) */
    for (;;) {
      const int no_room = ((fft1_na - fft1_nx + max_fft1n) & fft1n_mask) + (inflight + 1) * fft1_muln >= max_fft1n - 1 ||
                          ((timf2_pa - timf2_px + timf2_size) & timf2_mask) > timf2_size / 2;
      if (inflight == TH.workers || (no_room && inflight > 0)) { th_retire(oldest); oldest = (oldest + 1) % TH.workers; inflight--; continue; }   /* a full ring empties only through retired transforms */
      if (!no_room) break;
      { const double tw_ = sg_on ? sg_now() : 0; tev_await(TEV_SPACE); if (sg_on) { pthread_mutex_lock(&sg_m); sg_sec[SG_WAIT_DISP] += sg_now() - tw_; sg_calls[SG_WAIT_DISP]++; pthread_mutex_unlock(&sg_m); } }
    }
#ifdef SHIM_HARNESS
    /* finish_rx_read's hand-over (rxin.c:1423, patched): the block the input thread has just completed at timf1p_pa -- the new samples of this
       dispatch, [timf1p_px, timf1p_px + timf1_blockbytes) -- goes to the device before the wideband thread is told (EVENT_TIMF1) */
    if (glue_ingest) {                                              /* EVENT_TIMF1: a block has come in */
      const double tw_ = sg_on ? sg_now() : 0;
      pthread_mutex_lock(&IN.m);
      while (IN.in_done <= IN.disp_done) pthread_cond_wait(&IN.c, &IN.m);
      IN.disp_done++; pthread_cond_broadcast(&IN.c);
      pthread_mutex_unlock(&IN.m);
      if (sg_on) { pthread_mutex_lock(&sg_m); sg_sec[SG_WAIT_IN] += sg_now() - tw_; sg_calls[SG_WAIT_IN]++; pthread_mutex_unlock(&sg_m); }
    }
#endif
    TH.job[next].inptr = timf1p_px; TH.job[next].out = out; TH.job[next].busy = 1;
    out = (out + fft1_mulblock) & fft1_mask;
    timf1p_px = (timf1p_px + timf1_blockbytes) & timf1_bytemask;
    tev_set(TEV_DO + next);
    next = (next + 1) % TH.workers; inflight++;
  }
  while (inflight > 0) { th_retire(oldest); oldest = (oldest + 1) % TH.workers; inflight--; }
  for (int k = 0; k < TH.workers; k++) { TH.job[k].busy = -1; tev_set(TEV_DO + k); }
  TH.wide_done = 1; tev_set(TEV_TIMF2);
  for (int i = 0; i < n; i++) pthread_join(th[i], NULL);
#ifdef SHIM_HARNESS
  if (glue_ingest) pthread_join(thin, NULL);
#endif
}

/* ---- shim_threads=1 (SHIM_HARNESS): every stage function is called from the thread Linrad calls it from with more than one CPU --
   fft1_b from a worker (THREAD_FFT1B1, wcw.c:476-500), fft1_c / make_timf2 / first_noise_blanker / compute_timf2_powersum from
   THREAD_TIMF2 (wcw.c:401-441), make_fft2 from THREAD_SECOND_FFT (wcw.c:250-304), fft2_mix1_* / fft1_mix1_* and what follows from the
   narrowband thread (wcw.c:1240-1405), the limiter calls from THREAD_WIDEBAND_DSP (this thread, wcw.c:1124-1133) -- but handed over
   in LOCK STEP: the caller waits for the stage to return, so the order of calls is the single-CPU one and every product can be held
   against the unpatched goldens, while the library sees its context used from five host threads.  shim_threads=2 lets the same
   threads run free (run_reference_threads above). ---- */
enum { ST_WORKER, ST_TIMF2, ST_FFT2, ST_NARROW, ST_COUNT };
static struct stage_thread { pthread_t th; pthread_mutex_t m; pthread_cond_t c; void (*fn)(void); int state; } ST[ST_COUNT];
static int stage_threads_on = 0;
static void *stage_main(void *a)
{
  struct stage_thread *t = a;
  for (;;) {
    pthread_mutex_lock(&t->m);
    while (t->state != 1 && t->state != 3) pthread_cond_wait(&t->c, &t->m);
    if (t->state == 3) { pthread_mutex_unlock(&t->m); return NULL; }
    pthread_mutex_unlock(&t->m);
    t->fn();
    pthread_mutex_lock(&t->m); t->state = 2; pthread_cond_broadcast(&t->c); pthread_mutex_unlock(&t->m);
  }
}
static void stage_start(void)
{
  for (int k = 0; k < ST_COUNT; k++) { pthread_mutex_init(&ST[k].m, NULL); pthread_cond_init(&ST[k].c, NULL); ST[k].state = 0; pthread_create(&ST[k].th, NULL, stage_main, &ST[k]); }
  stage_threads_on = 1;
}
static void stage_stop(void)
{
  if (!stage_threads_on) return;
  for (int k = 0; k < ST_COUNT; k++) { pthread_mutex_lock(&ST[k].m); ST[k].state = 3; pthread_cond_broadcast(&ST[k].c); pthread_mutex_unlock(&ST[k].m); pthread_join(ST[k].th, NULL); }
  stage_threads_on = 0;
}
static void on_stage(int k, void (*fn)(void))
{
  if (!stage_threads_on) { fn(); return; }
  struct stage_thread *t = &ST[k];
  pthread_mutex_lock(&t->m); t->fn = fn; t->state = 1; pthread_cond_broadcast(&t->c);
  while (t->state != 2) pthread_cond_wait(&t->c, &t->m);
  t->state = 0; pthread_mutex_unlock(&t->m);
}
static struct { int inptr; float *out, *tmp; int handle; } fft1b_job;
static void fft1b_stage(void) { fft1_b(fft1b_job.inptr, fft1b_job.out, fft1b_job.tmp, fft1b_job.handle); }
#define ON(k, f) on_stage(k, f)

/* per-block scalar trace */
#define TR_COLS 16
int main(int argc, char **argv)
{
  int n1 = AI("n1", 10), sinpow1 = AI("sinpow1", 2);
  int n2 = AI("n2", 12), sinpow2 = AI("sinpow2", 2);
  int mixred = AI("mixred", 6);
  int att_n = AI("att_n", 6), gain = AI("gain", 27);
  int avg1 = AI("avg1num", 5), avg2 = AI("avg2num", 4);
  int nblk = AI("nblk", 32);
  int ring_log2 = AI("timf1_log2", 0);          /* 0: auto */
  int timf2pow_log2 = AI("timf2pow_log2", 0);   /* 0: auto = 8*N2 */
  int maxfft1n = AI("max_fft1n", 8), maxfft2n = AI("max_fft2n", 4);
  int sumsq_blocks = AI("sumsq_blocks", 8);
  int stupid = AI("stupid", 1);
  int bln_interval = AI("bln_interval", 4), bln_avgnum = AI("bln_avgnum", 32);
  int bln_minpts = AI("bln_minpts", -1);
  int pulsewidth = AI("pulsewidth", 0), fitrange = AI("blnfit_range", 48);
  int noise_floor0 = AI("noise_floor", 200);
  double fq = AF("fq", -1.0);                    /* selected frequency in fft2 bins; <0: none */
  int wf_avg = AI("wf_avgnum", 2);               /* fft2 spectra per waterfall line */
  int wf_first = AI("wf_first", 0), wf_pix = AI("wf_pixels", 0);
  int wf_mode = AI("wf_mode", 1);                /* 1: 1:1, k>1: k points/pixel (max), k<0: -k pixels/point (interp) */
  int second = AI("second_fft", 1);              /* genparm[SECOND_FFT_ENABLE] */
  int bp_block = AI("blockpower_block", 0), bp_size = AI("blockpower_size", 1024);
  int n3 = AI("fft3_n", 0), sinpow3 = AI("fft3_sinpow", 2), nm2 = AI("mix2_n", 0), maxfft3n = AI("max_fft3n", 8);
  int lim_every = AI("lim_every", 0);            /* liminfo record stride in blocks (0: single record) */
  int dword = AI("dword", 0);                    /* ui.rx_input_mode & DWORD_INPUT: the input file holds int32 I,Q */
  int sshift = AI("sample_shift", 0);            /* ui.sample_shift */
  int afc = AI("afc", 0);                        /* 1: fft2_mix1_afc / fft1_mix1_afc with a synthetic per-transform frequency */
  double afc_bw = AF("afc_bw", 20.0);            /* baseband_bw_hz */
  int direction = AI("direction", 1);            /* fft1_direction (fg.passband_direction): -1 mirrors the spectrum */
  int timing = AI("timing", 0);                  /* 1: print the wall time of the block loop (bench.py cpu_baseline), skip the ring dumps */
  int C = AI("channels", 1);                     /* ui.rx_rf_channels; 2: frames {I0,Q0,I1,Q1}, run stops after make_timf2 */
  int realin = AI("real", 0);                    /* 1: real samples (ui.rx_input_mode without IQ_DATA): fft1 version 2, the split-radix
                                                    real transform fft1_reherm_dit_one (fft1_re.c:32), 2*N1 reals per transform */
  int corr = AI("corr", 0);                      /* channels=2 only: genparm[FFT1_CORRELATION_SPECTRUM] = 1 -- fft1_c also forms the channels' cross spectrum
                                                    fft1_corrsum (fft1.c:4146-4150, 4189-4193) and update_fft1_slowsum its averages fft1_slowcorr / _tot (:4584-4603) */
  int chain2 = AI("chain2", 0);                  /* channels=2 only: run on through the two-channel first_noise_blanker, make_fft2
                                                    (fft2_xypower / fft2_xysum, polarisation-independent waterfall) and fft2_mix1_fixed */
  int sellim = AI("sellim", 0);                  /* 1: the selective limiter runs (fft1_update_liminfo, sellim.c:738) whenever fft1_c completes an
                                                    averaging period, in the single-CPU order of wcw.c:1124-1128; make_timf2 routes with its table */
  int sellim2 = AI("sellim2", 0);                /* 1 (with sellim=1): fft2_update_liminfo (sellim.c:159, hg.sellim_par1 = 2) whenever make_fft2 completes a waterfall
                                                    line, after fft1_update_liminfo like wcw.c:1124-1133 */
  int spur = AI("spur", 0);                      /* 1: a spur at fft2 bin spur_pnt (first of its SPUR_WIDTH bins) is acquired by the reference's own
                                                    store_new_spur / spur_phase_lock once spur_start transforms exist, then tracked and subtracted
                                                    by eliminate_spurs inside make_fft2 (fft2.c:647-652) */
  int clever = AI("clever", 0);                  /* 1: the linear ("clever") blanker: init_blanker (buf.c:1771-2057) builds the pulse tables from the amplitude
                                                    calibration fft1_desired (file desired=, N1 floats), first_noise_blanker then fits and subtracts pulses */
  double clever_factor = AF("clever_factor", 10.0);
  const char *fdesired = arg(argc, argv, "desired", NULL);
  int spur_pnt = AI("spur_pnt", 0), spur_start = AI("spur_start", 16), spur_spek = AI("spur_speknum", 0);
  /* spur_click<i>_at / _pnt (i = 1..4): at fft2 transform number _at the operator clicks the high-resolution graph at fft2 bin _pnt with the
     spur button armed: the reference's whole init_spur_elimination (spursub.c:181-343) -- peak search in spursearch_spectrum around the click,
     store_new_spur, spur_phase_lock, initial_remove_spur, then the ordering pass that drops the weaker of two spurs closer than four bins and
     keeps the list sorted by frequency (remove_spur, swap_spurs) -- instead of the harness's own store / lock calls */
  int click_at[4], click_pnt[4], nclick = 0;
  for (int i = 1; i <= 4; i++) { char ka[32], kp[32]; snprintf(ka, sizeof ka, "spur_click%d_at", i); snprintf(kp, sizeof kp, "spur_click%d_pnt", i);
    if (AI(ka, 0) > 0) { click_at[nclick] = AI(ka, 0); click_pnt[nclick] = AI(kp, 0); nclick++; } }
  int mix2on = AI("mix2", 0);                    /* 1: fft3_mix2's filter / decimate part (mixer_mode 1) after every make_fft3_all */
  int mixer_mode = AI("mixer_mode", 1);          /* 2: bg.mixer_mode = 2, the FIR decimator on timf3 (mix2.c:217-246) instead of the filter on fft3's bins */
  double pol_c1 = AF("pol_c1", 1.0), pol_c2 = AF("pol_c2", 0.0), pol_c3 = AF("pol_c3", 0.0);   /* pg.c1..c3 (two channels) */
  double ch2_c1 = AF("ch2_c1", 1.0), ch2_c2 = AF("ch2_c2", 0.0);   /* pg_ch2_c1 / pg_ch2_c2 (pol_graph.c:165-170), fft1.c:4064-4080 */
  const char *ffold = arg(argc, argv, "foldcorr", NULL);   /* N1 complex floats: enables CALIQ with this fft1_foldcorr */
  const char *fin = arg(argc, argv, "in", NULL);
  const char *flim = arg(argc, argv, "liminfo", NULL);
  const char *fout = arg(argc, argv, "out", "ref_dump.bin");
  int N1 = 1 << n1, N2 = 1 << n2;
  if (!fin) { fprintf(stderr, "need in=<int16 iq file>\n"); return 2; }
  fo = fopen(fout, "wb"); if (!fo) { perror(fout); return 2; }

  /* ---- ui / genparm ---- */
  memset(&ui, 0, sizeof(ui));
  if (C != 1 && C != 2) { fprintf(stderr, "channels must be 1 or 2\n"); return 2; }
  ui.rx_input_mode = (realin ? 0 : IQ_DATA) | (dword ? DWORD_INPUT : 0) | (C == 2 ? TWO_CHANNELS : 0); ui.rx_rf_channels = C; ui.rx_ad_channels = (realin ? 1 : 2) * C;
  pg_ch2_c1 = (float)ch2_c1; pg_ch2_c2 = (float)ch2_c2;
  ui.sample_shift = sshift; ui.rx_ad_speed = 1; ui.network_flag = 0; ui.operator_skil = 0;
  genparm[FIRST_FFT_SINPOW] = sinpow1; genparm[FIRST_FFT_VERNR] = 0;   /* -> fft_cntrl[7] radix-2 DIF C */
#ifdef SHIM_HARNESS
  genparm[FIRST_FFT_VERNR] = realin ? 4 : 7;                           /* -> fft_cntrl[21] "HIP MI355X" / [22] "HIP MI355X real" (the patch's new columns) */
#endif
  genparm[FIRST_FFT_GAIN] = gain; genparm[FIRST_FFT_BANDWIDTH] = 100;
  genparm[SECOND_FFT_ENABLE] = second; genparm[FIRST_BCKFFT_VERNR] = 0; genparm[FIRST_BCKFFT_ATT_N] = att_n;
  genparm[SECOND_FFT_SINPOW] = sinpow2; genparm[SECOND_FFT_VERNR] = 0; /* -> fft_cntrl[15] */
  genparm[SECOND_FFT_ATT_N] = 8; genparm[MAX_NO_OF_SPURS] = 0; genparm[AFC_ENABLE] = 0; genparm[AFC_LOCK_RANGE] = 0;
  genparm[MIX1_BANDWIDTH_REDUCTION_N] = mixred; genparm[MIX1_NO_OF_CHANNELS] = 1;
  fft1mode = (ui.rx_input_mode & (TWO_CHANNELS + IQ_DATA)) / 2;
  rx_channels = C; twice_rxchan = 2 * C; sw_onechan = C == 1; swfloat = 1; swmmx_fft2 = 0; swmmx_fft1 = 0;
  kill_all_flag = 0; lir_status = 0; fft1_correlation_flag = (corr && C == 2) ? 1 : 0; fft1afc_flag = 0; no_of_spurs = 0;
  ampinfo_flag = 0; audio_dump_flag = 0; fft1_use_gpu = 0; fft1_calibrate_flag = 0; fft1_direction = direction;
  yieldflag_wdsp_fft1 = 0; yieldflag_timf2_fft1 = 0; yieldflag_fft2_fft2 = 0; yieldflag_ndsp_mix1 = 0;

  /* ---- fft1 sizes and tables (buf.c:193-304, 1395-1459) ---- */
  fft1_n = n1; fft1_size = N1; fft1_block = 2 * C * N1; fft1_muln = 1; fft1_mulblock = fft1_block;
  {
    /* interleave: reference formula buf.c:303-304 with make_interleave_ratio (buf.c:113-136) */
    double ratio = 0;
    if (sinpow1 != 0) ratio = (sinpow1 == 9) ? 0.625 : (sinpow1 == 8) ? 0.8 : 2 * asin(pow(0.5, 1.0 / sinpow1)) / PI_L;
    fft1_interleave_ratio = (float)ratio;
    fft1_interleave_points = 1 + fft1_interleave_ratio * fft1_size;
    fft1_interleave_points &= 0xfffe;
    if (!second) {               /* buf.c:315-327: mix1 sized from fft1, interleave re-derived from it */
      mix1.n = n1 - mixred; if (mix1.n < 3) mix1.n = 3; mix1.size = 1 << mix1.n;
      mix1.interleave_points = fft1_interleave_ratio * mix1.size;
      mix1.interleave_points &= 0xfffffffe;
      fft1_interleave_points = mix1.interleave_points * (fft1_size / mix1.size);
      mix1.new_points = mix1.size - mix1.interleave_points;
    }
  }
  fft1_new_points = N1 - fft1_interleave_points;
  max_fft1n = maxfft1n; fft1n_mask = max_fft1n - 1; fft1_mask = max_fft1n * fft1_block - 1;
  fft1_float = zalloc(sizeof(float) * max_fft1n * fft1_block);
  fft1tab = zalloc(sizeof(COSIN_TABLE) * N1);
  fft1_permute = zalloc(sizeof(short) * N1 * 2);
  fft1_window = zalloc(sizeof(float) * (N1 + 32));
  fft1_filtercorr = (float *)zalloc(sizeof(float) * (2 * C * N1 + 32)) + 8;
  fft1_desired = zalloc(sizeof(float) * N1);
  fftw_tmp = zalloc(sizeof(float) * (4 * C * N1 + 64));
  /* buf.c:1403-1433 with fft_cntrl[FFT1_CURMODE]: IQ -> version 7 {window 1, permute 1}; real -> version 2 {window 2, permute 2} */
  make_sincos(realin ? 2 : 1, N1, fft1tab);
  make_permute(fft_cntrl[FFT1_CURMODE].permute, n1, N1, fft1_permute);
  make_window(fft_cntrl[FFT1_CURMODE].window, N1, sinpow1, fft1_window);
  clear_fft1_filtercorr();
  if (ffold) {                                    /* I/Q mirror-image calibration table (caliq.c), here from a file */
    fft1_foldcorr = zalloc(sizeof(float) * (2 * N1 + 32));
    FILE *ff = fopen(ffold, "rb"); if (!ff || fread(fft1_foldcorr, sizeof(float), 2 * N1, ff) != (size_t)(2 * N1)) { perror(ffold); return 2; }
    fclose(ff);
    fft1_calibrate_flag |= CALIQ;
  }

  /* timf1 input ring */
  FILE *fi = fopen(fin, "rb"); if (!fi) { perror(fin); return 2; }
  fseek(fi, 0, SEEK_END); long inbytes = ftell(fi); fseek(fi, 0, SEEK_SET);
  timf1_bytes = 1; while (timf1_bytes < inbytes) timf1_bytes <<= 1;
  if (ring_log2) timf1_bytes = 1 << ring_log2;
  timf1_bytemask = timf1_bytes - 1;
  timf1_char = zalloc(timf1_bytes); timf1_short_int = (short *)timf1_char; timf1_int = (int *)timf1_char;
  {
    /* file longer than the ring simply wraps, as a producer would */
    char *tmpb = malloc(inbytes); if (fread(tmpb, 1, inbytes, fi) != (size_t)inbytes) return 2;
    for (long i = 0; i < inbytes; i++) timf1_char[i & timf1_bytemask] = tmpb[i];
    free(tmpb);
  }
  fclose(fi);
  timf1_blockbytes = fft1_new_points * (dword ? 8 : 4) * C;
  timf1p_px = 0;

  /* spectrum averaging (fft1_c, update_fft1_slowsum) */
  wg.fft_avg1num = avg1; wg_fft_avg2num = avg2; wg.first_xpoint = 0; wg.xpoints = N1 - 1;
  wg.waterfall_avgnum = wf_avg; wg.spek_avgnum = avg1 * avg2;
  fft1_sumsq_bufsize = sumsq_blocks * N1; fft1_sumsq_mask = fft1_sumsq_bufsize - 1;
  fft1_sumsq = zalloc(sizeof(float) * fft1_sumsq_bufsize);
  fft1_slowsum = zalloc(sizeof(float) * N1);
  fft1_sumsq_pa = 0; fft1_sumsq_counter = 0; change_fft1_flag = 0; latest_wg_spectrum = 0;
  if (corr && C == 2) {                            /* buf.c:1223-1233, 1472, 1767 */
    fft1_corrsum = zalloc(sizeof(float) * 2 * fft1_sumsq_bufsize); fft1_slowcorr = zalloc(sizeof(double) * 2 * N1);
    fft1_slowcorr_tot = zalloc(sizeof(double) * 2 * N1);
    slowcorr_tot_avgnum = 0; correlation_reset_flag = 0; fft1corr_reset_flag = 0;
  }
  set_fft1_endpoints();            /* fft1_first_point=0, last=N1-1, recalc pointer, sym points */
  { /* an amplitude calibration moves the in-band end points inside the filter's skirts (fft1.c:4631-4637: fft1_desired < 0.5); a case may ask for that
       without carrying a calibration */
    int v_;
    if ((v_ = AI("first_inband", -1)) >= 0) fft1_first_inband = v_;
    if ((v_ = AI("last_inband", -1)) >= 0) fft1_last_inband = v_; }
  fft1_pa = fft1_pb = fft1_px = 0; fft1_na = fft1_nb = fft1_nx = 0; fft1_nm = 0; fft1_liminfo_cnt = 0;
  ag_pa = 0; ag_mask = 1023;

  /* ---- timf2 / back transform (buf.c:371-430, 1297-1345) ---- */
  fft2_n = n2; fft2_size = N2;
  timf2pow_size = timf2pow_log2 ? (1 << timf2pow_log2) : 8 * (N2 > N1 ? N2 : N1);
  timf2pow_mask = timf2pow_size - 1; timf2_size = 4 * C * timf2pow_size; timf2_mask = timf2_size - 1;
  timf2_input_block = fft1_new_points * 4 * C;
  timf2_pa = timf2_px = timf2_pn1 = timf2_pn2 = timf2_pb = timf2_pc = timf2_pt = 0; timf2p_fit = 0;
  timf2_float = zalloc(sizeof(float) * (timf2_size + 8 * C * N1));
  timf2_pwr_float = zalloc(sizeof(float) * (timf2pow_size + 2 * N1));
  timf2_tmp = zalloc(sizeof(float) * 8 * C * N1);
  fft1_split_float = zalloc(sizeof(float) * 8 * C * N1);
  fft1_back_scramble = zalloc(sizeof(short) * N1);
  fft1_inverted_window = zalloc(sizeof(float) * (N1 + 32));
  liminfo = zalloc(sizeof(float) * 2 * N1);      /* second half: old_liminfo (sellim.c:748) */
  if (sellim) {                                  /* buf.c:816-824, 956-958, 1469, 1523, 1759-1760; hires_graph.c:1175-1189 */
    int groups = AI("lim_groups", 16);
    liminfo_group_points = N1 / groups; if (liminfo_group_points < 1) liminfo_group_points = 1;
    liminfo_wait = zalloc(N1); liminfo_group_min = zalloc(sizeof(float) * (N1 + 8)); fftt_tmp = (float *)zalloc(sizeof(float) * (2 * N1 + 32)) + 8;
    fft1_sumsq_tot = 0; sel_ia = 0; sel_ib = 0;
    genparm[SELLIM_MAXLEVEL] = AI("maxlevel", 12000); fft1_blocktime = (float)AF("blocktime", 0.0008);
    baseband_bw_fftxpts = AI("bw_fftxpts", 40);
    memset(&mg, 0, sizeof mg);
    liminfo_groups = groups;                       /* buf.c:816-820, 972-978 */
    fftf_tmp = zalloc(sizeof(float) * ((size_t)N2 + N1 + 64));
    reg_noise = zalloc(sizeof(float) * (groups + 8)); reg_min = zalloc(sizeof(float) * (groups + 8)); reg_ston = zalloc(sizeof(float) * (groups + 8));
    reg_first_point = zalloc(sizeof(int) * (groups + 8)); reg_length = zalloc(sizeof(int) * (groups + 8));   /* buf.c:979-980 (variant 1) */
  }
  make_permute(0, n1, N1, fft1_back_scramble);
  if (fft_cntrl[FFT1_CURMODE].permute == 2) { fft1_backtab = zalloc(sizeof(COSIN_TABLE) * N1); make_sincos(0, N1, fft1_backtab); }   /* buf.c:1318-1326 */
  else fft1_backtab = fft1tab;
  if (sinpow1 != 2 && sinpow1 != 0) make_window(3, N1, sinpow1, fft1_inverted_window);
  fft1_lowlevel_fraction = .75f;
  float *limrecs = NULL; long nlimrec = 0;
  if (flim) {
    FILE *fl = fopen(flim, "rb"); if (!fl) { perror(flim); return 2; }
    fseek(fl, 0, SEEK_END); long lb = ftell(fl); fseek(fl, 0, SEEK_SET);
    nlimrec = lb / (4 * N1); limrecs = malloc(lb);
    if (fread(limrecs, 1, lb, fl) != (size_t)lb) return 2;
    fclose(fl);
    memcpy(liminfo, limrecs, 4 * N1);
  }

  /* ---- blanker (buf.c:337-346, 418-431, 2083-2086; hires_graph.c:1157-1162) ---- */
  memset(&hg, 0, sizeof(hg));
  timf2_noise_floor_avgnum = bln_avgnum; blanker_info_update_interval = bln_interval; blanker_info_update_counter = 0;
  timf2_noise_floor = noise_floor0; timf2_despiked_pwr[0] = timf2_noise_floor; timf2_despiked_pwrinc[0] = 1;
  timf2_despiked_pwr[1] = 0; timf2_despiked_pwrinc[1] = 0;
  timf2_fitted_pulses = 0; timf2_cleared_points = 0; timf2_blanker_points = 0;
  clever_blanker_rate = 0; stupid_blanker_rate = 0;
  hg.stupid_bln_mode = stupid; hg.clever_bln_mode = 0; hg.stupid_bln_factor = 5; hg.clever_bln_factor = 10;
  hg.stupid_bln_limit = (unsigned int)((float)timf2_noise_floor * hg.stupid_bln_factor);
  hg.clever_bln_limit = (unsigned int)((float)timf2_noise_floor * hg.clever_bln_factor);
  hg.timf2_oscilloscope = 0;
  hg.sellim_par1 = AI("par1", 2); hg.sellim_par2 = AI("par2", 0); hg.sellim_par3 = AI("par3", 0); hg.sellim_par4 = AI("par4", 0);
  hg.sellim_par5 = AI("par5", 0); hg.sellim_par6 = AI("par6", 0); hg.sellim_par7 = AI("par7", 0); hg.sellim_par8 = AI("par8", 0);
  hg.blanker_ston_fft1 = (float)AF("ston_fft1", 4.0); hg.blanker_ston_fft2 = (float)AF("ston_fft2", 30.0);
  blnfit_range = fitrange; blanker_pulsewidth = pulsewidth;
  blanker_flag = zalloc(timf2pow_size + 64);
  if (clever) {
    /* init_blanker's calibrated branch (buf.c:1786-2057) run on a synthetic amplitude calibration: the pulse response tables
       (bln[], blanker_refpulse, blanker_phasefunc, blanker_pulindex), blanker_pulsewidth and blnfit_range are the reference's own */
    if (!fdesired || (C == 2 && (chain2 || !AI("blanker2", 0)))) { fprintf(stderr, "clever=1 needs desired=<file>; two channels: with blanker2=1, without chain2\n"); return 2; }
    float *des = zalloc(sizeof(float) * N1), *keep = fft1_desired;
    FILE *fd = fopen(fdesired, "rb"); if (!fd || fread(des, sizeof(float), N1, fd) != (size_t)N1) { perror(fdesired); return 2; }
    fclose(fd);
    int keepflag = fft1_calibrate_flag;
    fft1_blockbytes = fft1_block * sizeof(float);
    screen_width = 64; screen_height = 64;
    fft1_desired = des; fft1_calibrate_flag |= CALAMP;
    init_blanker();
    fft1_desired = keep; fft1_calibrate_flag = keepflag;
    memset(fft1_float, 0, sizeof(float) * max_fft1n * fft1_block);      /* init_blanker used it as scratch */
    memset(fftw_tmp, 0, sizeof(float) * (4 * C * N1 + 64));
    hg.clever_bln_mode = 1; hg.clever_bln_factor = (float)clever_factor;
    hg.clever_bln_limit = (unsigned int)((float)timf2_noise_floor * hg.clever_bln_factor);
    fprintf(stderr, "init_blanker: refpul_size %d pulsewidth %d largest_blnfit %d blnfit_range %d\n", refpul_size, blanker_pulsewidth, largest_blnfit, blnfit_range);
    float bl[4 * BLN_INFO_SIZE];
    for (int i = 0; i < BLN_INFO_SIZE; i++) { bl[4 * i] = bln[i].size; bl[4 * i + 1] = bln[i].rest; bl[4 * i + 2] = bln[i].avgmax; bl[4 * i + 3] = bln[i].avgpwr; }
    PUTF("bln", bl, 4 * BLN_INFO_SIZE);
    int bi[5] = { refpul_size, blanker_pulsewidth, largest_blnfit, blnfit_range, MAX_REFPULSES }; PUTI("bln_ints", bi, 5);
    PUTF("bln_fparams", ((float[]){ liminfo_amplitude_factor, hg.clever_bln_factor }), 2);
    PUTF("blanker_refpulse", blanker_refpulse, (size_t)2 * MAX_REFPULSES * refpul_size);
    PUTF("blanker_phasefunc", blanker_phasefunc, (size_t)2 * refpul_size);
    PUTI("blanker_pulindex", blanker_pulindex, MAX_REFPULSES);
  }
  min_delay_time = (bln_minpts >= 0) ? (float)bln_minpts : (float)N2 / 3.0f;   /* x ui.rx_ad_speed(=1), buf.c:500-509 */
  timf2_oscilloscope_counter = 0; timf2_oscilloscope_maxpoint = 0; timf2_oscilloscope_maxval_float = 0;
  timf2_oscilloscope_powermax_float = 0; timf2_show_pointer = -1; timf2_oscilloscope_interval = 15;

  /* ---- fft2 (mode 15) ---- */
  fft2_float = zalloc(sizeof(float) * 2 * C * N2 * maxfft2n);
  if (C == 2) { fft2_xypower = zalloc(sizeof(TWOCHAN_POWER) * N2 * maxfft2n); fft2_xysum = zalloc(sizeof(TWOCHAN_POWER) * N2); }
  max_fft2n = maxfft2n; fft2n_mask = max_fft2n - 1;
  fft2_tab = zalloc(sizeof(COSIN_TABLE) * N2);
  fft2_bigpermute = zalloc(sizeof(int) * N2);
  fft2_window = zalloc(sizeof(float) * (N2 + 32));
  fft2_power_float = zalloc(sizeof(float) * N2 * maxfft2n);
  fft2_powersum_float = zalloc(sizeof(float) * N2);
  make_bigpermute(fft_cntrl[FFT2_CURMODE].permute, n2, N2, fft2_bigpermute);
  make_sincos(1, N2, fft2_tab);
  if (sinpow2 != 0) make_window(fft_cntrl[FFT2_CURMODE].window, N2, sinpow2, fft2_window);
  fft2_pa = 0; fft2_na = fft2_nb = fft2_nx = fft2_nm = 0; fft2_liminfo_cnt = 0;
  hg_redraw_counter = 0; hg.spek_avgnum = 1 << 30; fft2_blocktime = 0;
  fft2_to_fft1_ratio = N2 / N1; if (fft2_to_fft1_ratio < 1) fft2_to_fft1_ratio = 1;
  float *spur_trace = NULL, *spur_trace_all = NULL, click_log[4 * 8]; int nspur_trace = 0, nspur_all = 0, spur_locked_at = -1;
  memset(click_log, 0, sizeof click_log);
  float *ss_last = NULL, ss_thr[64]; int ss_completed = 0, ss_at[64];       /* the search spectrum as the newest spursearch_spectrum_cleanup left it (fft2.c:676-683) */
  if (spur) {                                    /* buf.c:1100-1172, 1252, 1647 */
    const int ms = 4;
    genparm[MAX_NO_OF_SPURS] = ms; genparm[AFC_ENABLE] = 1;
    max_fftxn = max_fft2n; fftxn_mask = max_fftxn - 1; fftx_size = fft2_size; fftx = fft2_float; fftx_pwr = fft2_power_float;
    if (!second) {               /* second fft off: the spurs live in the fft1 transforms, fft1_c's AFC branch takes them out (buf.c:729, 930-945, 1089; fft1.c:4196-4244) */
      fft1afc_flag = 1;
      max_fftxn = max_fft1n; fftxn_mask = max_fftxn - 1; fftx_size = fft1_size; fftx = fft1_float;
      fft1_power = zalloc(sizeof(float) * (size_t)max_fft1n * fft1_size); fftx_pwr = fft1_power;
    }
    swmmx_fft2 = 0; no_of_spurs = 0;
    spur_block = SPUR_WIDTH * max_fftxn * twice_rxchan;
    spur_table = zalloc(sizeof(float) * ms * spur_block); spur_location = zalloc(4 * ms); spur_flag = zalloc(4 * ms);
    spur_power = zalloc(4 * (SPUR_WIDTH + 1)); spur_d0pha = zalloc(4 * ms); spur_d1pha = zalloc(4 * ms); spur_d2pha = zalloc(4 * ms);
    spur_ampl = zalloc(4 * ms); spur_noise = zalloc(4 * ms); spur_avgd2 = zalloc(4 * ms); spur_pol = zalloc(12 * ms);
    spur_spectra = zalloc(4 * NO_OF_SPUR_SPECTRA * SPUR_SIZE + 64); spur_freq = zalloc(4 * ms);
    spur_ind = zalloc(4 * ms * max_fftxn); spur_signal = zalloc(4 * max_fftxn * twice_rxchan * ms);
    spursearch_spectrum = zalloc(4 * fftx_size); spursearch_powersum = zalloc(4 * fftx_size);
    for (int i = 0; i < ms * max_fftxn; i++) spur_ind[i] = -1;
    spur_freq_factor = second ? (float)fft2_new_points / fft2_size : (float)fft1_new_points / fft1_size;             /* buf.c:480, 1118 */
    spur_speknum = spur_spek > 0 ? spur_spek : max_fftxn / 4;
    if (spur_speknum < 4) spur_speknum = 4;
    /* the reference carves these out of its big fft scratch, fftx_size / 4 floats each (buf.c:1285-1293); spur_phase_lock / verify_spur_pll index them
       beyond 2 * max_fftxn floats (AddressSanitizer: 48 bytes past arrays of max_fftxn + 8 complex values, into the neighbouring heap block) */
    { const size_t nsp = sizeof(float) * (size_t)(fftx_size / 4 > 2 * (max_fftxn + 8) ? fftx_size / 4 : 2 * (max_fftxn + 8));
      sp_sig = zalloc(nsp); sp_der = zalloc(nsp); sp_pha = zalloc(nsp); sp_tmp = zalloc(nsp); }
    spursearch_sum_counter = 0;
    sp_numsub = spur_speknum - 1; sp_avgnum = spur_speknum / 3; if (sp_avgnum > 10) sp_avgnum = 10;
    spur_max_d2 = PI_L * spur_freq_factor / spur_speknum;
    spur_minston = 1 / sqrt(0.5 * (float)(spur_speknum));
    { float t1 = 0.5 * spur_speknum; spur_weiold = t1 / (1 + t1); spur_weinew = 1 / (1 + t1);
      t1 = -0.5 * sp_numsub; spur_linefit = 0; for (int i = 0; i < spur_speknum; i++) { spur_linefit += t1 * t1; t1 += 1; } }
    spur_search_first_point = 0; spur_search_last_point = fftx_size - 1;
    init_spur_spectra();
    spur_trace = zalloc(sizeof(float) * 12 * ((size_t)nblk * 8 + 64));
    spur_trace_all = zalloc(sizeof(float) * 40 * ((size_t)nblk * 8 + 64));
    ss_last = zalloc(4 * (size_t)fftx_size);
  }

  /* mix1 sizes first: fft2 interleave is re-derived from mix1 (buf.c:432-455) */
  if (second) { mix1.n = n2 - mixred; if (mix1.n < 3) mix1.n = 3; mix1.size = 1 << mix1.n; }
  if (second) {
    double ratio = 0;
    if (sinpow2 != 0) ratio = (sinpow2 == 9) ? 0.625 : (sinpow2 == 8) ? 0.8 : 2 * asin(pow(0.5, 1.0 / sinpow2)) / PI_L;
    fft2_interleave_ratio = (float)ratio;
    mix1.interleave_points = fft2_interleave_ratio * mix1.size;
    mix1.interleave_points &= 0xfffffffe;
    fft2_interleave_points = mix1.interleave_points * (fft2_size / mix1.size);
    fft2_new_points = fft2_size - fft2_interleave_points;
    mix1.new_points = mix1.size - mix1.interleave_points;
  }
  timf2_output_block = 4 * C * fft2_new_points;
  if (sellim2) fft2_blocktime = fft1_blocktime * (float)fft2_new_points / (float)fft1_new_points;   /* buf.c:456: both from timf1_sampling_speed */

  /* waterfall line from fft2 (fft2.c:707-815) */
  wg_xpixels = wf_pix ? wf_pix : (N2 < 1024 ? N2 : 1024);
  hgwat_first_xpoint = wf_first;
  if (wf_mode == 1) { hgwat_xpoints_per_pixel = 1; hgwat_pixels_per_xpoint = 1; }
  else if (wf_mode > 1) { hgwat_xpoints_per_pixel = wf_mode; hgwat_pixels_per_xpoint = 0; }
  else { hgwat_xpoints_per_pixel = 0; hgwat_pixels_per_xpoint = -wf_mode; }
  /* a2 in fft2.c:713-722 is the step in fft1 bins per pixel */
  if (hgwat_xpoints_per_pixel >= fft2_to_fft1_ratio && hgwat_xpoints_per_pixel > 0)
    { wg.xpoints_per_pixel = hgwat_xpoints_per_pixel / fft2_to_fft1_ratio; wg.pixels_per_xpoint = 0; }
  else
    { wg.xpoints_per_pixel = 0;
      wg.pixels_per_xpoint = hgwat_pixels_per_xpoint > 0 ? hgwat_pixels_per_xpoint * fft2_to_fft1_ratio
                                                         : fft2_to_fft1_ratio / (hgwat_xpoints_per_pixel > 0 ? hgwat_xpoints_per_pixel : 1);
      if (wg.pixels_per_xpoint < 1) wg.pixels_per_xpoint = 1; }
  wg.first_xpoint = wf_first / fft2_to_fft1_ratio;
  wg_waterf_size = 8 * wg_xpixels; wg_waterf = zalloc(2 * wg_waterf_size + 64); wg_waterf_ptr = 0;
  wg_waterf_yfac = zalloc(sizeof(float) * (N1 + 8));
  {
    /* make_wg_yfac (wide_graph.c:955-1001), second-fft branch, uncalibrated fft1_desired */
    float t1 = (float)(FFT2_WATERFALL_ZERO) / ((float)fft2_size * (float)fft1_size);
    t1 /= (float)sqrt((float)(wg.waterfall_avgnum));
    t1 *= (float)(1 << (2 * genparm[FIRST_BCKFFT_ATT_N]));
    t1 *= (float)(1 + 1 / (0.5 + genparm[FIRST_FFT_SINPOW]));
    t1 *= (float)(ui.rx_rf_channels * ui.rx_rf_channels);
    for (int i = 0; i < N1; i++)
      wg_waterf_yfac[i] = (fft1_desired[i] > 0.3162278) ? t1 / (float)pow(fft1_desired[i], 2.0) : t1 * 10;
    wg_waterf_yfac[0] = t1; wg_waterf_yfac[N1 - 1] = t1;
  }
  wg_waterf_sum_counter = 0;

  /* ---- mix1 (buf.c:55-111, 1297; mix1.c:781-861) ---- */
  mix1.table = zalloc(sizeof(COSIN_TABLE) * mix1.size);
  mix1.permute = zalloc(sizeof(short) * mix1.size * 2);
  mix1.window = zalloc(sizeof(float) * (mix1.size + 32));
  mix1.cos2win = zalloc(sizeof(float) * (mix1.size + 32));
  mix1.sin2win = zalloc(sizeof(float) * (mix1.size + 32));
  mix1_fqwin = zalloc(sizeof(float) * (mix1.size + 32));
  make_window(5, mix1.size, 4, mix1_fqwin);
  rx_mode = 0;
  prepare_mixer(&mix1, second ? SECOND_FFT_SINPOW : FIRST_FFT_SINPOW);   /* the reference's own, buf.c:55-111 */
  fftn_tmp = zalloc(sizeof(float) * (4 * C * mix1.size + 64));
  timf3_block = 2 * C * mix1.new_points;
  timf3_size = 16 * 2 * C * mix1.size; timf3_mask = timf3_size - 1;
  timf3_float = zalloc(sizeof(float) * (2 * timf3_size + 4 * mix1.size));
  timf3_pa = timf3_px = timf3_py = 0;
  fftx_points_per_hz = 1.0f; mix1_lowest_fq = 0; mix1_highest_fq = (float)(second ? N2 : N1);
  timf2_blockpower_block = bp_block; timf2_blockpower_size = bp_size; timf2_blockpower_mask = bp_size - 1;
  timf2_blockpower = zalloc(sizeof(float) * 2 * bp_size); timf2_blockpower_pa = 0; timf2_pb = 0;
  mix1_selfreq[0] = fq; old_mix1_selfreq = fq; mix1_point[0] = -1;
  mix1_phase[0] = 0; mix1_phase_step[0] = 0; mix1_phase_rot[0] = 0; mix1_old_phase[0] = 0; mix1_old_point[0] = 0;

  /* ---- fft3 (baseb_graph.c:636-645, 3369-3409, 3679-3680): transform part of make_fft3_all only ---- */
  int nfft3 = 0;
  if (n3 > 0) {
    fft3_n = n3; fft3_size = 1 << n3;
    mix2.n = nm2; mix2.size = 1 << nm2;
    {
      double ratio = 0;
      if (sinpow3 != 0) ratio = (sinpow3 == 9) ? 0.625 : (sinpow3 == 8) ? 0.8 : 2 * asin(pow(0.5, 1.0 / sinpow3)) / PI_L;
      fft3_interleave_ratio = (float)ratio;
      mix2.interleave_points = fft3_interleave_ratio * mix2.size;
      mix2.interleave_points &= 0xfffffffe;
      mix2.new_points = mix2.size - mix2.interleave_points;
      fft3_interleave_points = mix2.interleave_points * (fft3_size / mix2.size);
      fft3_new_points = fft3_size - fft3_interleave_points;
    }
    genparm[THIRD_FFT_SINPOW] = sinpow3;
    fft3_block = fft3_size * 2 * C; fft3_totsiz = fft3_block * maxfft3n; fft3_mask = fft3_totsiz - 1;
    fft3 = zalloc(sizeof(float) * fft3_totsiz); fft3_tmp = zalloc(sizeof(float) * (4 * C * fft3_size + 64));
    fft3_tab = zalloc(sizeof(COSIN_TABLE) * fft3_size); fft3_permute = zalloc(sizeof(short) * fft3_size * 2);
    fft3_window = zalloc(sizeof(float) * (fft3_size + 32));
    init_fft(1, fft3_n, fft3_size, fft3_tab, fft3_permute);
    make_window(1, fft3_size, sinpow3, fft3_window);
    fft3_pa = fft3_px = 0; timf3_px = 0; yieldflag_ndsp_fft3 = 0;
    thread_command_flag[THREAD_FFT3] = THRFLAG_ACTIVE;
    memset(&bg, 0, sizeof(bg)); bg.fft_avgnum = 1; bg.waterfall_avgnum = 1 << 30;
    bg_xpoints = 0; bg_show_pa = 0; bg_first_xpoint = 0; fft3_slowsum_recalc = 0; fft3_slowsum_cnt = 0; fft3_show_size = 1;
    bg_avg_counter = 0; bg_filter_points = 0; bg_waterf_sum_counter = 0;
    if (mix2on) {                 /* baseb_graph.c:718-730, 899; buffers of fft3_mix2's mixer_mode 1 part */
      mix2.table = zalloc(sizeof(COSIN_TABLE) * mix2.size); mix2.permute = zalloc(sizeof(short) * mix2.size * 2);
      mix2.window = zalloc(sizeof(float) * (mix2.size + 32)); mix2.cos2win = zalloc(sizeof(float) * (mix2.size + 32));
      mix2.sin2win = zalloc(sizeof(float) * (mix2.size + 32));
      prepare_mixer(&mix2, THIRD_FFT_SINPOW);
      mi2_tmp = zalloc(sizeof(float) * (4 * C * mix2.size + 64)); carr_tmp = zalloc(sizeof(float) * (4 * C * mix2.size + 64));
      bg_filterfunc = zalloc(sizeof(float) * (fft3_size + 8)); bg_carrfilter = zalloc(sizeof(float) * (fft3_size + 8));
      for (int i = 0; i < fft3_size; i++) { double u = (i - fft3_size / 2.0) / (fft3_size / 6.0); bg_filterfunc[i] = (float)exp(-u * u); }   /* stand-in for make_bg_filter's output */
      baseband_size = 4096; baseband_mask = baseband_size - 1;
      baseb_raw = zalloc(sizeof(float) * (2 * baseband_size + 8 * mix2.size)); baseb_raw_orthog = zalloc(sizeof(float) * (2 * baseband_size + 8 * mix2.size));
      fft3_slowsum = zalloc(sizeof(float) * (2 * C * fft3_size + 64));   /* read (not used with pg.adapt != 0) by the two-channel branch, mix2.c:344 */
      baseb_pa = 0; bg.mixer_mode = mixer_mode; fm_pilot_size = 0; genparm[CW_DECODE_ENABLE] = 0; yieldflag_ndsp_mix2 = 1;
      timf3_py = 0;
      if (mixer_mode == 2) {        /* stand-in for the FIR make_bg_filter derives from the filter function (baseb_graph.c:1560-1634): a Hann-windowed
                                       low pass of fft3_size/4 + 1 points, half the baseband wide, unit sum; symmetric (centre = pts/2) like the reference's */
        basebraw_fir_pts = fft3_size / 4 + 1;
        basebraw_fir = zalloc(sizeof(float) * (fft3_size + 8));
        double sum = 0; const int h = basebraw_fir_pts / 2;
        for (int i = 0; i < basebraw_fir_pts; i++) { double x = (i - h) * 0.5 * mix2.size / fft3_size, sinc = fabs(x) < 1e-12 ? 1.0 : sin(PI_L * x) / (PI_L * x);
          double w = 0.5 + 0.5 * cos(PI_L * (i - h) / (h + 1.0)); basebraw_fir[i] = (float)(sinc * w); sum += basebraw_fir[i]; }
        for (int i = 0; i < basebraw_fir_pts; i++) basebraw_fir[i] = (float)(basebraw_fir[i] / sum);
      }
      memset(&pg, 0, sizeof(pg)); pg.c1 = (float)pol_c1; pg.c2 = (float)pol_c2; pg.c3 = (float)pol_c3; pg.adapt = 1; pg.avg = 1;
    }
  }

  /* ---- dump tables ---- */
  {
    int hdr[24] = { n1, N1, fft1_interleave_points, n2, N2, fft2_interleave_points, (int)mix1.n, (int)mix1.size,
                    (int)mix1.interleave_points, att_n, gain, avg1, avg2, nblk, timf2pow_size, max_fft1n, max_fft2n,
                    fft1_sumsq_bufsize, timf1_bytes, timf3_size, wg_xpixels, hgwat_first_xpoint, wf_mode, wf_avg };
    PUTI("hdr", hdr, 24);
    PUTF("fft1_window", fft1_window, N1);
    put("fft1_permute", "u2", fft1_permute, N1, 2);
    PUTF("fft1tab", fft1tab, N1);                     /* N1/2 {sin,cos} pairs */
    PUTF("fft1_filtercorr", fft1_filtercorr, 2 * C * N1);
    PUTF("fft1_desired", fft1_desired, N1);
    put("fft1_back_scramble", "u2", fft1_back_scramble, N1, 2);
    PUTF("fft1_inverted_window", fft1_inverted_window, N1 / 2 + 1);
    PUTF("fft2_window", fft2_window, N2);
    PUTF("mix1_fqwin", mix1_fqwin, mix1.size / 2 + 1);
    PUTF("wg_waterf_yfac", wg_waterf_yfac, N1);
  }

#define RUN_FFT3() do { if (n3 > 0) while (((timf3_pa - timf3_px + timf3_size) & timf3_mask) >= 2 * C * fft3_size && \
      ((fft3_pa - fft3_px + fft3_totsiz) & fft3_mask) < fft3_totsiz - 2 * fft3_block) { make_fft3_all(); nfft3++; \
      if (mix2on) { thread_command_flag[THREAD_MIX2] = mixer_mode == 2 ? THRFLAG_IDLE : THRFLAG_ACTIVE; /* (mode 2 has no back transform to yield in: the check of mix2.c:749 is the first one) */ \
        mix2_trip = 1; fft3_mix2(); mix2_trip = 0; \
        baseb_pa = (baseb_pa + mix2.new_points) & baseband_mask; /* = last_point, mix2.c:1079, 2056 */ \
        timf3_py = (timf3_py + 2 * C * fft3_new_points) & timf3_mask; /* mix2.c:2060 */ } \
      fft3_px = (fft3_px + fft3_block) & fft3_mask; /* mix2.c:2058; the rest of fft3_mix2 (demodulators) is not run */ } } while (0)
  /* ---- AFC tables (buf.c:1089-1092, 1255-1258) and the synthetic frequency supplier for the afc variants ---- */
  int afcn = second ? max_fft2n : max_fft1n;
  mix1_fq_mid = zalloc(4 * afcn); mix1_fq_start = zalloc(4 * afcn); mix1_fq_curv = zalloc(4 * afcn); mix1_fq_slope = zalloc(4 * afcn);
  for (int i = 0; i < afcn; i++) { mix1_fq_mid[i] = -1; mix1_fq_start[i] = -1; }
  baseband_bw_hz = (float)afc_bw; fftxn_mask = afcn - 1;
  int afc_t = 0;
  float *afc_supplied = zalloc(4 * ((size_t)nblk * 64 + 64));
#define AFC_FQ(t) ((float)(fq + 1.5 * sin(2 * PI_L * (t) / 23.0) + ((t) >= 30 && (t) < 60 ? 3.0 : 0.0)))
#define AFC_SUPPLY(nx, mask) do { if (afc_t == 0) mix1_fq_mid[nx] = AFC_FQ(0); \
      mix1_fq_mid[((nx) + 1) & (mask)] = AFC_FQ(afc_t + 1); afc_supplied[afc_t] = mix1_fq_mid[((nx) + 1) & (mask)]; afc_t++; } while (0)
#ifdef SHIM_HARNESS
  /* what get_wideband_sizes (buf.c:247-257, patched) and wideband_dsp's start (wcw.c:576, patched) do with version 21 selected */
  if (fft_cntrl[FFT1_CURMODE].gpu != GPU_HIP) { fprintf(stderr, "version 21 not selected: FFT1_CURMODE %d\n", FFT1_CURMODE); return 2; }
  const int shim_threads = AI("shim_threads", 0);   /* 1: stage functions on Linrad's stage threads in lock step; 2: the same threads running free */
  const int shim_batch = AI("shim_batch", 1);       /* gpu_fft1_batch_size = 2^gpu.fft1_batch_n transforms per fft1_b call (buf.c:248-264, 610-613) */
  fft1_use_gpu = GPU_HIP; gpu_fft1_batch_size = shim_batch; gpu.fft1_device = 0; no_of_fft1b = shim_threads ? 1 : 0;
  fft1_muln = shim_batch; fft1_mulblock = fft1_block * fft1_muln; timf1_blockbytes *= shim_batch;
  if (shim_batch < 1 || (shim_batch & (shim_batch - 1)) || shim_batch > max_fft1n / 2 || nblk % shim_batch) { fprintf(stderr, "shim_batch: a power of two <= max_fft1n/2 that divides nblk\n"); return 2; }
  if (AI("shim_refuse", 0) == 1) genparm[MIX1_NO_OF_CHANNELS] = 2;      /* a mode version 21 must refuse */
  const int shim_net = AI("shim_net", 0);           /* 1: ui.network_flag asks for the FFT1 / TIMF2 / FFT2 multicasts: the hooks the patch puts in front of the senders'
                                                       reads (wcw.c:1024-1043, rxin.c:944, 1026) are called where the senders would read, and the host rings they fill are dumped */
  if (shim_net) ui.network_flag = NET_RXOUT_FFT1 | NET_RXOUT_TIMF2 | NET_RXOUT_FFT2;
  hip_sparse_rings = AI("shim_sparse", 0) ? -1 : 0;                     /* 0 (default here): the device rings themselves are compared below; 1: as a patched xlinrad64 opens them */
  { int rc = hip_open(); fprintf(stderr, "hip_open: %d\n", rc); if (rc != 0) { printf("{\"hip_open\": %d}\n", rc); fclose(fo); return AI("shim_refuse", 0) ? 0 : 3; } }
  glue_ingest = timing && shim_threads == 2;                            /* the timed glue: block by block from the dispatcher, like the input thread */
  if (!glue_ingest)
  hip_timf1_new(0, timf1_bytes);                                        /* finish_rx_read's hand-over (rxin.c:1425, patched), the whole recording at once */
#endif
  /* ---- run ---- */
  float *trace = zalloc(sizeof(float) * TR_COLS * nblk);
  int *itrace = zalloc(sizeof(int) * TR_COLS * nblk);
  int nfft2 = 0; int nwf = 0;
  size_t max_fft2_calls = (size_t)nblk * (size_t)(fft1_new_points / (fft2_new_points > 0 ? fft2_new_points : 1) + 2) + 64;
  float *mixtrace = zalloc(sizeof(float) * 8 * max_fft2_calls);
  short *wf_lines = zalloc(2 * (size_t)wg_xpixels * max_fft2_calls);
  float *fft1_first = zalloc(sizeof(float) * 2 * C * N1);    /* fft1_b output of block 0 before fft1_c */
  int local_fft1_liminfo_cnt = 0, nlimupd = 0;
  float *limtrace = sellim ? zalloc(sizeof(float) * N1 * (size_t)(nblk / avg1 + 2)) : NULL;
  int *limtrace_blk = zalloc(sizeof(int) * (nblk / avg1 + 2));
  int local_fft2_liminfo_cnt = 0, nlimupd2 = 0, namp = 0;
  float *limtrace2 = (sellim && sellim2) ? zalloc(sizeof(float) * N1 * (size_t)(4 * nblk + 8)) : NULL;
  int *limtrace2_blk = zalloc(sizeof(int) * (4 * nblk + 8));
  float *amptrace = zalloc(sizeof(float) * (5 * nblk + 16));
  fft2_liminfo_cnt = 0;
  struct timespec ts0, ts1; clock_gettime(CLOCK_MONOTONIC, &ts0);
  int nthreads = AI("threads", 0);
  if (timing && nthreads && second && C == 1) {
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    int workers = nthreads > 0 ? nthreads - 3 : (int)ncpu - 3;
    if (workers < 1) workers = 1;
    if (workers > 6) workers = 6;                 /* MAX_FFT1_THREADS, thrdef.h:107 */
    memset(&TH, 0, sizeof TH);
    TH.nblk = nblk; TH.workers = workers; TH.C = C; TH.n3 = n3; TH.mix2on = mix2on; TH.N2 = N2; TH.fq_ok = fq >= 0;
    for (int k = 0; k < workers; k++) TH.tmp[k] = zalloc(sizeof(float) * (4 * C * N1 + 64));
    run_reference_threads();
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    printf("{\"loop_seconds\": %.6f, \"blocks\": %d, \"samples\": %ld, \"fft2\": %d, \"cleared\": %d, \"threads\": %d, "
           "\"topology\": \"dispatcher + %d fft1_b workers + timf2 + second_fft + narrowband threads\"}\n",
           (ts1.tv_sec - ts0.tv_sec) + 1e-9 * (ts1.tv_nsec - ts0.tv_nsec), nblk, (long)nblk * fft1_new_points, TH.nfft2, timf2_cleared_points,
           workers + 4, workers);
    fclose(fo);
    return harness_err ? 3 : 0;
  }
#ifdef SHIM_HARNESS
  if (shim_threads == 2) {                         /* free-running stage threads + fft1_b workers: final rings and pointers only */
    if (!second || C != 1) { fprintf(stderr, "shim_threads=2: one channel, second fft on\n"); return 2; }
    int workers = AI("shim_workers", 3);
    if (workers < 1) workers = 1; if (workers > 6) workers = 6;
    memset(&TH, 0, sizeof TH);
    TH.nblk = nblk / shim_batch; TH.workers = workers; TH.C = C; TH.n3 = n3; TH.mix2on = mix2on; TH.N2 = N2; TH.fq_ok = fq >= 0;
    no_of_fft1b = workers;
    for (int k = 0; k < workers; k++) TH.tmp[k] = zalloc(sizeof(float) * (4 * C * N1 + 64));
    if (timing) {
      /* THE DROP-IN'S RATE (bench.py "glue"): the patched reference objects + hipshim.c + liblinrad_hip.so, Linrad's thread topology
         (dispatcher + fft1_b workers + THREAD_TIMF2 + THREAD_SECOND_FFT + the narrowband thread, wcw.c:401-441, 250-304, 476-500), samples
         through the producer hook a dispatch at a time, every host-visible product brought back by the glue as in a running xlinrad64.
         warm=W dispatches first (module load, lazily made streams and buffers), then the clock runs over the rest until every thread has
         ended and the device is idle. */
      const int warm = AI("warm", 64) / shim_batch > 0 ? AI("warm", 64) / shim_batch : 1;
      const int total = TH.nblk;
      if (total <= warm) { fprintf(stderr, "timing: nblk must exceed warm\n"); return 2; }
      TH.nblk = warm; run_reference_threads(); lrh_sync(hip_context());
      const int nfft2_warm = TH.nfft2;
      for (int k = 0; k < SG_N; k++) { sg_sec[k] = 0; sg_calls[k] = 0; }
      sg_on = 1;
      clock_gettime(CLOCK_MONOTONIC, &ts0);
      TH.nblk = total - warm; run_reference_threads(); lrh_sync(hip_context());
      clock_gettime(CLOCK_MONOTONIC, &ts1);
      const double dt = (ts1.tv_sec - ts0.tv_sec) + 1e-9 * (ts1.tv_nsec - ts0.tv_nsec);
      printf("{\"loop_seconds\": %.6f, \"blocks\": %d, \"samples\": %ld, \"fft2\": %d, \"threads\": %d, \"fft1_batch\": %d, \"warm_blocks\": %d, "
             "\"topology\": \"rx input thread (producer hook) + dispatcher + %d fft1_b workers + timf2 + second_fft + narrowband threads\", \"stage_calls\": {",
             dt, (total - warm) * shim_batch, (long)(total - warm) * shim_batch * fft1_new_points, TH.nfft2 - nfft2_warm, workers + 5, shim_batch, warm * shim_batch, workers);
      for (int k = 0; k < SG_N; k++) printf("%s\"%s\": {\"calls\": %ld, \"us_per_call\": %.2f, \"busy_frac\": %.3f}", k ? ", " : "", sg_name[k], sg_calls[k],
                                            sg_calls[k] ? 1e6 * sg_sec[k] / sg_calls[k] : 0.0, sg_sec[k] / dt);
      printf("}}\n");
      hip_close(); fclose(fo);
      return harness_err ? 3 : 0;
    }
    run_reference_threads();
    nfft2 = TH.nfft2;
    goto run_done;
  }
  if (shim_threads == 1) stage_start();
  const int blk_step = shim_batch;
#else
  const int blk_step = 1;
#endif
  for (int b = 0; b < nblk && !harness_err; b += blk_step) {
    if (lim_every > 0 && limrecs && (b % lim_every) == 0) {
      long r = b / lim_every; if (r >= nlimrec) r = nlimrec - 1;
      memcpy(liminfo, limrecs + r * N1, 4 * N1);
    }
    fft1b_job.inptr = timf1p_px; fft1b_job.out = &fft1_float[fft1_pa]; fft1b_job.tmp = fftw_tmp; fft1b_job.handle = 0;
    ON(ST_WORKER, fft1b_stage);
#ifdef SHIM_HARNESS
    if (shim_net && b == 0) hip_net_fft1(timf1p_px, fft1_pa);            /* the dispatcher's NET_RXOUT_FFT1 branch (wcw.c:1038-1043, patched) */
#endif
    if (b == 0) memcpy(fft1_first, &fft1_float[fft1_pa], 8 * C * N1);
    timf1p_px = (timf1p_px + timf1_blockbytes) & timf1_bytemask;
    fft1_pa = (fft1_pa + fft1_mulblock) & fft1_mask;
    fft1_na = fft1_pa / fft1_block;
    if (fft1_nm != fft1n_mask) fft1_nm++;
    if (!second) {               /* second fft disabled: fft1_c, then the narrowband thread's fft1_mix1_fixed */
      while (fft1_na != fft1_nb) {
        const int ss_before = spursearch_sum_counter;
        ON(ST_TIMF2, fft1_c);              /* THREAD_DO_FFT1C with more than one CPU (wcw.c:1070-1078) */
        if (!spur) continue;
        spur_max_d2 = PI_L * spur_freq_factor / spur_speknum;
        if (ss_before > 3 * spur_speknum && spursearch_sum_counter == 0) {      /* this transform completed a search spectrum (fft1.c:4432-4440) */
          memcpy(ss_last, spursearch_spectrum, 4 * (size_t)fftx_size);
          if (ss_completed < 64) { ss_thr[ss_completed] = spur_search_threshold; ss_at[ss_completed] = b; }
          ss_completed++;
        }
        if (no_of_spurs == 0 && b + 1 == spur_start) {    /* acquisition behind fft1_c (spur_removal, wcw.c:708-713) */
          /* ffts_na: the slot behind the newest transform, as second_fft leaves it for the fft2 flavour (wcw.c:288-289), so that the history
             store_new_spur takes ends with the newest transform.  Linrad's own fft1 flow leaves ffts_na at the newest transform itself
             (fft1.c:4206), i.e. one short: an empty row then sits in the loop's first fits, a transient whose outcome hangs on the last bit
             of libm (the same oracle binary kept lock on one host and lost it on another) -- no pin.  The glue passes Linrad's value through. */
          ffts_na = fft1_nb; ffts_nm = fft1_nm;
          spurno = 0; spur_ampl[0] = 1; spur_noise[0] = 0.001; spur_avgd2[0] = 0;
          int rc1 = store_new_spur(spur_pnt), rc2 = rc1 ? -9 : spur_phase_lock(ffts_na);
          fprintf(stderr, "spur acquisition at fft1 transform %d: store %d lock %d loc %d freq %.4f ampl %.4g noise %.4g d0 %.4f d1 %.4f d2 %.5f\n", b, rc1, rc2,
                  spur_location[0], spur_freq[0], spur_ampl[0], spur_noise[0], spur_d0pha[0], spur_d1pha[0], spur_d2pha[0]);
          if (!rc1 && !rc2) {
            no_of_spurs = 1; spur_locked_at = b + 1;
            initial_remove_spur();                        /* init_spur_elimination's next step (spursub.c:309) */
            float st[12] = { (float)spur_location[0], (float)spur_flag[0], spur_freq[0], spur_d0pha[0], spur_d1pha[0], spur_d2pha[0], spur_ampl[0], spur_noise[0], spur_avgd2[0],
                             (float)ffts_na, (float)spur_speknum, spur_freq_factor };
            PUTF("spur_init_state", st, 12);
            PUTF("spur_init_table", spur_table, spur_block); PUTF("spur_init_signal", spur_signal, 2 * max_fftxn); PUTI("spur_init_ind", spur_ind, max_fftxn);
            PUTF("spur_spectra", spur_spectra, NO_OF_SPUR_SPECTRA * SPUR_SIZE);
          }
        } else if (no_of_spurs > 0) {
          float *q = spur_trace + 12 * nspur_trace++;
          q[0] = spur_location[0]; q[1] = spur_flag[0]; q[2] = spur_freq[0]; q[3] = spur_d0pha[0]; q[4] = spur_d1pha[0]; q[5] = spur_d2pha[0];
          q[6] = spur_ampl[0]; q[7] = spur_noise[0]; q[8] = spur_avgd2[0]; q[9] = b; q[10] = no_of_spurs;
        }
      }
      while (fq >= 0 && fft1_nx != fft1_nb) {                        /* narrowband_dsp: until fft1_nx has caught up (wcw.c:1690-1712) */
        if (afc) { AFC_SUPPLY(fft1_nx, fft1n_mask); ON(ST_NARROW, fft1_mix1_afc); } else
        ON(ST_NARROW, fft1_mix1_fixed);
        float *m = mixtrace + 8 * nfft2;
        m[0] = mix1_point[0]; m[1] = mix1_phase[0]; m[2] = mix1_phase_rot[0]; m[3] = mix1_phase_step[0];
        m[4] = mix1_old_phase[0]; m[5] = mix1_old_point[0]; m[6] = timf3_pa; m[7] = fft1_nx;
        nfft2++;
        RUN_FFT3();
      }
      int *it0 = itrace + TR_COLS * b;
      it0[9] = fft1_sumsq_pa; it0[10] = fft1_sumsq_counter; it0[15] = fft1_liminfo_cnt; it0[14] = nfft2; it0[8] = fft1_nx;
      continue;
    }
    while (fft1_na != fft1_nb) { ON(ST_TIMF2, fft1_c); ON(ST_TIMF2, make_timf2); }
    if (C == 2 && !chain2) {     /* two channels: make_timf2, then (blanker2=1) the two-channel first_noise_blanker; fft2 / mix1 only with chain2=1 */
      int *it2 = itrace + TR_COLS * b; float *t2 = trace + TR_COLS * b;
      int pbeg2 = timf2p_fit;
      if (AI("blanker2", 0)) first_noise_blanker();
      it2[0] = timf2_pa; it2[1] = timf2p_fit; it2[2] = timf2_pn2; it2[4] = pbeg2; it2[5] = timf2_cleared_points;
      it2[6] = timf2_blanker_points; it2[7] = blanker_info_update_counter;
      it2[9] = fft1_sumsq_pa; it2[10] = fft1_sumsq_counter; it2[11] = fft1_lowlevel_points; it2[12] = timf2_noise_floor;
      it2[13] = (int)hg.stupid_bln_limit; it2[15] = fft1_liminfo_cnt;
      t2[0] = (float)timf2_noise_floor; t2[1] = (float)hg.stupid_bln_limit; t2[2] = stupid_blanker_rate;
      t2[3] = timf2_despiked_pwr[0]; t2[4] = timf2_despiked_pwrinc[0]; t2[5] = fft1_lowlevel_fraction;
      t2[6] = timf2_despiked_pwr[1]; t2[7] = timf2_despiked_pwrinc[1];
      t2[8] = (float)hg.clever_bln_limit; t2[9] = clever_blanker_rate; t2[10] = (float)timf2_fitted_pulses;      /* get_pulse_pol / subtract_twochan_pulse at work (clever=1) */
      continue;
    }
    int pbeg = timf2p_fit;
    ON(ST_TIMF2, first_noise_blanker);
    if (bp_block > 0) ON(ST_TIMF2, compute_timf2_powersum);
    while (((timf2_pn2 - timf2_px + timf2_size) & timf2_mask) >= 4 * C * fft2_size) {     /* wcw.c:265-266 */
      int wptr = wg_waterf_ptr;
      if (spur) { ffts_na = fft2_na; ffts_nm = fft2_nm;          /* what the previous pass of second_fft left (wcw.c:288-289) */
        spur_freq_factor = (float)fft2_new_points / fft2_size; spur_max_d2 = PI_L * spur_freq_factor / spur_speknum; }   /* buf.c:480, 1152 (fft2_new_points is known by now) */
      make_fft2_status = FFT2_NOT_ACTIVE;
      const int ss_before = spursearch_sum_counter;
      while (make_fft2_status != FFT2_COMPLETE) ON(ST_FFT2, make_fft2);
      if (spur && ss_before > 3 * spur_speknum && spursearch_sum_counter == 0) {      /* this transform completed a search spectrum */
        memcpy(ss_last, spursearch_spectrum, 4 * (size_t)fftx_size);
        if (ss_completed < 64) { ss_thr[ss_completed] = spur_search_threshold; ss_at[ss_completed] = nfft2; }
        ss_completed++;
      }
      if (spur && nclick) {
        for (int c = 0; c < nclick; c++) if (nfft2 + 1 == click_at[c]) {
          ffts_na = fft2_na; ffts_nm = fft2_nm;                    /* what second_fft leaves for spur_removal (wcw.c:288-303) */
          no_of_scro = 1; scro[0].no = HIRES_GRAPH; scro[0].x1 = 0; scro[0].x2 = 1 << 30; scro[0].y1 = 0; scro[0].y2 = 1 << 30;
          mouse_x = click_pnt[c]; mouse_y = 1; hg_first_xpixel = 0; hg_first_point = 0;
          autospur_point = spur_search_first_point + SPUR_WIDTH / 2 + 1;      /* (the manual branch insists on it, spursub.c:189) */
          const int before = no_of_spurs;
          if (getenv("LRH_HARNESS_DEBUG")) { fprintf(stderr, "search spectrum (threshold %g, ffts_nm %d) around the click:", spur_search_threshold, ffts_nm);
            for (int i = click_pnt[c] - 8; i <= click_pnt[c] + 8; i++) fprintf(stderr, " %.3g", spursearch_spectrum[i]); fprintf(stderr, "\n"); }
          init_spur_elimination();
          float *q = click_log + 8 * c;
          q[0] = nfft2; q[1] = before; q[2] = no_of_spurs; for (int i = 0; i < 4 && i < no_of_spurs; i++) q[3 + i] = spur_location[i];
          fprintf(stderr, "click %d at transform %d on bin %d: spurs %d -> %d, locations", c, nfft2, click_pnt[c], before, no_of_spurs);
          for (int i = 0; i < no_of_spurs; i++) fprintf(stderr, " %d (ampl %.4g)", spur_location[i], spur_ampl[i]);
          if (no_of_spurs == before && before < 4) { q[7] = spur_location[before];     /* the slot behind the list: what an acquisition that was dropped again left there */
            fprintf(stderr, "; slot %d behind the list: location %d ampl %.4g", before, spur_location[before], spur_ampl[before]); }
          fprintf(stderr, "\n");
          if (spur_locked_at < 0 && no_of_spurs > 0) spur_locked_at = nfft2 + 1;
        }
        float *q = spur_trace_all + 40 * nspur_all++;
        q[0] = nfft2; q[1] = no_of_spurs;
        for (int i = 0; i < 4 && i < no_of_spurs; i++) { float *r = q + 2 + 9 * i;
          r[0] = spur_location[i]; r[1] = spur_flag[i]; r[2] = spur_freq[i]; r[3] = spur_d0pha[i]; r[4] = spur_d1pha[i]; r[5] = spur_d2pha[i];
          r[6] = spur_ampl[i]; r[7] = spur_noise[i]; r[8] = spur_avgd2[i]; }
      } else
      if (spur && no_of_spurs == 0 && nfft2 + 1 == spur_start) {   /* acquisition, tail of init_spur_elimination (spursub.c:282-309) */
        ffts_na = fft2_na; ffts_nm = fft2_nm;
        spurno = 0; spur_ampl[0] = 1; spur_noise[0] = 0.001; spur_avgd2[0] = 0;
        int rc1 = store_new_spur(spur_pnt), rc2 = rc1 ? -9 : spur_phase_lock(ffts_na);
        fprintf(stderr, "spur acquisition at transform %d: store %d lock %d loc %d freq %.4f ampl %.4g noise %.4g d0 %.4f d1 %.4f d2 %.5f\n", nfft2, rc1, rc2,
                spur_location[0], spur_freq[0], spur_ampl[0], spur_noise[0], spur_d0pha[0], spur_d1pha[0], spur_d2pha[0]);
        if (!rc1 && !rc2) {
          no_of_spurs = 1; spur_locked_at = nfft2 + 1;
          initial_remove_spur();                          /* init_spur_elimination's next step (spursub.c:309) */
          float st[12] = { (float)spur_location[0], (float)spur_flag[0], spur_freq[0], spur_d0pha[0], spur_d1pha[0], spur_d2pha[0], spur_ampl[0], spur_noise[0], spur_avgd2[0],
                           (float)fft2_na, (float)spur_speknum, spur_freq_factor };
          PUTF("spur_init_state", st, 12);
          PUTF("spur_init_table", spur_table, spur_block); PUTF("spur_init_signal", spur_signal, 2 * max_fftxn); PUTI("spur_init_ind", spur_ind, max_fftxn);
          PUTF("spur_spectra", spur_spectra, NO_OF_SPUR_SPECTRA * SPUR_SIZE);
          PUTF("spur_init_fft2", fft2_float, (size_t)2 * N2 * max_fft2n);
        }
      } else if (spur && no_of_spurs > 0) {
        float *q = spur_trace + 12 * nspur_trace++;
        q[0] = spur_location[0]; q[1] = spur_flag[0]; q[2] = spur_freq[0]; q[3] = spur_d0pha[0]; q[4] = spur_d1pha[0]; q[5] = spur_d2pha[0];
        q[6] = spur_ampl[0]; q[7] = spur_noise[0]; q[8] = spur_avgd2[0]; q[9] = nfft2; q[10] = no_of_spurs;
      }
      if (wg_waterf_ptr != wptr) { memcpy(wf_lines + (size_t)nwf * wg_xpixels, wg_waterf + wptr, 2 * wg_xpixels); nwf++; }
      if (fq >= 0) {
        if (afc) { AFC_SUPPLY(fft2_nx, fft2n_mask); ON(ST_NARROW, fft2_mix1_afc); } else
        ON(ST_NARROW, fft2_mix1_fixed);
        float *m = mixtrace + 8 * nfft2;
        m[0] = mix1_point[0]; m[1] = mix1_phase[0]; m[2] = mix1_phase_rot[0]; m[3] = mix1_phase_step[0];
        m[4] = mix1_old_phase[0]; m[5] = mix1_old_point[0]; m[6] = timf3_pa; m[7] = fft2_nx;
        RUN_FFT3();
      } else {
        fft2_nx = (fft2_nx + 1) & fft2n_mask;
      }
      nfft2++;
    }
    float *t = trace + TR_COLS * b; int *it = itrace + TR_COLS * b;
    t[0] = (float)timf2_noise_floor; t[1] = (float)hg.stupid_bln_limit; t[2] = stupid_blanker_rate;
    t[3] = timf2_despiked_pwr[0]; t[4] = timf2_despiked_pwrinc[0]; t[5] = fft1_lowlevel_fraction;
    it[0] = timf2_pa; it[1] = timf2p_fit; it[2] = timf2_pn2; it[3] = timf2_px; it[4] = pbeg;
    it[5] = timf2_cleared_points; it[6] = timf2_blanker_points; it[7] = blanker_info_update_counter;
    it[8] = fft2_na; it[9] = fft1_sumsq_pa; it[10] = fft1_sumsq_counter; it[11] = fft1_lowlevel_points;
    it[12] = timf2_noise_floor; it[13] = (int)hg.stupid_bln_limit; it[14] = nfft2; it[15] = fft1_liminfo_cnt;
    t[6] = (float)hg.clever_bln_limit; t[7] = clever_blanker_rate; t[8] = (float)timf2_fitted_pulses;
    if (sellim && fft1_liminfo_cnt != local_fft1_liminfo_cnt) {        /* wcw.c:1124-1128 */
      fft1_update_liminfo();
      local_fft1_liminfo_cnt = fft1_liminfo_cnt;
      memcpy(limtrace + (size_t)nlimupd * N1, liminfo, 4 * N1); limtrace_blk[nlimupd] = b; nlimupd++;
      amptrace[namp++] = liminfo_amplitude_factor;
    }
    if (sellim && sellim2 && fft2_liminfo_cnt != local_fft2_liminfo_cnt) {   /* wcw.c:1129-1133 */
      fft2_update_liminfo();
      local_fft2_liminfo_cnt = fft2_liminfo_cnt;
      memcpy(limtrace2 + (size_t)nlimupd2 * N1, liminfo, 4 * N1); limtrace2_blk[nlimupd2] = b; nlimupd2++;
      amptrace[namp++] = liminfo_amplitude_factor;
    }
  }

#ifdef SHIM_HARNESS
  stage_stop();
run_done:
#endif
  clock_gettime(CLOCK_MONOTONIC, &ts1);
  if (timing) {
    printf("{\"loop_seconds\": %.6f, \"blocks\": %d, \"samples\": %ld, \"fft2\": %d, \"cleared\": %d}\n",
           (ts1.tv_sec - ts0.tv_sec) + 1e-9 * (ts1.tv_nsec - ts0.tv_nsec), nblk, (long)nblk * fft1_new_points, nfft2, timf2_cleared_points);
    fclose(fo);
    return harness_err ? 3 : 0;
  }
#ifdef SHIM_HARNESS
  /* the rings Linrad never sees with version 21, fetched for the comparison; everything else below is the host's own copy as the
     glue kept it (fft1_sumsq, fft1_slowsum, fft2_powersum_float, wg_waterf lines, timf3_float, timf2_blockpower, liminfo, scalars) */
  if (C == 2) {
    /* two channels: one context each; Linrad's arrays hold {ch0, ch1} per bin / sample (fft1.c:2041, timf2.c:210, fft2.c:1622) */
    const size_t n1f = (size_t)max_fft1n * 2 * N1, n2f = (size_t)timf2pow_size * 4, nf2 = (size_t)2 * N2 * max_fft2n;
    float *t = zalloc(sizeof(float) * (n1f > n2f ? (n1f > nf2 ? n1f : nf2) : (n2f > nf2 ? n2f : nf2)));
    for (int ch = 0; ch < 2; ch++) {
      lrh_export(hip_context_of(ch), LRH_RING_FFT1_FLOAT, t, 0, n1f);
      for (size_t i = 0; i < n1f / 2; i++) { fft1_float[4 * i + 2 * ch] = t[2 * i]; fft1_float[4 * i + 2 * ch + 1] = t[2 * i + 1]; }
      if (!second) continue;
      lrh_export(hip_context_of(ch), LRH_RING_TIMF2_FLOAT, t, 0, n2f);
      for (size_t i = 0; i < (size_t)timf2pow_size; i++) for (int ws = 0; ws < 2; ws++) for (int k = 0; k < 2; k++) timf2_float[8 * i + 4 * ws + 2 * ch + k] = t[4 * i + 2 * ws + k];
      lrh_export(hip_context_of(ch), LRH_RING_TIMF2_PWR, t, 0, timf2pow_size);
      for (size_t i = 0; i < (size_t)timf2pow_size; i++) timf2_pwr_float[i] = (ch ? timf2_pwr_float[i] : 0) + t[i];
      lrh_export(hip_context_of(ch), LRH_RING_FFT2_FLOAT, t, 0, nf2);
      for (size_t i = 0; i < nf2 / 2; i++) { fft2_float[4 * i + 2 * ch] = t[2 * i]; fft2_float[4 * i + 2 * ch + 1] = t[2 * i + 1]; }
    }
    if (second && shim_net) {                        /* ... and over them what the network thread's walks fetch through the patched senders' hooks: must be the same */
      memset(timf2_float, 0, sizeof(float) * timf2_size); memset(fft2_float, 0, sizeof(float) * 2 * nf2);
      for (int pt = 0; pt < timf2_size; pt += 696) hip_net_timf2(pt, pt + 696 <= timf2_size ? 696 : timf2_size - pt);
      for (int pt = 0; pt < (int)(2 * nf2); pt += 348) hip_net_fft2(pt, pt + 348 <= (int)(2 * nf2) ? 348 : (int)(2 * nf2) - pt);
    }
    if (second && chain2) {
      lrh_export(hip_context(), LRH_RING_FFT2_XYPOWER, fft2_xypower, 0, (size_t)4 * N2 * max_fft2n);
      lrh_export(hip_context(), LRH_RING_FFT2_XYSUM, fft2_xysum, 0, (size_t)4 * N2);
    }
    free(t);
  } else {
  lrh_export(hip_context(), LRH_RING_FFT1_FLOAT, fft1_float, 0, (size_t)max_fft1n * fft1_block);
  if (second && shim_net) {                          /* the network thread's walks (rxin.c:944-966, 1026-1035), a packet's worth at a time, from the pointers it keeps */
    for (int pt = 0; pt < timf2_size; pt += 696) hip_net_timf2(pt, pt + 696 <= timf2_size ? 696 : timf2_size - pt);
    for (int pt = 0; pt < 2 * N2 * max_fft2n; pt += 348) hip_net_fft2(pt, pt + 348 <= 2 * N2 * max_fft2n ? 348 : 2 * N2 * max_fft2n - pt);
    lrh_export(hip_context(), LRH_RING_TIMF2_PWR, timf2_pwr_float, 0, timf2pow_size);
    lrh_export(hip_context(), LRH_RING_FFT2_POWER, fft2_power_float, 0, (size_t)N2 * max_fft2n);
  } else if (second) {
    lrh_export(hip_context(), LRH_RING_TIMF2_FLOAT, timf2_float, 0, timf2_size);
    lrh_export(hip_context(), LRH_RING_TIMF2_PWR, timf2_pwr_float, 0, timf2pow_size);
    lrh_export(hip_context(), LRH_RING_FFT2_FLOAT, fft2_float, 0, (size_t)2 * N2 * max_fft2n);
    lrh_export(hip_context(), LRH_RING_FFT2_POWER, fft2_power_float, 0, (size_t)N2 * max_fft2n);
  }
  }
  hip_close();
#endif
  /* ---- dump results ---- */
  PUTF("fft1_first_raw", fft1_first, 2 * C * N1);
  PUTF("fft1_float", fft1_float, (size_t)max_fft1n * fft1_block);
  PUTF("fft1_sumsq", fft1_sumsq, fft1_sumsq_bufsize);
  PUTF("fft1_slowsum", fft1_slowsum, N1);
  if (fft1_correlation_flag == 1) { PUTF("fft1_corrsum", fft1_corrsum, (size_t)2 * fft1_sumsq_bufsize); PUTF("fft1_slowcorr", fft1_slowcorr, (size_t)2 * N1);
    put("fft1_slowcorr_tot", "f8", fft1_slowcorr_tot, (size_t)2 * N1, 8); PUTI("slowcorr_tot_avgnum", &slowcorr_tot_avgnum, 1); }
  PUTF("timf2_float", timf2_float, timf2_size);
  PUTF("timf2_pwr_float", timf2_pwr_float, timf2pow_size);
  PUTF("fft2_float", fft2_float, (size_t)2 * C * N2 * max_fft2n);
  if (C == 2) { PUTF("fft2_xypower", (float *)fft2_xypower, (size_t)4 * N2 * max_fft2n); PUTF("fft2_xysum", (float *)fft2_xysum, (size_t)4 * N2); }
  PUTF("fft2_power_float", fft2_power_float, (size_t)N2 * max_fft2n);
  PUTF("fft2_powersum_float", fft2_powersum_float, N2);
  PUTF("timf3_float", timf3_float, timf3_size);
  if (n3 > 0) { PUTF("fft3", fft3, fft3_totsiz); PUTF("fft3_window", fft3_window, fft3_size);
    int f3[4] = { nfft3, fft3_pa, timf3_px, fft3_interleave_points }; PUTI("fft3_ptrs", f3, 4);
    if (mix2on) { PUTF("baseb_raw", baseb_raw, 2 * baseband_size); PUTF("baseb_raw_orthog", baseb_raw_orthog, 2 * baseband_size);
      PUTF("bg_filterfunc", bg_filterfunc, fft3_size); int bp_[2] = { baseb_pa, fft3_px }; PUTI("baseb_ptrs", bp_, 2);
      if (mixer_mode == 2) { PUTF("basebraw_fir", basebraw_fir, basebraw_fir_pts); PUTI("timf3_py", &timf3_py, 1); } } }
  if (sellim) {
    PUTF("liminfo_trace", limtrace, (size_t)N1 * (nlimupd > 0 ? nlimupd : 1)); PUTI("liminfo_trace_blk", limtrace_blk, nlimupd > 0 ? nlimupd : 1);
    int sp[16] = { nlimupd, genparm[SELLIM_MAXLEVEL], wg.spek_avgnum, liminfo_group_points, fft1_first_point, fft1_last_point, fft1_first_inband,
                   fft1_last_inband, baseband_bw_fftxpts, hg.sellim_par2, hg.sellim_par3, hg.sellim_par4, hg.sellim_par5, hg.sellim_par6, hg.sellim_par7, hg.sellim_par8 };
    PUTI("sellim_params", sp, 16);
    float sf[2] = { fft1_blocktime, hg.blanker_ston_fft1 }; PUTF("sellim_fparams", sf, 2);
    PUTF("liminfo_final", liminfo, N1);
    PUTF("amp_factor_trace", amptrace, namp > 0 ? namp : 1);
    if (sellim2) {
      PUTF("liminfo_trace2", limtrace2, (size_t)N1 * (nlimupd2 > 0 ? nlimupd2 : 1)); PUTI("liminfo_trace2_blk", limtrace2_blk, nlimupd2 > 0 ? nlimupd2 : 1);
      float s2[4] = { hg.blanker_ston_fft2, fft2_blocktime, (float)wg.waterfall_avgnum, (float)nlimupd2 }; PUTF("sellim2_fparams", s2, 4);
      if (hg.sellim_par1 != 2) { int v1[1] = { hg.sellim_par1 }; PUTI("sellim2_par1", v1, 1); }
    }
  }
  if (spur && nclick) { PUTF("spur_trace_all", spur_trace_all, (size_t)40 * nspur_all); PUTF("spur_clicks", click_log, 32); PUTF("spur_spectra", spur_spectra, NO_OF_SPUR_SPECTRA * SPUR_SIZE); }
  if (spur) { PUTF("spur_trace", spur_trace, (size_t)12 * (nspur_trace > 0 ? nspur_trace : 1)); int sl[2] = { spur_locked_at, nspur_trace }; PUTI("spur_locked", sl, 2);
    PUTF("spursearch_spectrum", ss_last, fftx_size); PUTF("spursearch_thresholds", ss_thr, ss_completed < 64 ? (ss_completed > 0 ? ss_completed : 1) : 64);
    int si[4] = { ss_completed, spursearch_sum_counter, spur_search_first_point, spur_search_last_point }; PUTI("spursearch_info", si, 4);
    PUTI("spursearch_at", ss_at, ss_completed < 64 ? (ss_completed > 0 ? ss_completed : 1) : 64); }
  PUTF("timf2_blockpower", timf2_blockpower, bp_size);
  { int bp[2] = { timf2_blockpower_pa, timf2_pb }; PUTI("blockpower_ptrs", bp, 2); }
  put("wf_lines", "i2", wf_lines, (size_t)nwf * wg_xpixels, 2);
  PUTF("trace", trace, (size_t)TR_COLS * nblk);
  PUTI("itrace", itrace, (size_t)TR_COLS * nblk);
  PUTF("mixtrace", mixtrace, (size_t)8 * (nfft2 > 0 ? nfft2 : 1));
  if (afc) {
    PUTF("afc_fq0", &(float){AFC_FQ(0)}, 1);
    PUTF("afc_supplied", afc_supplied, afc_t > 0 ? afc_t : 1);
    PUTF("afc_fq_mid", mix1_fq_mid, afcn); PUTF("afc_fq_slope", mix1_fq_slope, afcn);
    PUTF("afc_fq_curv", mix1_fq_curv, afcn); PUTF("afc_fq_start", mix1_fq_start, afcn);
  }
  {
    int fin_[12] = { fft1_pa, fft1_nb, fft1_nx, timf2_pa, timf2p_fit, timf2_pn2, timf2_px, fft2_na, fft2_nx, timf3_pa, nfft2, nwf };
    PUTI("final", fin_, 12);
  }
  fclose(fo);
  return harness_err ? 3 : 0;
}
