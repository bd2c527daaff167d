/*
 * lrh_threads.c -- the C ABI (include/linrad_hip.h) driven from Linrad's thread topology.
 *
 * Linrad runs its hot path as one thread per stage, handing blocks over through rings whose pointers are globals and
 * through binary auto-reset condition "events" (lxsys.c:415-447; names thrdef.h:136-170):
 *
 *   rx input thread      finish_rx_read: timf1p_pa advances, EVENT_TIMF1                        rxin.c:1425-1431
 *   THREAD_WIDEBAND_DSP  hands every complete block to an idle THREAD_FFT1Bk, retires the workers
 *                        in order, advances fft1_pa / fft1_na, EVENT_TIMF2                       wcw.c:969-1047, 1091
 *   THREAD_FFT1B1..n     fft1_b(timf1p_ref, out, tmp, gpu_handle_number = k)                     wcw.c:476-500
 *   THREAD_TIMF2         while(fft1_na != fft1_nb) { fft1_c; make_timf2; } first_noise_blanker;
 *                        EVENT_FFT2 when a transform's worth of samples is released              wcw.c:401-441
 *   THREAD_SECOND_FFT    make_fft2 until FFT2_COMPLETE, EVENT_FFT1_READY                        wcw.c:250-304
 *   THREAD_NARROWBAND    fft2_mix1_fixed for every new fft2 transform                            wcw.c:1240-1405
 *
 * This program does the same with pthreads over ONE lrh_ctx and ONE shared lrh_ptrs (the reference's globals): eight
 * host threads call the library concurrently, each stage advancing only its own pointer fields.  With `threads` = 0 the
 * same stage calls are made from one thread in the single-CPU order of wcw.c:1036-1118.  Both modes dump every ring to
 * `out`; tests/test_gpu_threads.py checks that the dumps are equal bit for bit.  (For that check the blanker is called once
 * per fft1 block in both modes: its statistics depend on the call pattern, which in Linrad depends on thread timing.)
 *
 *   gcc -O2 -Iinclude examples/lrh_threads.c -Llinrad_amd -llinrad_hip -lpthread -lm -o lrh_threads
 *   ./lrh_threads threads nblk out.bin [fft1_n fft2_n workers]
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "linrad_hip.h"

/* ---- events: binary, auto-reset, like lir_set_event / lir_await_event (lxsys.c:415-447) ---- */
enum { EVENT_TIMF1, EVENT_TIMF2, EVENT_FFT2, EVENT_FFT1_READY, EVENT_DO_FFT1B1, EVENT_FFT1B_DONE = EVENT_DO_FFT1B1 + 6, NEVENTS };
static pthread_mutex_t ev_mutex[NEVENTS];
static pthread_cond_t ev_cond[NEVENTS];
static int ev_flag[NEVENTS];
static void lir_set_event(int n) { pthread_mutex_lock(&ev_mutex[n]); ev_flag[n] = 1; pthread_cond_signal(&ev_cond[n]); pthread_mutex_unlock(&ev_mutex[n]); }
static void lir_await_event(int n)
{
  pthread_mutex_lock(&ev_mutex[n]);
  while (!ev_flag[n]) pthread_cond_wait(&ev_cond[n], &ev_mutex[n]);
  ev_flag[n] = 0;
  pthread_mutex_unlock(&ev_mutex[n]);
}

/* a failed stage ends the run (lirerr posts EVENT_KILL_ALL and every loop polls kill_all_flag, lxsys.c:494-505): wake everybody */
static volatile int failed;
static void space_freed(void);
static void fail_all(void) { failed = 1; for (int i = 0; i < NEVENTS; i++) lir_set_event(i); space_freed(); }

/* ---- the "globals" ---- */
static lrh_ctx *rx;
static lrh_config cfg;
static volatile lrh_ptrs p;                  /* shared like the reference's pointer globals; volatile like thrdef.h's flags */
static volatile int timf1p_pa;               /* producer pointer, bytes (rxin.c) */
static volatile int input_done, wide_done, timf2_done, fft2_done;
static int N1, N2, M1, M2, timf1_blockbytes, nblk_total, workers;
static char *timf1_char;
static lrh_synth sig;
#define P ((lrh_ptrs *)&p)
static void fail_all(void);
#define CHK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s: rc=%d (%s)\n", #call, rc_, lrh_last_error(rx)); fail_all(); } } while (0)

static int timf1_avail(void) { return (timf1p_pa - p.timf1p_px + cfg.timf1_bytes) & (cfg.timf1_bytes - 1); }
static int fft2_ready(void) { return ((p.timf2_pn2 - p.timf2_px + 4 * cfg.timf2pow_size) & (4 * cfg.timf2pow_size - 1)) >= 4 * N2; }   /* wcw.c:265 */
/* Room in the rings: a producer waits until its consumer has moved on (the reference counts overruns instead: "BUFFER ERROR",
   wcw.c:770-785).  One condition variable, broadcast, because several producers may wait at once. */
static pthread_mutex_t sp_mutex = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t sp_cond = PTHREAD_COND_INITIALIZER;
static void space_freed(void) { pthread_mutex_lock(&sp_mutex); pthread_cond_broadcast(&sp_cond); pthread_mutex_unlock(&sp_mutex); }
#define AWAIT_SPACE(cond) do { pthread_mutex_lock(&sp_mutex); while (!(cond) && !failed) pthread_cond_wait(&sp_cond, &sp_mutex); pthread_mutex_unlock(&sp_mutex); } while (0)
static int fft1_used(void) { return (p.fft1_na - p.fft1_nx + cfg.max_fft1n) & (cfg.max_fft1n - 1); }
static int timf2_used(void) { return (p.timf2_pa - p.timf2_px + 4 * cfg.timf2pow_size) & (4 * cfg.timf2pow_size - 1); }
static int fft2_used(void) { return (p.fft2_na - p.fft2_nx + cfg.max_fft2n) & (cfg.max_fft2n - 1); }

/* ---- rx input thread ---- */
static void *input_thread(void *arg)
{
  (void)arg;
  const int chunk = M1;                       /* one "soundcard read" */
  for (long done = 0; done < (long)nblk_total * M1 && !failed; done += chunk) {
    AWAIT_SPACE(timf1_avail() <= cfg.timf1_bytes / 2);
    if (timf1p_pa == 0) CHK(lrh_timf1_write_wait(rx));        /* once per lap: every copy of the previous lap has left the arena */
    lrh_synth_iq(&sig, done, chunk, (int16_t *)(timf1_char + timf1p_pa));
    CHK(lrh_timf1_write_async(rx, timf1_char + timf1p_pa, timf1p_pa, chunk * 4));
    timf1p_pa = (timf1p_pa + chunk * 4) & (cfg.timf1_bytes - 1);
    lir_set_event(EVENT_TIMF1);
  }
  input_done = 1;
  lir_set_event(EVENT_TIMF1);
  return NULL;
}

/* ---- fft1_b workers (do_fft1b, wcw.c:476-513) ---- */
static struct { volatile int inptr, out, busy; } job[6];
static void *fft1b_thread(void *arg)
{
  const int k = (int)(long)arg;
  for (;;) {
    lir_await_event(EVENT_DO_FFT1B1 + k);
    if (job[k].busy < 0 || failed) return NULL;
    CHK(lrh_fft1_b(rx, k + 1, job[k].inptr, job[k].out, 1));      /* gpu_handle_number = worker number (wcw.c:500) */
    job[k].busy = 2;                                               /* finished: the dispatcher retires it in order */
    lir_set_event(EVENT_FFT1B_DONE);
  }
}

/* ---- THREAD_WIDEBAND_DSP: dispatcher (wcw.c:969-1047) ---- */
static void retire(int k)
{
  while (job[k].busy != 2 && !failed) lir_await_event(EVENT_FFT1B_DONE);
  job[k].busy = 0;
  p.fft1_pa = (p.fft1_pa + 2 * N1) & (cfg.max_fft1n * 2 * N1 - 1);
  p.fft1_na = p.fft1_pa / (2 * N1);
  if (p.fft1_nm != cfg.max_fft1n - 1) p.fft1_nm++;
  lir_set_event(EVENT_TIMF2);
}
static void *wideband_thread(void *arg)
{
  (void)arg;
  int next = 0, oldest = 0, inflight = 0, out = 0;
  for (;;) {
    lir_await_event(EVENT_TIMF1);
    while (timf1_avail() >= timf1_blockbytes && !failed) {
      if (inflight == workers) { retire(oldest); oldest = (oldest + 1) % workers; inflight--; }
      /* the input thread may be a long way ahead (its copies do not wait for the stage calls): transforms handed out and not yet
         consumed stay within half the fft1 ring */
      AWAIT_SPACE(fft1_used() + inflight < cfg.max_fft1n / 2);
      job[next].inptr = p.timf1p_px; job[next].out = out; job[next].busy = 1;
      out = (out + 2 * N1) & (cfg.max_fft1n * 2 * N1 - 1);
      p.timf1p_px = (p.timf1p_px + timf1_blockbytes) & (cfg.timf1_bytes - 1);
      lir_set_event(EVENT_DO_FFT1B1 + next);
      next = (next + 1) % workers; inflight++;
    }
    if (failed || (input_done && timf1_avail() < timf1_blockbytes)) break;
  }
  while (inflight > 0 && !failed) { retire(oldest); oldest = (oldest + 1) % workers; inflight--; }
  for (int k = 0; k < workers; k++) { job[k].busy = -1; lir_set_event(EVENT_DO_FFT1B1 + k); }
  wide_done = 1;
  lir_set_event(EVENT_TIMF2);
  return NULL;
}

/* ---- THREAD_TIMF2 (timf2_routine, wcw.c:401-441) ---- */
static void *timf2_thread(void *arg)
{
  (void)arg;
  for (;;) {
    lir_await_event(EVENT_TIMF2);
    while (p.fft1_na != p.fft1_nb && !failed) {
      AWAIT_SPACE(timf2_used() < 2 * cfg.timf2pow_size);
      CHK(lrh_fft1_c(rx, P, 1));
      CHK(lrh_make_timf2(rx, P, 1));
      CHK(lrh_first_noise_blanker(rx, P));       /* per block: see the header comment */
      if (fft2_ready()) lir_set_event(EVENT_FFT2);
      space_freed();
    }
    if (failed || (wide_done && p.fft1_na == p.fft1_nb)) break;
  }
  timf2_done = 1;
  lir_set_event(EVENT_FFT2);
  return NULL;
}

/* ---- THREAD_SECOND_FFT (second_fft, wcw.c:250-304) ---- */
static void *fft2_thread(void *arg)
{
  (void)arg;
  for (;;) {
    lir_await_event(EVENT_FFT2);
    while (fft2_ready() && !failed) {
      AWAIT_SPACE(fft2_used() < cfg.max_fft2n / 2);
      CHK(lrh_make_fft2(rx, P, 1));
      lir_set_event(EVENT_FFT1_READY);
      space_freed();
    }
    if (failed || (timf2_done && !fft2_ready())) break;
  }
  fft2_done = 1;
  lir_set_event(EVENT_FFT1_READY);
  return NULL;
}

/* ---- THREAD_NARROWBAND_DSP (wcw.c:1240-1405) ---- */
static void *narrowband_thread(void *arg)
{
  (void)arg;
  for (;;) {
    lir_await_event(EVENT_FFT1_READY);
    while (p.fft2_nx != p.fft2_na && !failed) { CHK(lrh_fft2_mix1_fixed(rx, P, 1)); space_freed(); }
    if (failed || (fft2_done && p.fft2_nx == p.fft2_na)) break;
  }
  return NULL;
}

/* ---- the same calls from one thread: single-CPU order of wideband_dsp (wcw.c:1036-1118) ---- */
static void single_thread(void)
{
  for (long done = 0; done < (long)nblk_total * M1 && !failed; done += M1) {
    if (timf1p_pa == 0) CHK(lrh_timf1_write_wait(rx));
    lrh_synth_iq(&sig, done, M1, (int16_t *)(timf1_char + timf1p_pa));
    CHK(lrh_timf1_write_async(rx, timf1_char + timf1p_pa, timf1p_pa, M1 * 4));
    timf1p_pa = (timf1p_pa + M1 * 4) & (cfg.timf1_bytes - 1);
    while (timf1_avail() >= timf1_blockbytes && !failed) {
      CHK(lrh_fft1_b(rx, 0, p.timf1p_px, p.fft1_pa, 1));
      p.timf1p_px = (p.timf1p_px + timf1_blockbytes) & (cfg.timf1_bytes - 1);
      p.fft1_pa = (p.fft1_pa + 2 * N1) & (cfg.max_fft1n * 2 * N1 - 1);
      p.fft1_na = p.fft1_pa / (2 * N1);
      if (p.fft1_nm != cfg.max_fft1n - 1) p.fft1_nm++;
      CHK(lrh_fft1_c(rx, P, 1)); CHK(lrh_make_timf2(rx, P, 1)); CHK(lrh_first_noise_blanker(rx, P));
      while (fft2_ready() && !failed) { CHK(lrh_make_fft2(rx, P, 1)); CHK(lrh_fft2_mix1_fixed(rx, P, 1)); }
    }
  }
}

static void dump_ring(FILE *f, lrh_ring ring, size_t count, size_t esz)
{
  void *buf = calloc(count, esz);
  CHK(lrh_export(rx, ring, buf, 0, count));
  fwrite(buf, esz, count, f);
  free(buf);
}

int main(int argc, char **argv)
{
  if (argc < 4) { fprintf(stderr, "usage: %s threads(0|1) nblk out.bin [fft1_n fft2_n workers]\n", argv[0]); return 2; }
  const int threaded = atoi(argv[1]);
  nblk_total = atoi(argv[2]);
  const char *out = argv[3];
  const int fft1_n = argc > 4 ? atoi(argv[4]) : 12, fft2_n = argc > 5 ? atoi(argv[5]) : 13;
  workers = argc > 6 ? atoi(argv[6]) : 6;
  if (workers < 1) workers = 1;
  if (workers > 6) workers = 6;                 /* MAX_FFT1_THREADS, thrdef.h:107 */
  lrh_config_defaults(&cfg, fft1_n, fft2_n);
  cfg.fft1_gain = 27; cfg.max_batch = 4; cfg.max_fft1n = 64; cfg.max_fft2n = 16; cfg.timf1_bytes = 1 << 22;
  cfg.timf2pow_size = 32 << (fft1_n > fft2_n ? fft1_n : fft2_n);
  cfg.timf3_size = 1 << 16; cfg.blanker_min_points = 0;
  int rc = lrh_open(&cfg, &rx);
  if (rc) { fprintf(stderr, "lrh_open failed: %d (needs an MI355X / HIP device)\n", rc); return 2; }
  int i1, i2, nm, im, t3b;
  lrh_get_derived(rx, &i1, &i2, &nm, &im, &t3b);
  N1 = 1 << fft1_n; N2 = 1 << fft2_n; M1 = N1 - i1; M2 = N2 - i2; timf1_blockbytes = M1 * 4;
  lrh_ptrs_init(rx, P);
  lrh_synth_defaults(&sig, N1, 0);
  float *lim = calloc(N1, sizeof(float));
  for (int k = 0; k < sig.ncarriers; k++) if (sig.carrier_amp[k] >= 90) {
    int c = N1 / 2 + (int)(sig.carrier_bin[k] + (sig.carrier_bin[k] < 0 ? -0.5 : 0.5));
    for (int j = c - 3; j <= c + 3; j++) if (j >= 0 && j < N1) lim[j] = 1;
  }
  lrh_set_liminfo(rx, lim);
  lrh_set_mix1_selfreq(rx, 0.31 * N2 + 0.3);
  if (posix_memalign((void **)&timf1_char, 4096, cfg.timf1_bytes)) return 2;
  memset(timf1_char, 0, cfg.timf1_bytes);
  CHK(lrh_host_register(rx, timf1_char, cfg.timf1_bytes));
  for (int i = 0; i < NEVENTS; i++) { pthread_mutex_init(&ev_mutex[i], NULL); pthread_cond_init(&ev_cond[i], NULL); }

  if (threaded) {
    pthread_t th[5 + 6];
    int n = 0;
    pthread_create(&th[n++], NULL, narrowband_thread, NULL);
    pthread_create(&th[n++], NULL, fft2_thread, NULL);
    pthread_create(&th[n++], NULL, timf2_thread, NULL);
    for (int k = 0; k < workers; k++) pthread_create(&th[n++], NULL, fft1b_thread, (void *)(long)k);
    pthread_create(&th[n++], NULL, wideband_thread, NULL);
    pthread_create(&th[n++], NULL, input_thread, NULL);
    for (int i = 0; i < n; i++) pthread_join(th[i], NULL);
  } else single_thread();
  CHK(lrh_sync(rx));

  FILE *f = fopen(out, "wb");
  if (!f) { perror(out); return 2; }
  lrh_ptrs pe = *P;
  fwrite(&pe, sizeof pe, 1, f);
  lrh_blanker_state bs; CHK(lrh_get_blanker_state(rx, &bs));
  fwrite(&bs, sizeof bs, 1, f);
  dump_ring(f, LRH_RING_FFT1_FLOAT, (size_t)cfg.max_fft1n * 2 * N1, 4);
  dump_ring(f, LRH_RING_FFT1_SUMSQ, cfg.fft1_sumsq_bufsize, 4);
  dump_ring(f, LRH_RING_FFT1_SLOWSUM, N1, 4);
  dump_ring(f, LRH_RING_TIMF2_FLOAT, 4 * (size_t)cfg.timf2pow_size, 4);
  dump_ring(f, LRH_RING_TIMF2_PWR, cfg.timf2pow_size, 4);
  dump_ring(f, LRH_RING_FFT2_FLOAT, (size_t)cfg.max_fft2n * 2 * N2, 4);
  dump_ring(f, LRH_RING_FFT2_POWERSUM, N2, 4);
  dump_ring(f, LRH_RING_WG_WATERF, (size_t)cfg.wf_lines * cfg.wf_xpixels, 2);
  dump_ring(f, LRH_RING_TIMF3_FLOAT, cfg.timf3_size, 4);
  fclose(f);
  printf("%s: %d blocks of %d samples, fft1_na %d fft2_na %d timf3_pa %d, noise floor %d, cleared %d, %s\n", threaded ? "threads" : "single",
         nblk_total, M1, pe.fft1_na, pe.fft2_na, pe.timf3_pa, bs.timf2_noise_floor, bs.last_call_cleared, failed ? "FAILED" : "ok");
  lrh_timf1_write_wait(rx);
  lrh_host_unregister(rx, timf1_char);
  lrh_close(rx); free(timf1_char); free(lim);
  return failed ? 1 : 0;
}
