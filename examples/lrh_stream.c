/*
 * lrh_stream.c -- plain-C host driver over the C ABI (include/linrad_hip.h).
 *
 * Emulates what Linrad does around the hot path: an input thread's finish_rx_read (rxin.c:1143-1436)
 * makes new int16 IQ visible in timf1 and advances timf1p_pa; the wideband loop (wcw.c:1036-1118) then
 * runs fft1_b -> fft1_c -> make_timf2 -> first_noise_blanker -> make_fft2* -> fft2_mix1_fixed on every
 * complete block, advancing the same pointer variables Linrad keeps in globals (here: one lrh_ptrs).
 * Host code is C; the GPU is reached only through liblinrad_hip.so.
 *
 *   gcc -O2 -Iinclude examples/lrh_stream.c -Llinrad_amd -llinrad_hip -Wl,-rpath,$PWD/linrad_amd -lm -o lrh_stream
 *   ./lrh_stream [fft1_n fft2_n seconds_of_signal_at_2Msps]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "linrad_hip.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
  int fft1_n = argc > 1 ? atoi(argv[1]) : 13, fft2_n = argc > 2 ? atoi(argv[2]) : 15;
  double seconds = argc > 3 ? atof(argv[3]) : 2.0;
  const double fs = 2e6;                       /* BASELINE.json config 0 rate; pacing is "as fast as possible" */
  lrh_config cfg;
  lrh_config_defaults(&cfg, fft1_n, fft2_n);
  cfg.fft1_gain = 27; cfg.max_batch = 4; cfg.max_fft1n = 8; cfg.max_fft2n = 4;
  cfg.timf1_bytes = 1 << 22;
  lrh_ctx *rx = NULL;
  int rc = lrh_open(&cfg, &rx);
  if (rc) { fprintf(stderr, "lrh_open failed: %d (needs an MI355X / HIP device)\n", rc); return 2; }
  int i1, i2, nm, im, t3b;
  lrh_get_derived(rx, &i1, &i2, &nm, &im, &t3b);
  const int N1 = 1 << fft1_n, N2 = 1 << fft2_n, M1 = N1 - i1, M2 = N2 - i2;
  const int timf1_blockbytes = M1 * 4;          /* buf.c:601-617 */
  lrh_ptrs p; lrh_ptrs_init(rx, &p);
  lrh_synth sig; lrh_synth_defaults(&sig, N1, 0);
  float *lim = calloc(N1, sizeof(float));       /* control plane: route the strong carriers (sellim.c does this in Linrad) */
  for (int k = 0; k < sig.ncarriers; k++) if (sig.carrier_amp[k] >= 90) {
    int c = N1 / 2 + (int)(sig.carrier_bin[k] + (sig.carrier_bin[k] < 0 ? -0.5 : 0.5));
    for (int j = c - 3; j <= c + 3; j++) if (j >= 0 && j < N1) lim[j] = 1;
  }
  lrh_set_liminfo(rx, lim);
  lrh_set_mix1_selfreq(rx, 0.31 * N2 + 0.3);

  const long total = (long)(seconds * fs);
  int chunk = 4 * M1;                           /* what one "soundcard read" delivers */
  while ((cfg.timf1_bytes % (chunk * 4)) != 0) chunk /= 2;      /* a read never straddles the end of the ring here */
  /* Linrad's timf1 arena (buf.c:744-770): the input thread writes into it, the device ring mirrors it.  Page-locked once,
     like a maintainer would do after get_buffers(); the shim never frees it. */
  char *timf1_char = NULL;
  if (posix_memalign((void **)&timf1_char, 4096, cfg.timf1_bytes)) return 2;
  memset(timf1_char, 0, cfg.timf1_bytes);
  if ((rc = lrh_host_register(rx, timf1_char, cfg.timf1_bytes))) { fprintf(stderr, "lrh_host_register: %d\n", rc); return 2; }
  int timf1p_pa = 0;                            /* producer pointer, bytes */
  long done = 0, nfft2 = 0;
  double t0 = now();
  while (done < total) {
    /* ---- input thread: finish_rx_read ---- */
    /* back-pressure once per lap: every copy of the previous lap has left the arena before it is written again
       (a real producer also checks its distance to timf1p_px like rxin.c does) */
    if (timf1p_pa == 0 && (rc = lrh_timf1_write_wait(rx))) goto fail;
    lrh_synth_iq(&sig, done, chunk, (int16_t *)(timf1_char + timf1p_pa));
    if ((rc = lrh_timf1_write_async(rx, timf1_char + timf1p_pa, timf1p_pa, chunk * 4))) goto fail;   /* no host wait: the next fft1 waits on the device */
    timf1p_pa = (timf1p_pa + chunk * 4) & (cfg.timf1_bytes - 1);
    done += chunk;
    /* ---- wideband_dsp: one block at a time while a full block is available (wcw.c:940-1047) ---- */
    while (((timf1p_pa - p.timf1p_px + cfg.timf1_bytes) & (cfg.timf1_bytes - 1)) >= timf1_blockbytes) {
      if ((rc = lrh_fft1_b(rx, 0, p.timf1p_px, p.fft1_pa, 1))) goto fail;
      p.timf1p_px = (p.timf1p_px + timf1_blockbytes) & (cfg.timf1_bytes - 1);
      p.fft1_pa = (p.fft1_pa + 2 * N1) & (cfg.max_fft1n * 2 * N1 - 1);
      p.fft1_na = p.fft1_pa / (2 * N1);
      if ((rc = lrh_fft1_c(rx, &p, 1)) || (rc = lrh_make_timf2(rx, &p, 1)) || (rc = lrh_first_noise_blanker(rx, &p))) goto fail;
      while (((p.timf2_pn2 - p.timf2_px + 4 * cfg.timf2pow_size) & (4 * cfg.timf2pow_size - 1)) >= 4 * N2) {   /* wcw.c:265 */
        if ((rc = lrh_make_fft2(rx, &p, 1)) || (rc = lrh_fft2_mix1_fixed(rx, &p, 1))) goto fail;
        nfft2++;
      }
    }
  }
  lrh_sync(rx);
  double dt = now() - t0;
  lrh_blanker_state bs; lrh_get_blanker_state(rx, &bs);
  float t3[8]; lrh_export(rx, LRH_RING_TIMF3_FLOAT, t3, 0, 8);
  printf("fft1_size %d fft2_size %d (new points %d / %d) mix1 size %d: %ld samples in %.3f s = %.2f Msamples/s, %ld fft2 transforms\n",
         N1, N2, M1, M2, nm, done, dt, done / dt / 1e6, nfft2);
  printf("blanker: noise floor %d limit %u cleared %.2f %%; timf3[0..3] = %g %g %g %g\n", bs.timf2_noise_floor, bs.stupid_bln_limit,
         bs.stupid_blanker_rate, t3[0], t3[1], t3[2], t3[3]);
  lrh_timf1_write_wait(rx);
  lrh_host_unregister(rx, timf1_char);          /* before the arena is freed (free_buffers, buf.c:2105) */
  lrh_close(rx); free(timf1_char); free(lim);
  return 0;
fail:
  fprintf(stderr, "stage failed rc=%d: %s\n", rc, lrh_last_error(rx));
  lrh_close(rx);
  return 1;
}
