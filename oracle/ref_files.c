/*
 * ref_files.c -- TEST INFRASTRUCTURE ONLY (build container): the COMPILED REFERENCE's own file readers, called head-less.
 *
 *   ref_files raw <file.raw>   open_savefile (modesub.c:606-733) parses the header of a Linrad .raw recording
 *   ref_files wav <file.wav>   init_wavread (modesub.c:1022-1347) parses a .wav header (fmt, rcvr / auxi chunks)
 *
 * Prints what the reference left in its globals as one JSON line; tests/golden/make_rawhdr_golden.py feeds it the files
 * linrad_amd/rawfile.py and wavfile.py write and commits the answers (tests/golden/filehdr.npz), which pin the readers in
 * linrad_amd/ (tests/test_rawfile_cpu.py, tests/test_wavfile_cpu.py).  The reference's screen / keyboard calls on the way are empty here.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "globdef.h"
#include "uidef.h"
#include "fft1def.h"
#include "screendef.h"
#include "thrdef.h"
#include "sdrdef.h"

int open_savefile(char *s);
int init_wavread(int sel_file);
extern int remember_proprietery_chunk[2];
extern FILE *save_rd_file;
extern double diskread_time;
extern int freq_from_file, diskread_flag, save_init_flag;
extern unsigned char perseus_hdr[];      /* RCVR (modesub.c:113-127, a type of that file): chunkID[4], chunkSize, nCenterFrequencyHz at 8, SamplingRateIdx at 12, time_t timeStart at 16 */

/* what the two readers call on their way (GUI, error reporting): nothing to show here */
void lirerr(int e) { fprintf(stderr, "lirerr(%d)\n", e); lir_errcod = e; }
void lir_text(int x, int y, char *s) { fprintf(stderr, "lir_text: %s\n", s); }
void await_keyboard(void) {}
void await_processed_keyboard(void) {}
static void no_screen(void) {}
void (*clear_screen)(void) = no_screen;           /* (function pointers in the reference: one per graphics back end, lscreen.h:16-17) */
void settextcolor(unsigned char c) {}
void help_message(int n) { fprintf(stderr, "help_message(%d)\n", n); }
void lir_sched_yield(void) {}
void (*lir_refresh_screen)(void) = no_screen;
void lir_sleep(int us) {}

#include <execinfo.h>
#include <signal.h>
static void on_segv(int sig) { void *bt[32]; int n = backtrace(bt, 32); backtrace_symbols_fd(bt, n, 2); _exit(3); }
int main(int argc, char **argv)
{
  signal(SIGSEGV, on_segv);
  if (argc < 3) return 2;
  int rc;
  memset(&ui, 0, sizeof ui); memset(&fg, 0, sizeof fg);
  if (!strcmp(argv[1], "raw")) rc = open_savefile(argv[2]);
  else {
    char tmpl[] = "/tmp/ref_files.XXXXXX";
    if (!mkdtemp(tmpl) || chdir(tmpl)) return 2;
    FILE *f = fopen("adwav", "w"); fprintf(f, "%s\n", argv[2]); fclose(f);      /* init_wavread takes the file name from ./adwav */
    rc = init_wavread(0);
    unlink("adwav"); if (chdir("/")) return 2; rmdir(tmpl);
  }
  long at = save_rd_file ? ftell(save_rd_file) : -1;
  printf("{\"rc\": %d, \"rx_input_mode\": %d, \"rx_rf_channels\": %d, \"rx_ad_channels\": %d, \"rx_ad_speed\": %d, \"save_init_flag\": %d, "
         "\"remember0\": %d, \"remember1\": %d, \"diskread_time\": %.17g, \"passband_center\": %.17g, \"passband_direction\": %d, \"fft1_direction\": %d, "
         "\"freq_from_file\": %d, \"diskread_flag\": %d, \"perseus_center_hz\": %u, \"perseus_time\": %lld, \"file_pos\": %ld}\n",
         rc, ui.rx_input_mode, ui.rx_rf_channels, ui.rx_ad_channels, ui.rx_ad_speed, save_init_flag, remember_proprietery_chunk[0], remember_proprietery_chunk[1],
         diskread_time, fg.passband_center, fg.passband_direction, fft1_direction, freq_from_file, diskread_flag, *(unsigned *)(perseus_hdr + 8),
         *(long long *)(perseus_hdr + 16), at);
  return 0;
}
