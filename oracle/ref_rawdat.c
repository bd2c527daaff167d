/*
 * ref_rawdat.c -- TEST INFRASTRUCTURE ONLY.  Drives the compiled reference's 18-bit raw-file packing
 * (csplit.c:20-118: expand_rawdat, compress_rawdat_disk) head-less, for the golden vectors of
 * tests/golden/rawdat_18bit.npz.  Built by `make -C oracle ref` from the sources under /root/reference.
 *
 *   ref_rawdat expand   in=<packed bytes> out=<int32 ring image>  pa=<ring byte position> ring_log2=<n>
 *   ref_rawdat compress in=<int32 samples> out=<packed bytes>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern char *rawsave_tmp, *rawsave_tmp_disk;
extern char *timf1_char;
extern int rx_read_bytes, timf1p_pa, timf1p_pc_disk;
void expand_rawdat(void);
void compress_rawdat_disk(void);
void lirerr(int e) { fprintf(stderr, "lirerr %d\n", e); exit(3); }

static const char *arg(int argc, char **argv, const char *k, const char *d)
{
  size_t n = strlen(k);
  for (int i = 2; i < argc; i++) if (!strncmp(argv[i], k, n) && argv[i][n] == '=') return argv[i] + n + 1;
  return d;
}

int main(int argc, char **argv)
{
  if (argc < 4) return 2;
  FILE *fi = fopen(arg(argc, argv, "in", ""), "rb"); if (!fi) return 2;
  fseek(fi, 0, SEEK_END); long n = ftell(fi); fseek(fi, 0, SEEK_SET);
  char *in = malloc(n + 64); if (fread(in, 1, n, fi) != (size_t)n) return 2; fclose(fi);
  FILE *fo = fopen(arg(argc, argv, "out", "rawdat.bin"), "wb"); if (!fo) return 2;
  if (!strcmp(argv[1], "expand")) {
    int ring = 1 << atoi(arg(argc, argv, "ring_log2", "16"));
    timf1p_pa = atoi(arg(argc, argv, "pa", "0"));
    rx_read_bytes = (int)(n / 9) * 16;                 /* expanded bytes of this read (csplit.c:34) */
    if (timf1p_pa + rx_read_bytes > ring) return 2;    /* the reference never wraps inside a read */
    timf1_char = calloc(ring, 1); rawsave_tmp = in;
    expand_rawdat();
    fwrite(timf1_char, 1, ring, fo);
  } else {
    rx_read_bytes = (int)n; timf1p_pc_disk = 0; timf1_char = in;
    rawsave_tmp_disk = calloc(n, 1);
    compress_rawdat_disk();
    fwrite(rawsave_tmp_disk, 1, n / 16 * 9, fo);
  }
  fclose(fo);
  return 0;
}
