// lrh_kernels.hip.h -- argument blocks shared by the kernels (lrh_kernels.hip) and the host side (lrh_host.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lrh {

#define LRH_FFT2_FUSED_MAXLOG 13   // largest fft2 size whose register file also holds the 16 power accumulators
#define LRH_PH_CHUNK 64      // mix1 phases: the host uploads every 64th value of its float recursion, the kernel replays the rest


// ---- fft1_b (+ filter correction of fft1_c) ----
struct Fft1Args {
  const void *timf1;        // interleaved I,Q ring: int16 pairs, or int32 pairs when dword != 0 (DWORD_INPUT)
  int dword;                // sample container: 0 int16, 1 int32 (fft1.c:420 / :526)
  int shift_i, shift_q;     // ui.sample_shift as per-component sample offsets: I from n+shift_i, Q from n+shift_q (fft1.c:470-482)
  int ring_mask;            // in complex samples
  int p0_first;             // first sample of transform 0 = timf1p_ref/4 - I1 (fft1.c:421-426)
  int step;                 // new samples per transform (M1)
  const float *window;      // natural order, N1 (all ones when sinpow = 0); real input: 2*N1, one value per real sample
  int real;                 // real samples (fft1_reherm_dit_one): pairs (x[2n], x[2n+1]) packed as one complex point, bare transform out
  const float2 *filtercorr; // N1
  const float2 *tw;         // exp(-2 pi j m/N1)
  float2 *out;              // fft1_float ring
  int first_nb, nb_mask;
  int direction;
  int xcd;                  // XCD-aware block order on/off
  int batch;                // transforms in this launch (workgroups are persistent)
  int chan_count, chan_index; // frame layout {I0,Q0,I1,Q1,...}: sample s of this channel is short2 s*chan_count+chan_index
  unsigned long long *stamps; // diagnostics (LRH_STAMP=1): s_memtime of workgroups 0 and 128 at phase boundaries, else null
  int spare_cus;            // compute units left free for side-stream kernels (see persistent_grid)
};
#define LRH_STAMPS_PER_WG 64

// ---- I/Q mirror-image cancellation + filter correction (fft1.c:3598-3658, 4119-4127) ----
struct FoldcorrArgs { float2 *spec; int first_nb, nb_mask; int n; const float2 *foldcorr, *filtercorr; int direction; };

// ---- fft1_c power sums ----
// Averaging groups are derived in-kernel: group g covers transforms [g*avg - c0, ...) of the batch (the first one
// continues a group started by an earlier call when c0 > 0) and lands in sumsq block (pa0 + g*n) & mask.
struct SumsqArgs {
  const float2 *spec; int nb_mask; int n; float *sumsq; int sumsq_mask;
  int first_nb; int batch; int avg; int c0; int pa0;
};
struct SlowsumArgs {
  const float *sumsq; float *slowsum; int n; int bufsize; int avg2;
  int nupd; int pa0; int recalc0; int step;
  int e0, recalc_e0;        // the search for a bin's last refresh starts at update e0 (recalc pointer there), see k_slowsum
};

// ---- correlation spectrum of two coupled channels (fft1_corrsum, fft1_slowcorr, fft1_slowcorr_tot) ----
struct CorrArgs {
  const float2 *x, *y; int n, batch;            // the batch's transforms of channel 0 and channel 1 (LRH_X_SPEC slots), [batch][n]
  float2 *corrsum; int sumsq_mask;              // ring [sumsq_bufsize] like fft1_sumsq, two floats per bin
  int avg, c0, pa0;                             // fft_avg1num, fft1_sumsq_counter and fft1_sumsq_pa before the batch (SumsqArgs)
  float2 *slowcorr; double2 *tot; int bufsize, avg2, nupd, recalc0, step;   // update_fft1_slowsum's walk (SlowsumArgs)
};
hipError_t launch_corrsum(const CorrArgs &a, hipStream_t st);

// ---- make_timf2 ----
struct Timf2Args {
  const float2 *spec; int first_nb, nb_mask;
  const unsigned int *pack_cur, *pack_prev;   // packed weak flags for the batch / for the transform before it
  const float2 *tw;
  float2 *timf2w, *timf2s; float *pwr; int pa_first; int mask; int step;   // planar weak / strong rings
  int mode;                 // 0: no window, 1: sin^2 overlap-add, 2: centre part x inverted window
  int ia;                   // interleave/2 for mode 2
  const float *invwin;      // natural order N1 (mode 2)
  float ampfac;
  int xcd;
  int batch;                // transforms in this launch (set by launch_timf2)
  // fused fft1_c power sums (set by launch_timf2 when it is handed a SumsqArgs): ring, pieces of groups that straddle
  // workgroup runs [grid][2][N], transforms per workgroup
  float *ss_ring, *ss_part; int ss_mask, ss_avg, ss_c0, ss_pa0, ss_run;
  int ss_split;             // a launch of ONE transform: two workgroups, one per stream (each adds its own stream's bins to the sums)
  int spare_cus;            // compute units left free for side-stream kernels (see persistent_grid)
  int sd_kmax;              // strong-only pass: the launch has been offered to k_timf2_sd first, which takes it when no more than this many bins
                            // are routed strong (0: not offered) -- the transform kernel then returns at once (set by launch_timf2_strong)
};

// ---- fft1_b + fft1_c's sums + the weak stream of make_timf2 in one kernel (k_fft1w, fft1_size 16384) ----
struct Fft1wArgs {
  const void *timf1; int ring_mask, p0_first, step, chan_count, chan_index;   // as Fft1Args (int16 I/Q, no skew)
  const float *window; const float2 *filtercorr, *tw;
  const float2 *filtercorr_v;                   // k_fft1v: the same table in the order its threads hold the bins, [j][thread] (lrh_host.hip upload_filtercorr)
  float2 *spec; int first_nb, nb_mask;          // fft1 ring: strong bins, or every bin with keep_spec
  int keep_spec;
  const unsigned int *pack_cur, *pack_prev;     // as Timf2Args
  float2 *timf2w; float *pwr; int pa_first, mask; float ampfac;
  int have_prev;                                // 0: the stream starts with this launch (nothing to overlap the first transform with)
  float *ss_ring, *ss_part; int ss_mask, ss_avg, ss_c0, ss_pa0;   // as Timf2Args (k_sumsq_join finishes split groups)
  int batch, run;                               // run: consecutive transforms per workgroup (set by launch_fft1w)
  int max_wg;                                   // workgroups the scratch of split groups ss_part has room for
  unsigned long long *stamps;                   // diagnostics (LRH_FFT1V_EXP=2): 2 x 32 shader-clock stamps
  int stagger;                                  // start delay per workgroup, (blockIdx & 15) x stagger x 2048 cycles (set by launch_fft1v)
  int spare_cus;
  float real_peak;                                // k_fft1v<REAL>: the half window's peak value h[N] (make_window mode 2; the table on the device ends at h[N-1])
};
hipError_t launch_phase_expand(const float2 *h_inc, const float2 *h_st, const int *h_point, float2 *d_inc, float2 *d_start, int *d_point, int batch, int nchunks, int chunk, hipStream_t st);
hipError_t launch_copy_words(const unsigned int *src, unsigned int *dst, size_t n, hipStream_t st);   // n 32-bit words; src may be page-locked host memory
hipError_t launch_fft1w(const Fft1wArgs &a, hipStream_t st, int *run);
hipError_t launch_fft1v(int log2n, bool dword, bool real, const Fft1wArgs &a, hipStream_t st, int *run);
hipError_t launch_timf2_strong(int log2n, const Timf2Args &a, int batch, hipStream_t st);
constexpr int LRH_SD_KMAX = 128;           // strong bins k_timf2_sd (lrh_timf2_sd.hip) takes; more: the transform kernel
hipError_t launch_timf2_sd(int log2n, const Timf2Args &a, hipStream_t st);   // a.batch set; returns at once on the device when too many bins are strong

// fft1_size 32768 (the reference's maximum with the second fft on, buf.c:335): one block no longer fits a workgroup's LDS, so
// fft1 and timf2 take the four-step form of the large fft2 (column transforms, step twiddle, row transforms through an HBM scratch)
struct Fft1BigArgs {
  Fft1Args f;                                   // ring, window, filter correction, output ring, direction (int16 / int32 I/Q, no skew, no real input)
  const float2 *tw_a, *tw_b, *tw_big;           // forward tables of size NA, NB, N1
  float2 *scratch;                              // [batch][NB][NA]
  int run;                                      // consecutive blocks per workgroup of the column step (set by launch_fft1_big)
};
struct Timf2BigArgs {
  Timf2Args t;                                  // spectra, rings, ampfac; mode 1 (sin^2 overlap) only; pack_cur / pack_prev are dense bit words here
  const float2 *tw_a, *tw_b, *tw_big;
  float2 *scratch;                              // [batch][2 streams][NB][NA]
};
hipError_t launch_fft1_big(int log2n, const Fft1BigArgs &a, int batch, hipStream_t st, int steps = 3);   // steps: 1 column step, 2 row step, 3 both
// fft1_size 32768 inside lrh_wideband_dsp: the row step of fft1, fft1_c's sums and the column step of both timf2 streams as one kernel
// (the spectrum is written once and never read back; k_timf2_rows follows)
struct Fft1rT2cArgs { Fft1BigArgs f1; Timf2BigArgs t2; SumsqArgs ss; int groups_per_run; int keep_spec; };   // keep_spec 0: only the launch's last transform reaches the fft1 ring (the next launch's predecessor)
hipError_t launch_fft1r_t2c(const Fft1rT2cArgs &a, int batch, hipStream_t st);
hipError_t launch_timf2_big(int log2n, const Timf2BigArgs &a, int batch, hipStream_t st);

// ---- blanker ----
struct BlankState {          // device resident; mirrors lrh_blanker_state + scratch
  int noise_floor; unsigned int limit;
  float despiked_pwr[2]; float despiked_pwrinc[2];
  float stupid_rate; int cleared_acc; int last_cleared; int slow_calls;
  int call_cleared;          // scratch: cleared_points of the running call (atomic)
  int need_slow;             // scratch: a lane found no clean restart point
  // linear ("clever") blanker
  unsigned int clever_limit; float clever_rate; int fitted_acc; int last_fitted; int last_rejected;
  int clever_out[4];         // what k_clever hands the host: ring position where the scan stopped (pf), pulses fitted, pulses rejected, [3] != 0: extents collided -- the host issues the one-wave replay
  int clever_serial_calls;   // calls that fell back to the one-wave replay
  float amp_factor;          // liminfo_amplitude_factor: the limiter kernels keep it current, k_clever scales its reference pulse with it
  int need_slow2;            // scratch: ... and none within the long look-back of the second scan either: the serial walk takes the call
};
#define LRH_BLN_WTILE 4096
struct BlankArgs {
  float *pwr; float2 *timf2w; unsigned int *mask_bits; int mask;   // mask: timf2pow_mask
  int pbeg, total;          // positions pbeg+1 .. pbeg+total are scanned
  int clr1, clr2;
  int mode;                 // stupid_bln_mode
  BlankState *st;
  float *partials; int npartials; int nremoved;   // doubles: [npartials] sums, then [nremoved] removed power
  int *counts; int ncounts;  // cleared samples per scan tile (summed by k_blank_update: 2048 atomics on one word cost 24 us)
  // statistics / update (blank1.c:1472-1601)
  int m; int nstat;         // m: points counted; nstat: samples in the every-4th sum
  int blanker_points;       // timf2_blanker_points after adding m
  int do_update; float lowlevel_fraction; int interval; int avgnum; float factor;
  int debug;                // tuning experiments only (LRH_BLN_DEBUG), 0 in normal operation
  // two coupled RF channels, one per context (blank1.c:1017, 1236-1300, 1510-1545, 1570): `pwr` is then the exchanged
  // channel power sum the decisions are taken on, `own` this channel's own power ring (cleared alongside, source of the
  // per-channel noise statistic), xstat the two-float exchange buffer; phase 1 stops after the own statistic,
  // phase 2 resumes with both channels' values (phase 0: single channel, everything in one go)
  int chans; float *own; float *xstat; int own_slot; int phase;
  float4 *tiles;            // per-tile run summaries of the long-run replay (k_blank_runs_pre / k_blank_runs)
  // tile-parallel form of the calibrated blanker's serial walk (k_blank_walk_*): one 64-bit word per 64 sequence positions ("the
  // speculative walk was inside a run or a guard here"), and per tile of LRH_BLN_WTILE positions the state a walk leaves it in
  // [0 .. nwt) and the state the true walk enters it in [nwt .. 2 nwt)
  unsigned long long *wbusy; int4 *wstate; int nwt;
  // linear blanker ran before this call's stupid pass: the every-4th-sample statistic ends at timf2p_fit (blank1.c:1458-1461),
  // short of the scanned span, so it is summed after the clearing (post_stats); fitted / rejected pulses for the bookkeeping
  int post_stats; int fitted, rejected; int clever_mode; float clever_factor;
};

// ---- linear ("clever") blanker: pulse search / fit / subtract of blank1.c:765-1003 ----
struct CleverArgs {
  float *pwr; float2 *timf2w; unsigned char *flag; unsigned long long *cand; int mask;    // mask: timf2pow_mask
  int pbeg, total;          // span: ring positions pbeg .. pbeg+total (= blnk_pend)
  int R, pwid, rs, largest; // blnfit_range, blanker_pulsewidth, refpul_size, largest_blnfit
  float amp_factor;         // liminfo_amplitude_factor
  const float *amp_dev;     // where the search reads it on the device (NULL: BlankState::amp_factor)
  const float *refpulse, *phasefunc; const int *pulindex;
  int bln_size[7]; float bln_rest[7], bln_avgmax[7];
  BlankState *st;
  // region-parallel replay: pulses further apart than `gap` samples cannot see each other, so the span is cut into regions at the
  // quiet stretches of the candidate bits and one wave replays each region; every wave reports the extent of samples it looked at
  // or changed, k_clever_check verifies that neighbouring extents stay apart, and when they do not (a monotone run of hundreds of
  // samples) the span is restored from the backup and replayed by one wave in the reference's order
  int gap; int *reg_start; int max_regions; int *reg_ext;     // [max_regions] first candidate offset; [2*max_regions] lo, hi
  int *reg_dbg;             // diagnostics (LRH_CLEVER_DEBUG), or NULL: [2*max_regions] candidates handled, wall clock ticks (100 MHz) per region
  int *reg_ctl;             // [0] number of regions, [1] violation flag, [2] pf of the last region, [3] undo log entries, [8 ..] per-block region counts (k_clever_regions)
  // undo log instead of a copy of the span: the first wave to rewrite a ring sample in a call (bit in `logged`, taken with an atomic or)
  // appends the sample's values as it staged them -- read before its own atomic, hence before any other wave's write, which comes after
  // that wave's (losing) atomic -- so a replay finds the call's original samples whatever order the waves ran in.  One entry per sample
  // at most: capacity = span + margins, never exceeded.  reg_ctl[3] counts the entries.
  unsigned long long *logged; int *bk_pos; float *bk_pwr; float2 *bk_tf; int bk_margin;
  int phase;                // k_clever_prep: 0 first pass, 1 after a violation (k_clever_restore has put the samples back); k_clever: 0 parallel, 1 serial if violated
  int force_serial;         // tests: report a violation whatever the extents say
  // two coupled channels (blank1.c:984-992): pwr is the ring of summed powers, timf2w the own channel (number `chan`), timf2y the
  // partner's samples of the exchanged span in ring places, pwr_own the own channel's power ring (bk_ty / bk_pwo: their backups).
  int twochan, chan; float2 *timf2y; float *pwr_own; float2 *bk_ty; float *bk_pwo;
};
hipError_t launch_clever(const CleverArgs &a, hipStream_t st, int parts = 3);   // parts: 1 candidate bits + regions, 2 region replay + check, 4 restore + one-wave replay (only after clever_out[3] came back set)

// ---- fft2 ----
struct Fft2Args {
  const float2 *timf2w, *timf2s; int mask; int px_first; int step;
  const float *window; const float2 *tw;
  float2 *out; float *power; int first_na, na_mask;
  int xcd;
  int batch, run;           // transforms in this launch; consecutive transforms per workgroup (set by launch_fft2)
  // fused power sums (fft2.c:655-670): with ps_avgnum > 0 a workgroup takes one waterfall averaging group instead of
  // a fixed run, keeps sum |X|^2 in registers and writes the group line; `power` may then be null
  const float *ps_in; float *ps_out; float *wf_scratch; int ps_counter; int ps_avgnum;
  int keep_lo, keep_hi;     // bins [keep_lo, keep_hi) of a transform reach the fft2 ring (cfg.fft2_float_sparse: the band mix1 cuts out); 0, N: all
};
// four-step fft2 (N2 > 16384): N2 = NA*NB, column transforms of length NA, twiddle, row transforms of length NB
struct Fft2BigArgs {
  const float2 *timf2w, *timf2s; int mask; int px_first; int step;
  const float *window;
  const float2 *tw_a, *tw_b, *tw_big;         // forward tables of size NA, NB and N2
  float2 *scratch;                            // [batch][NB][NA]
  float2 *out; float *power; int first_na, na_mask;
  // fused power sums as in Fft2Args (ps_avgnum > 0): the rows kernel takes one averaging group per blockIdx.y
  const float *ps_in; float *ps_out; float *wf_scratch; int ps_counter; int ps_avgnum; int batch;
  int run;                  // consecutive transforms per workgroup of the column step (set by launch_fft2_big)
  int keep_lo, keep_hi;     // as Fft2Args
};
struct Powersum2Args {
  const float *power; int na_mask; int first_na; int count; int n;
  const float *powersum_in; float *powersum_out; float *wf_scratch; int counter; int avgnum;
};
// real input, second half of fft1_reherm_dit_one (fft1_re.c:96-131): the N1-point transform of the packed pairs is split into
// the 2*N1-point real transform, laid out like the reference does, and multiplied by the filter correction of fft1_c
struct RealSplitArgs { float2 *spec; int first_nb, nb_mask, n; const float2 *filtercorr; int direction; };
// two coupled channels: cross products of the channels' fft2 bins (TWOCHAN_POWER), sums per waterfall group
struct XyArgs {
  const float2 *x, *y;        // [batch][n] bins of channel 0 / channel 1: the two slots of the exchange buffer, or for the own channel the span of the fft2 ring itself
  float4 *xypower;            // ring [na_mask+1][n] {x2, y2, im_xy, re_xy}
  int first_na, na_mask, n, batch;
  const float4 *sum_in; float4 *sum_out;   // fft2_xysum, ping-pong like fft2_powersum
  float *lines;               // [group][n] polarisation-independent power of each completed group, for k_waterfall
  int counter, avgnum;
};
struct WaterfallArgs {
  const float *ps; const float *yfac; const int *itab; int16_t *line;
  int npix; int first; int siz; int hx; int hp;
  int ptr0; int wf_size; int line_stride;   // line l goes to wg_waterf[(ptr0 - l*npix) mod wf_size], reads ps + l*line_stride
};

// ---- mix1 ----
struct Mix1Args {
  const float2 *fft2; int n2; int first_nx, nx_mask;   // source ring: fft2_float (or fft1_float when the second fft is off)
  const float *fqwin; const float2 *tw;
  float2 *scratch;           // [batch][Nm] raw back transforms
  int point; int lim_hi;     // bins >= lim_hi are zeroed (mix1.c:957-958), bins < 0 zeroed
  const int *points;         // per-transform mix1_point (AFC variants, mix1.c:880-882); null: `point` for all
  int nm;
};
struct Mix1OutArgs {
  const float2 *scratch;
  const float2 *ph_inc;      // [batch] per-sample phase increments {new, old} of each transform
  const float2 *ph_start;    // [batch][nchunks] phases {new, old} at samples 0, 64, 128, ... (LRH_PH_CHUNK)
  int nchunks;
  float2 *timf3; int mask2;  // timf3 mask in complex samples
  int pa_first; int block;   // in complex samples
  int nm; int overlap; int selected;
  int rotate;                // 1: mix1 (phase rotation, mix1.c:172-186); 0: mix2 plain overlap-add (mix2.c:158-168)
  // crossover-window mix1 (mix1.c:196-262): xover = mix1.crossover_points (> 0 selects it), im = interleave_points,
  // win = inverted window (mode 3, nm/2+1 values), sin2win / cos2win = crossover functions (prepare_mixer, buf.c:95-109)
  int xover, im; const float *win, *sin2win, *cos2win;
};

// ---- fft3 / mix2 (fft3.c:240-283, mix2.c:145-176) ----
struct Fft3Args {
  const float2 *timf3; int mask; int px_first; int step;
  const float *window; const float2 *tw;
  float2 *out; int first_slot, slot_mask;
};
struct Mix2Args {
  const float2 *fft3; int n3; int first_slot, slot_mask;
  const float *filt; const float2 *tw; float2 *scratch; int nm;
  const float2 *pol;          // two coupled channels: [batch][nm] polarisation-combined bins standing in for the spectrum; null: fft3
};
// the search for new spurs: make_fft2's sums over 3 spur_speknum power rows (fft2.c:673-699) and spursearch_spectrum_cleanup (spursub.c:40-175)
struct SpurSearchArgs {
  float *sum, *spec, *mins;             // spursearch_powersum, spursearch_spectrum (fft2_size floats each), minima of the groups of 32 bins
  const float2 *z;                      // the transform's spectrum behind the spur subtraction: the power row is |z|^2 (the fused fft2 kernels keep sums only)
  int first, last, mode;                // mode 0: sum = row, 1: sum += row, 2: spec = sum + row (then the cleanup)
  const float *spectra;                 // spur_spectra [256][8] (init_spur_spectra)
  double noise_factor, thr_factor;      // 10^(0.7 / sqrt(3 spur_speknum)), 10^(1.5 / sqrt(3 spur_speknum))
  float *out;                           // [0] spur_search_threshold, [1] the noise floor
};
hipError_t launch_spur_search_row(const SpurSearchArgs &a, hipStream_t st);
hipError_t launch_spur_search_cleanup(const SpurSearchArgs &a, hipStream_t st);
// bg.mixer_mode = 2: FIR decimator on timf3 (mix2.c:217-246)
struct Mix2FirArgs {
  const float2 *timf3; int mask;        // complex samples
  int py_first, step;                   // timf3_py of the first transform (complex samples), fft3_new_points
  int n3, m3, nm2new, resamp;           // fft3_size, fft3_new_points, mix2.new_points, fft3_size / mix2.size
  const float *fir; int pts;
  float2 *baseb; int bmask, pa_first;
};
hipError_t launch_mix2_fir(const Mix2FirArgs &a, int batch, hipStream_t st);
// own channel's share of the polarisation sums A and B (mix2.c:340-343): w_a, w_b complex weights of this channel
struct PolArgs {
  const float2 *fft3; int n3; int first_slot, slot_mask; int nm, batch;
  float2 wa, wb; float2 *out;   // out [2][batch][nm]
};

// ---- compute_timf2_powersum (wcw.c:80-138) ----
struct BlockpowerArgs { const float2 *timf2w; int mask; int first; int block; float *out; int out_mask; int out_first; };

}  // namespace lrh

// ---- selective limiter (fft1_update_liminfo + selfreq_liminfo, sellim.c:738-1157, 38-157) ----
namespace lrh {
struct SellimState { int sumsq_tot, sel_ia, sel_ib, low; };   // device resident between calls (the reference's globals)
struct SellimArgs {
  const float *sumsq;       // the fft1_sumsq block at fft1_sumsq_pa
  const float *slowsum, *yfac;
  float *liminfo, *old_liminfo, *tmp;    // N floats each (tmp: fftt_tmp, kept between calls like the reference's)
  unsigned char *wait;      // liminfo_wait
  unsigned int *pack;       // routing words k_timf2 consumes
  SellimState *st;
  int n, n2, avg1, r0;      // fft1_size, fft2_size, wg.fft_avg1num, first-pass radix of the back transform (pack layout)
  int maxlevel, spek_avgnum; float blocktime, ston;
  int par2, par3, par4, par5, par6, par7, par8, group_points, first_point, last_point, first_inband, last_inband, bw_fftxpts, ston_scale;
  double selfreq; float points_per_hz; int second_fft;
  // liminfo_amplitude_factor (selfreq_liminfo, sellim.c:108-155) goes to the blanker's device state; calibrated form with `desired`
  BlankState *bst; const float *desired; float desired_totsum;
  // fft2_update_liminfo (k_sellim2): summed fft2 power spectrum, hg.blanker_ston_fft2, seconds per fft2 transform, waterfall_avgnum
  const float *powersum2; float ston2, blocktime2; int wf_avgnum;
  int debug;                // LRH_SELLIM_DEBUG=1: thread 0 prints the phase times (100 MHz ticks)
  float *big_b, *big_g;     // fft1_size 32768: the table (n floats) and the group minima (n/4 + 8 floats) in global memory
  int par1;                 // hg.sellim_par1 of fft2_update_liminfo: 2, 1 (k_sellim2_regions) or 0 (k_sellim2_median)
  float *reg_noise; int *reg_first, *reg_len;   // par1 = 1: the region list, n / group_points + 8 entries each, kept between calls
};
hipError_t launch_sellim(const SellimArgs &a, hipStream_t st);
hipError_t launch_sellim2(const SellimArgs &a, hipStream_t st);
}

// ---- spur subtraction (eliminate_spurs, spur.c:36-494) ----
namespace lrh {
struct SpurArgs {
  float2 *fft2; int n2, first_na, na_mask, batch;       // fft2_float ring, transforms first_na .. first_na + batch - 1
  int nspurs, speknum, avgnum, numsub;
  float freq_factor, max_d2, minston, weiold, weinew, linefit;
  const float *spectra;                                  // [256][8] reference line shapes
  void *spurs;                                           // lrh_spur [nspurs]
  float *table, *signal; int *ind;                       // per spur: [maxn][7][2], [maxn][2], [maxn]
  int *touched;                                          // per spur [2]: first and one-past-last fft2 bin the batch's subtractions touched
};
hipError_t launch_spur(const SpurArgs &a, hipStream_t st);
hipError_t launch_spur_acquire(const SpurArgs &a, int pnt, int *result, hipStream_t st);   // a.nspurs = index of the new spur, a.first_na = ffts_na
// power sums of the touched bins redone from the cleaned spectra (group arithmetic of Powersum2Args)
struct SpurPatchArgs {
  const float2 *fft2; int n, first_na, na_mask, count, counter, avgnum;
  const int *touched; const float *powersum_in; float *powersum_out, *wf_scratch;
};
hipError_t launch_spur_patch(const SpurPatchArgs &a, int nspurs, int ngroups, hipStream_t st);
}
