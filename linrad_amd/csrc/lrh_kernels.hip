// lrh_kernels.hip -- hand-written HIP kernels (gfx950) for Linrad's wideband hot path.
//
//   k_fft1       fft1_b (fft1.c:413-447, fft0.c:161-195, fft1.c:637-650) + filter correction of fft1_c (fft1.c:4119-4127)
//   k_sumsq      power accumulation of fft1_c (fft1.c:4115-4171), one averaging group per blockIdx.y
//   k_slowsum    update_fft1_slowsum (fft1.c:4526-4605) + new_fft1_averages (wide_graph.c:1003-1032)
//   k_timf2      make_timf2 + fft1back_one + fft1back_fp_finish (timf2.c:31-75, 689-967, 970-1064)
//   k_blank_*    stupid blanker + noise statistics of first_noise_blanker (blank1.c:1003-1087, 1458-1601)
//   k_fft2       make_fft2 mode 15 (fft2.c:91-141, 647-670)
//   k_powersum2 / k_waterfall   fft2.c:655-670, 707-815
//   k_mix1_back / k_mix1_out    fft2_mix1_fixed + do_mix1 (mix1.c:934-993, 55-195)
//
// All spectra / time functions live in device rings; every kernel is HBM-bound, so the design rule is one pass
// per stage with everything element-wise fused into the transform's load or store.
#include <type_traits>
#include "lrh_fft.hip.h"
#include "lrh_kernels.hip.h"

namespace lrh {
constexpr int LRH_MAX_REFPULSES_K = 256;            // MAX_REFPULSES, blnkdef.h:6
// streaming store: a ring that is written once here and read next by another kernel, a whole launch later, need not displace what
// this kernel re-reads from the L2
__device__ __forceinline__ float2 load_stream(const float2 *p)      // last use of a line: it need not stay in the L2
{
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(p));
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ void store_stream(float2 *p, float2 v)
{
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f ov = { v.x, v.y };
  __builtin_nontemporal_store(ov, reinterpret_cast<v2f *>(p));
}

// =====================================================================================================
// fft1
// =====================================================================================================
// DW: int32 samples (DWORD_INPUT); SKEW: I and Q taken from different sample instants (ui.sample_shift != 0);
// REAL: real samples, the pair (x[2n], x[2n+1]) is one complex point with its own window value per component, nothing
// negated, natural bin order, bare transform (k_realsplit finishes fft1_reherm_dit_one)
template <int LOG2N, bool DW, bool SKEW, bool REAL = false>
__global__ __launch_bounds__(fft1_threads(LOG2N), fft_min_waves(LOG2N)) void k_fft1(Fft1Args a)
{
  using Raw = typename std::conditional<DW, int2, short2>::type;
  using Comp = typename std::conditional<DW, int, short>::type;
  constexpr int P = points_fft1(LOG2N);
  using Plan = FftPlan<LOG2N, P>;
  using Fft = BlockFftL<LOG2N, P, +1>;
  constexpr int N = Plan::N, T = Plan::T, R0 = Plan::R0, RL = Plan::RL;
  __shared__ float2 lds[Fft::LDS_CELLS];
  const int tid0 = threadIdx.x;
  // Persistent workgroups: with one ~150 KiB workgroup per CU at N = 16384 nothing else can overlap the store tail
  // of transform t with the loads of transform t+1, so each workgroup walks over several transforms itself and
  // fetches the raw samples of the next one before it starts computing the current one.  Everything that does not
  // change between transforms stays on chip (twiddles in LDS) or is fetched ahead of the stores (window
  // values, filter correction): the only loads that follow a transform's stores are the next prefetch, which is
  // not needed for a whole transform, so no wait ever covers a freshly issued store (vmcnt retires in order).
  // diagnostics, compiled in with -DLRH_STAMP_BUILD only (the counter costs registers the N = 16384 kernel does not have):
  // one lane of two workgroups writes the shader clock at the phase boundaries; run with LRH_STAMP=1
#ifdef LRH_STAMP_BUILD
  int nstamp = 0;
  auto stamp = [&]() {
    if (a.stamps && tid0 == 0 && (blockIdx.x == 0 || blockIdx.x == 128) && nstamp < LRH_STAMPS_PER_WG)
      a.stamps[(blockIdx.x ? LRH_STAMPS_PER_WG : 0) + nstamp++] = __builtin_amdgcn_s_memtime();
  };
#else
  auto stamp = []() {};
#endif
  stamp();
  Fft::init(lds, a.tw, tid0);
  float win[P];
  float winq[REAL ? P : 1];
  auto load_window = [&](int tid) {
#pragma unroll
    for (int m = 0; m < P / R0; m++)
#pragma unroll
      for (int s = 0; s < R0; s++) {
        if constexpr (REAL) { const float2 w = ((const float2 *)a.window)[(tid + m * T) + s * (N / R0)]; win[m * R0 + s] = w.x; winq[m * R0 + s] = w.y; }
        else win[m * R0 + s] = a.window[(tid + m * T) + s * (N / R0)];
      }
  };
  auto out_index = [&](int tid, int e) {
    const int k = (tid + (e / RL) * T) + (e % RL) * (N / RL);
    if constexpr (REAL) return k;
    int kk = (k + N / 2) & (N - 1);                     // DC at N/2 (make_permute mode 1, fft0.c:1196-1204)
    if (a.direction < 0) kk = (N - kk) & (N - 1);       // fft1.c:3660-3679
    return kk;
  };
  load_window(tid0);
  Raw nxt[P];
  auto fetch = [&](int bi, int tid) {
    const int p0 = a.p0_first + bi * a.step;
#pragma unroll
    for (int m = 0; m < P / R0; m++)
#pragma unroll
      for (int s = 0; s < R0; s++) {
        const int n = p0 + (tid + m * T) + s * (N / R0);
        if constexpr (REAL) {
          if (a.chan_count > 1) {                          // frames {a_k, b_k, ..}: this channel's reals 2n and 2n+1 (fft1_re.c:146-156)
            const Comp *c = (const Comp *)a.timf1;
            const int cm = 2 * a.ring_mask + 1;
            nxt[m * R0 + s].x = c[((2 * n) * a.chan_count + a.chan_index) & cm];
            nxt[m * R0 + s].y = c[((2 * n + 1) * a.chan_count + a.chan_index) & cm];
            continue;
          }
        }
        if constexpr (!SKEW) nxt[m * R0 + s] = ((const Raw *)a.timf1)[(n * a.chan_count + a.chan_index) & a.ring_mask];
        else {
          const Comp *c = (const Comp *)a.timf1;
          nxt[m * R0 + s].x = c[2 * (((n + a.shift_i) * a.chan_count + a.chan_index) & a.ring_mask)];
          nxt[m * R0 + s].y = c[2 * (((n + a.shift_q) * a.chan_count + a.chan_index) & a.ring_mask) + 1];
        }
      }
  };
  int bi = blockIdx.x;
  if (bi < a.batch) fetch(a.xcd ? xcd_order(bi, a.batch) : bi, tid0);
  __syncthreads();                                       // twiddle tables are in place
  // Retire the prologue's loads here: the compiler merges wait counts over both loop entries, and the first
  // prefetch being the youngest load on this path would otherwise turn the loop-top wait into vmcnt(0) for every
  // trip -- which also waits for the stores of the previous transform.
  __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0), other counters untouched
  stamp();
#pragma unroll 1
  for (; bi < a.batch; bi += gridDim.x) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));                        // keep index math inside the loop (see k_timf2)
    const int b = a.xcd ? xcd_order(bi, a.batch) : bi;
    stamp();
    float2 x[P];
#pragma unroll
    for (int e = 0; e < P; e++) {
      const Raw v = nxt[e];
      // Q negated before the e^{+j} transform: conj(FFT(x w)) (fft1.c:432-447)
      if constexpr (REAL) x[e] = make_float2((float)v.x * win[e], (float)v.y * winq[e]);
      else x[e] = make_float2((float)v.x * win[e], -((float)v.y * win[e]));
    }
    const int bn = bi + gridDim.x;
    if (bn < a.batch) fetch(a.xcd ? xcd_order(bn, a.batch) : bn, tid);
    // Every load the next transform needs before its first wait (its window values) and this one's filter correction
    // are issued BEFORE the stores, so that no later wait has to cover the stores; the filter correction already
    // before the last pass, whose butterflies hide its latency.
    float2 fc[P];
    constexpr int EARLY = (DW || SKEW || REAL) ? 0 : P / 2;       // as many as the register file holds next to the last pass
    Fft::run(x, lds, tid, [&]() {
#pragma unroll
      for (int e = 0; e < EARLY; e++) fc[e] = a.filtercorr[out_index(tid, e)];
    });
#pragma unroll
    for (int e = EARLY; e < P; e++) fc[e] = a.filtercorr[out_index(tid, e)];
    stamp();
    float2 *out = a.out + (size_t)((a.first_nb + b) & a.nb_mask) * N;
    load_window(tid);
#pragma unroll
    for (int e = 0; e < P; e++) {
      float2 v = x[e];
      if (a.direction < 0) v = make_float2(v.y, v.x);   // fft1.c:3660-3679
      store_stream(&out[out_index(tid, e)], cmul(v, fc[e]));
    }
    stamp();                                             // no barrier: BlockFftL protects its buffer itself
  }
}

// Real input (fft1 version 2, fft1_reherm_dit_one, fft1_re.c:32-131), second half.  k_fft1<REAL> left
// S[k] = sum_n (x[2n] w[2n] + j x[2n+1] w[2n+1]) e^{+2 pi j nk/N} = F[(N-k) mod N] in natural order; the 2N-point real
// transform Z (kernel e^{-j}) follows from the even / odd split
//   E_k = (F_k + conj F_{N-k}) / 2,  O_k = (F_k - conj F_{N-k}) / 2j,  Z_k = E_k + e^{-j pi k/N} O_k,  Z_{N-k} = conj(E_k - e^{-j pi k/N} O_k),
//   Z_0 = Re F_0 + Im F_0,  Z_N = Re F_0 - Im F_0,  Z_{N/2} = conj F_{N/2}.
// Layout of the reference (the real part goes to the imaginary slot): direction > 0: out[k] = (Im Z_k, Re Z_k), out[0] =
// (Z_N, Z_0); direction < 0: out[N-k] = (Re Z_k, Im Z_k), out[0] = (Z_N, Z_N) (fft1_re.c:100-128).  One thread per pair
// (k, N-k), in place, then the filter correction of fft1_c like k_fft1's store epilogue.
__global__ __launch_bounds__(256) void k_realsplit(RealSplitArgs a)
{
  const int k = blockIdx.x * 256 + threadIdx.x, N = a.n;
  if (k >= N / 2) return;
  float2 *out = a.spec + (size_t)((a.first_nb + blockIdx.y) & a.nb_mask) * N;
  if (k == 0) {
    const float2 s0 = out[0], sh = out[N / 2];
    const float z0 = s0.x + s0.y, zn = s0.x - s0.y;
    const float2 zh = make_float2(sh.x, -sh.y);
    float2 o0, oh;
    if (a.direction > 0) { o0 = make_float2(zn, z0); oh = make_float2(zh.y, zh.x); }
    else { o0 = make_float2(zn, zn); oh = zh; }
    out[0] = cmul(o0, a.filtercorr[0]); out[N / 2] = cmul(oh, a.filtercorr[N / 2]);
    return;
  }
  const float2 sa = out[k], sb = out[N - k];              // F_{N-k}, F_k
  const float2 e = make_float2(0.5f * (sb.x + sa.x), 0.5f * (sb.y - sa.y));
  const float2 d = make_float2(0.5f * (sb.x - sa.x), 0.5f * (sb.y + sa.y));     // (F_k - conj F_{N-k}) / 2
  const float2 o = make_float2(d.y, -d.x);                                       // / j
  float sn, cs; sincospif((float)k / (float)N, &sn, &cs);
  const float2 t = make_float2(cs * o.x + sn * o.y, cs * o.y - sn * o.x);        // e^{-j pi k/N} O_k
  const float2 zk = make_float2(e.x + t.x, e.y + t.y), zm = make_float2(e.x - t.x, -(e.y - t.y));   // Z_k, Z_{N-k}
  if (a.direction > 0) {
    out[k] = cmul(make_float2(zk.y, zk.x), a.filtercorr[k]);
    out[N - k] = cmul(make_float2(zm.y, zm.x), a.filtercorr[N - k]);
  } else {
    out[N - k] = cmul(zk, a.filtercorr[N - k]);
    out[k] = cmul(zm, a.filtercorr[k]);
  }
}
// NET_RXOUT_TIMF2 payload, float form (rxin.c:949-956): gain * (weak + strong_scale * strong) per sample
__global__ __launch_bounds__(256) void k_timf2_net(const float2 *w, const float2 *s, int mask, int first, int count, float gain, float strong, float2 *dst)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const float2 a = w[(first + i) & mask], b = s[(first + i) & mask];
  dst[i] = make_float2(gain * (a.x + strong * b.x), gain * (a.y + strong * b.y));
}
hipError_t launch_timf2_net(const float2 *w, const float2 *s, int mask, int first, int count, float gain, float strong, float2 *dst, hipStream_t st)
{
  hipLaunchKernelGGL(k_timf2_net, dim3((count + 255) / 256), dim3(256), 0, st, w, s, mask, first, count, gain, strong, dst);
  return hipGetLastError();
}
hipError_t launch_realsplit(const RealSplitArgs &a, int batch, hipStream_t st)
{
  hipLaunchKernelGGL(k_realsplit, dim3((a.n / 2 + 255) / 256, batch), dim3(256), 0, st, a);
  return hipGetLastError();
}

// =====================================================================================================
// recorded-IQ ingest: expand_rawdat (csplit.c:20-73), 18-bit packed -> left-justified int32 with the half-LSB bit
// =====================================================================================================
// One thread per 9-byte group: eight data bytes (four little-endian 16-bit top parts) and one byte holding the four
// 2-bit bottom parts, first sample in the top bits.  Output: one 16-byte store into the ring.
__global__ __launch_bounds__(256) void k_expand18(const unsigned char *packed, int ngroups, int4 *ring, int first_group, int group_mask)
{
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const unsigned char *r = packed + (size_t)9 * g;
  unsigned int m = r[8];
  int v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const unsigned int n = ((m << (2 * k)) & 0xc0u) | 0x20u;                      // csplit.c:37-39 (m is an unsigned char there)
    v[k] = (int)(((unsigned int)r[2 * k] << 16) | ((unsigned int)r[2 * k + 1] << 24) | (n << 8));
  }
  ring[(first_group + g) & group_mask] = make_int4(v[0], v[1], v[2], v[3]);
}
hipError_t launch_expand18(const unsigned char *packed, int ngroups, void *ring, int first_group, int group_mask, hipStream_t st)
{
  hipLaunchKernelGGL(k_expand18, dim3((ngroups + 255) / 256), dim3(256), 0, st, packed, ngroups, (int4 *)ring, first_group, group_mask);
  return hipGetLastError();
}

// =====================================================================================================
// I/Q mirror-image cancellation (CALIQ, fft1.c:3598-3658) followed by the filter correction of fft1_c
// =====================================================================================================
// Runs only with a calibration table: k_fft1 then stores the bare transform (unit filter table, direction +1) and
// this pass orthogonalises each bin ib against its mirror N - ib, applies the direction flip of the reference's
// combined branch and the filter correction.  One thread per mirror pair; thread 0 takes bins 0 and N/2.
__global__ __launch_bounds__(256) void k_foldcorr(FoldcorrArgs a)
{
  const int ia = blockIdx.x * 256 + threadIdx.x;
  if (ia >= a.n / 2) return;
  float2 *out = a.spec + (size_t)((a.first_nb + blockIdx.y) & a.nb_mask) * a.n;
  if (ia == 0) {                                          // bins 0 (pc) and N/2: no mirror partner (m = 1)
    float2 z0 = out[0], zh = out[a.n / 2];
    if (a.direction < 0) { z0 = make_float2(z0.y, z0.x); zh = make_float2(zh.y, zh.x); }   // fft1.c:3648-3653
    out[0] = cmulc(z0, a.filtercorr[0]); out[a.n / 2] = cmulc(zh, a.filtercorr[a.n / 2]);
    return;
  }
  const int ib = ia, ic = a.n - ia;
  const float2 zb = out[ib], zc = out[ic], fa = a.foldcorr[ia], fcc = a.foldcorr[ic];
  float2 nb, nc;
  if (a.direction > 0) {                                  // fft1.c:3611-3627
    const float t1 = zb.x * fa.x - zb.y * fa.y, t2 = zb.x * fa.y + zb.y * fa.x;
    nb = make_float2(zb.x - (zc.x * fcc.x + zc.y * fcc.y), zb.y - (zc.x * fcc.y - zc.y * fcc.x));
    nc = make_float2(zc.x - t1, zc.y + t2);
  } else {                                                // fft1.c:3629-3647
    const float t1 = zc.x - zb.x * fa.x + zb.y * fa.y;
    const float t2 = zc.y + zb.x * fa.y + zb.y * fa.x;
    const float t3 = zb.x - zc.x * fcc.x - zc.y * fcc.y;
    const float t4 = zb.y - zc.x * fcc.y + zc.y * fcc.x;
    nb = make_float2(t2, t1); nc = make_float2(t4, t3);
  }
  out[ib] = cmulc(nb, a.filtercorr[ib]); out[ic] = cmulc(nc, a.filtercorr[ic]);
}
hipError_t launch_foldcorr(const FoldcorrArgs &a, int batch, hipStream_t st)
{
  hipLaunchKernelGGL(k_foldcorr, dim3((a.n / 2 + 255) / 256, batch), dim3(256), 0, st, a);
  return hipGetLastError();
}

// =====================================================================================================
// fft1_c power sums and slow average
// =====================================================================================================
__global__ __launch_bounds__(256) void k_sumsq(SumsqArgs a)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int g = blockIdx.y;
  const int start = g == 0 ? 0 : g * a.avg - a.c0;
  int count = a.avg - (g == 0 ? a.c0 : 0);
  if (count > a.batch - start) count = a.batch - start;
  const bool accumulate = g == 0 && a.c0 > 0;
  float *dst = a.sumsq + ((a.pa0 + g * a.n) & a.sumsq_mask);
  float acc = accumulate ? dst[i] : 0.0f;
  for (int j = 0; j < count; j++) {
    const float2 z = a.spec[(size_t)((a.first_nb + start + j) & a.nb_mask) * a.n + i];
    const float pw = z.x * z.x + z.y * z.y;
    acc = (j == 0 && !accumulate) ? pw : acc + pw;      // "=" for the first spectrum of a group, "+=" after (fft1.c:4126, 4169)
  }
  dst[i] = acc;
}

// Averaging groups that k_timf2<.., SS> could not finish inside one workgroup's run: add the pieces in transform order.
__global__ __launch_bounds__(256) void k_sumsq_join(SumsqArgs a, const float *part, int run)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  // blockIdx.y = w < nruns: the group that run w finds already begun, taken by the first run that starts inside it;
  // blockIdx.y = nruns: the unfinished last group of the call when it began inside the last run
  const int nruns = (a.batch + run - 1) / run, y = blockIdx.y;
  const int g = y < nruns ? (y * run + a.c0) / a.avg : (a.batch - 1 + a.c0) / a.avg;
  const int gs = g * a.avg - a.c0;                       // first transform of the group (< 0: it continues an earlier call)
  const int start = gs < 0 ? 0 : gs;
  int end = gs + a.avg; if (end > a.batch) end = a.batch;
  const int w0 = start / run, w1 = (end - 1) / run;
  if (y < nruns) { if (!(gs < y * run && (y == 0 || gs >= (y - 1) * run))) return; }
  else if (!(gs >= (nruns - 1) * run && gs + a.avg > a.batch)) return;
  float *dst = a.sumsq + ((a.pa0 + g * a.n) & a.sumsq_mask);
  float acc = 0.f; bool first = gs >= 0;
  if (!first) acc = dst[i];
  for (int w = w0; w <= w1; w++) {
    const float v = part[(size_t)(2 * w + (gs < w * run ? 0 : 1)) * a.n + i];
    acc = first ? v : acc + v; first = false;
  }
  dst[i] = acc;
}

#define LRH_FFT1_SMALL 0.00000001F
__global__ __launch_bounds__(64) void k_slowsum(SlowsumArgs a)
{
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
  const int mask = a.bufsize - 1, last = a.n - 1;
  // Every update recomputes a rolling window of bins from scratch (fft1.c:4567-4573), which overwrites the running
  // value there.  The value after the batch therefore depends only on the bin's last refresh and the sliding
  // updates after it: find that update first (integer bookkeeping only), then replay from it.
  // The host hands over a start e0 at least one full refresh cycle before the end when the call is long enough: every
  // bin is refreshed inside it, so what happened before e0 cannot matter (e0 = 0: short call, whole history replayed).
  int e_start = 0, recalc = a.recalc_e0, recalc_at_start = a.recalc0; bool refreshed = false;
  for (int e = a.e0; e < a.nupd; e++) {
    const int before = recalc;
    if (recalc == last) recalc = 0;
    const int ia = recalc;
    recalc += a.step; if (recalc > last) recalc = last;
    if (i >= ia && i <= recalc) { e_start = e; recalc_at_start = before; refreshed = true; }
  }
  float slow = refreshed ? 0.f : a.slowsum[i];
  recalc = recalc_at_start;
  int e = e_start;
  if (refreshed) {                                     // the bin's last refresh: from scratch over the window (wide_graph.c:1016-1031)
    const int pa = (a.pa0 + e * a.n) & mask;
    int p0 = (pa - (a.avg2 - 1) * a.n + a.bufsize) & mask;
    slow = a.sumsq[p0 + i];
    p0 = (p0 + a.n) & mask;
    for (int m = 1; m < a.avg2; m++) {
      slow += a.sumsq[p0 + i];
      if (slow < LRH_FFT1_SMALL) slow = LRH_FFT1_SMALL;
      p0 = (p0 + a.n) & mask;
    }
    e++;
  }
  // everything after it is the sliding update (fft1.c:4574-4583): same additions in the same order, but the loads do
  // not depend on the running value, so eight updates' worth are in flight at a time
  for (; e < a.nupd; e += 8) {
    float va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int pa = (a.pa0 + (e + u) * a.n) & mask, pb = (pa - a.avg2 * a.n + a.bufsize) & mask;
      va[u] = (e + u < a.nupd) ? a.sumsq[pa + i] : 0.f; vb[u] = (e + u < a.nupd) ? a.sumsq[pb + i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (e + u < a.nupd) { slow += va[u] - vb[u]; if (slow < LRH_FFT1_SMALL) slow = LRH_FFT1_SMALL; }
  }
  a.slowsum[i] = slow;
}

// =====================================================================================================
// timf2: strong/weak split, two back transforms, window handling, power
// =====================================================================================================
// sin^2 window, 50 % overlap: the reference adds the first half of transform t onto the stored second half of
// transform t-1 (timf2.c:1003-1026).  Second half of DFT(S)[n + N/2] = DFT(S (-1)^k)[n], so
//     out_t[n] = ampfac * DFT( S_t + (-1)^k S_{t-1} )[n],  n < N/2
// i.e. one transform of the combined spectrum, written once -- no read-modify-write of the timf2 ring.
// One transform's outputs of stream ST (0 weak, 1 strong) to the planar rings.
template <int LOG2N, int MODE, int ST>
__device__ __forceinline__ void timf2_store(const Timf2Args &a, const float2 (&x)[points_fft1(LOG2N)], int pa, int tid)
{
  constexpr int P = points_fft1(LOG2N);
  using Plan = FftPlan<LOG2N, P>;
  constexpr int N = Plan::N, T = Plan::T, RL = Plan::RL;
#pragma unroll
  for (int m = 0; m < P / RL; m++)
#pragma unroll
    for (int q = 0; q < RL; q++) {
      if constexpr (MODE == 1) { if (q >= RL / 2) continue; }   // first half only: n = i + q N/RL < N/2  <=>  q < RL/2
      const int n = (tid + m * T) + q * (N / RL);
      float amp = a.ampfac; int pos = n; bool keep = true;
      if constexpr (MODE == 2) {                          // centre part x inverted window (timf2.c:1031-1061)
        keep = (n >= a.ia) && (n < N - a.ia); pos = n - a.ia; amp = a.invwin[n] * a.ampfac;
      }
      if (keep) {
        const float2 v = x[m * RL + q];
        const float2 o = make_float2(amp * v.x, amp * v.y);
        if constexpr (MODE == 1) {
          // pa is a multiple of N/2 here (launch_timf2 checks it): no wrap inside the half block, so the address
          // is a uniform base plus the thread index
          const size_t base = (size_t)(pa & a.mask) + m * T + q * (N / RL);
          const unsigned int t = (unsigned int)tid;
          // written once and not read again before a whole round has passed: streaming stores leave the L2 to the spectra,
          // whose second read (the previous transform's copy, one trip later) is what can hit there
          typedef float v2f __attribute__((ext_vector_type(2)));
          const v2f ov = { o.x, o.y };
          if constexpr (ST == 0) {
            __builtin_nontemporal_store(ov, reinterpret_cast<v2f *>(a.timf2w + base) + t);
            __builtin_nontemporal_store(o.x * o.x + o.y * o.y, (a.pwr + base) + t);   // weak power only (timf2.c:1010-1012)
          } else __builtin_nontemporal_store(ov, reinterpret_cast<v2f *>(a.timf2s + base) + t);
        } else {
          const int r = (pa + pos) & a.mask;
          if constexpr (ST == 0) { a.timf2w[r] = o; a.pwr[r] = o.x * o.x + o.y * o.y; }
          else a.timf2s[r] = o;
        }
      }
    }
}

// SS: fft1_c's power sums ride along (fft1.c:4115-4171).  The masked loads of the two streams together are exactly
// the transform's spectrum, so sum |X|^2 costs two FMAs per bin here instead of a second pass over the fft1 ring.  A
// workgroup then takes a run of consecutive transforms and keeps the running sums of the current averaging group in
// registers; a group that lies inside one run goes straight to the fft1_sumsq ring (same additions in the same order
// as the reference), the pieces of a group that straddles runs (or continues an earlier call) go to `ss_part` and
// k_sumsq_join adds them up in order.
// STRONG_ONLY: the weak stream (and the sums) of these transforms came out of k_fft1w, which left the strong bins of every spectrum
// in the fft1 ring: only the strong stream is transformed here, one item per transform.
template <int LOG2N, int MODE, bool SS, bool STRONG_ONLY = false>
__global__ __launch_bounds__(fft1_threads(LOG2N), fft_min_waves(LOG2N)) void k_timf2(Timf2Args a)
{
  static_assert(!STRONG_ONLY || (MODE == 1 && !SS), "strong-only pass: sin^2 overlap form, sums elsewhere");
  constexpr int P = points_fft1(LOG2N);
  using Plan = FftPlan<LOG2N, P>;
  using Fft = BlockFftL<LOG2N, P, -1>;
  constexpr int N = Plan::N, T = Plan::T, R0 = Plan::R0, NB0 = P / R0;
  __shared__ float2 lds[Fft::LDS_CELLS];
  const int tid0 = threadIdx.x;
  if constexpr (STRONG_ONLY) {
    if (a.sd_kmax > 0) {                                 // a handful of strong bins: k_timf2_sd (launched in front) has written this launch's samples
      __shared__ int s_strong;
      if (tid0 == 0) s_strong = 0;
      __syncthreads();
      int n = 0;
#pragma unroll
      for (int m = 0; m < NB0; m++) n += __popc(~(a.pack_cur[tid0 + m * T] & a.pack_prev[tid0 + m * T]) & ((1u << R0) - 1u));   // strong under either table, as k_timf2_sd counts
      for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
      if ((tid0 & 63) == 0 && n) atomicAdd(&s_strong, n);
      __syncthreads();
      if (s_strong <= a.sd_kmax) return;                 // (before the tables go to the LDS: the launch then costs a few microseconds)
    }
  }
  Fft::init(lds, a.tw, tid0);
  // weak/strong routing flags, packed by the host per first-pass butterfly: bit s of pack[i] is set when bin
  // i + s*(N/R0) is weak (liminfo == 0, timf2.c:50).  One dword per thread instead of R0 float loads; the same
  // for every transform of the launch except the "previous" half of the very first one.
  unsigned int wk_cur[NB0], wk_first[NB0];
#pragma unroll
  for (int m = 0; m < NB0; m++) { wk_cur[m] = a.pack_cur[tid0 + m * T]; wk_first[m] = a.pack_prev[tid0 + m * T]; }
  // Persistent workgroups walk over (transform, stream) items: weak stream then strong stream of each transform,
  // 16 points/thread of one stream being all the register file holds at 1024 threads.  The masked spectrum loads
  // of the NEXT item are issued before the stores of the current one (vmcnt retires in issue order: a load issued
  // behind the stores would not return before they have drained), and the spectra are re-read for the second
  // stream (L2 hits).  Bins routed to the other stream are not even fetched.
  float2 c[P], pv[P];
  auto issue = [&](int b, int st, int tid) {
    const float2 *cur = a.spec + (size_t)((a.first_nb + b) & a.nb_mask) * N;
    const float2 *prv = a.spec + (size_t)((a.first_nb + b - 1) & a.nb_mask) * N;
#pragma unroll
    for (int m = 0; m < NB0; m++) {
      const unsigned int flip = st ? 0xffffffffu : 0u;
      const unsigned int mc = wk_cur[m] ^ flip;           // bit set -> bin belongs to this stream
      const unsigned int mp = (b == 0 ? wk_first[m] : wk_cur[m]) ^ flip;
#pragma unroll
      for (int s = 0; s < R0; s++) {
        const int ku = m * T + s * (N / R0);               // uniform part of the bin index; + tid per thread
        const unsigned int t = (unsigned int)tid;
        c[m * R0 + s] = make_float2(0.f, 0.f);
        if ((mc >> s) & 1u) c[m * R0 + s] = (cur + ku)[t];
        if constexpr (MODE == 1) {
          pv[m * R0 + s] = make_float2(0.f, 0.f);
          if ((mp >> s) & 1u) pv[m * R0 + s] = (prv + ku)[t];
        }
      }
    }
  };
  float acc[SS ? P : 1];
  if constexpr (SS) {
#pragma unroll
    for (int e = 0; e < P; e++) acc[e] = 0.f;
  }
  auto combine = [&](float2 (&x)[P], int tid) {
    // sin^2 overlap: S_t + (-1)^k S_{t-1}; k = tid + even offsets, so the sign is one per thread
    const float sg = (tid & 1) ? -1.f : 1.f;
#pragma unroll
    for (int e = 0; e < P; e++) {
      if constexpr (SS) acc[e] += c[e].x * c[e].x + c[e].y * c[e].y;   // a bin is zero in one of the two streams
      if constexpr (MODE == 1) x[e] = make_float2(c[e].x + sg * pv[e].x, c[e].y + sg * pv[e].y);
      else x[e] = c[e];
    }
    if constexpr (SS) {
      // the sums are due here: left to the scheduler they sink into the transform and keep the spectrum alive with them
#pragma unroll
      for (int e = 0; e < P; e++) asm volatile("" : "+v"(acc[e]));
    }
  };
  // end of an averaging group or of this workgroup's run: the sums leave the registers
  auto flush_sums = [&](int b, int r0, int r1, int tid) {
    const int gb = b + a.ss_c0, g = gb / a.ss_avg;
    const bool group_end = gb - g * a.ss_avg == a.ss_avg - 1;
    if (!group_end && b != r1 - 1) return;
    const bool head = g * a.ss_avg - a.ss_c0 < r0;       // began in an earlier run (or an earlier call)
    // ss_part == nullptr: one workgroup runs the whole launch (a call of one or two transforms), so nobody else holds a piece of any
    // group: the piece of a group an earlier call began is added to the ring here, in k_sumsq_join's order, and that launch is saved
    const bool direct = a.ss_part == nullptr;
    float *dst;
    if ((!head && group_end) || direct) dst = a.ss_ring + ((a.ss_pa0 + g * N) & a.ss_mask);
    else dst = a.ss_part + (size_t)(2 * blockIdx.x + (head ? 0 : 1)) * N;
    const bool add = direct && head;
#pragma unroll
    for (int m = 0; m < NB0; m++)
#pragma unroll
      for (int s = 0; s < R0; s++) {
        float *const q = dst + m * T + s * (N / R0);
        float v = acc[m * R0 + s];
        if (add) v = q[(unsigned int)tid] + v;
        q[(unsigned int)tid] = v;
        acc[m * R0 + s] = 0.f;
      }
  };
  // Pins the transform's live outputs in registers at this point: without it the last butterflies sink below the
  // prefetch (towards the stores that use them) and their 16 inputs stay live across 32 loads in flight.
  auto pin = [&](float2 (&x)[P]) {
    constexpr int RL = Plan::RL;
#pragma unroll
    for (int e = 0; e < P; e++)
      if (MODE != 1 || (e % RL) < RL / 2) asm volatile("" : "+v"(x[e].x), "+v"(x[e].y));
    asm volatile("" ::: "memory");
  };
  if constexpr (SS) {
    if (a.ss_split) {
      // One transform per call (Linrad's own call pattern, wcw.c:1036-1047): the two streams' transforms one after the other in one
      // workgroup are two latencies of ~11 us; here one workgroup per stream.  A bin is zero in one of the two streams, so each
      // workgroup holds exactly its own stream's bins of sum |X|^2 and writes those to the ring ("=" for the first transform of an
      // averaging group, "+=" after: fft1.c:4126, 4169 -- the addition k_sumsq_join would make).
      const int st = (int)blockIdx.x & 1;
      issue(0, st, tid0);
      __syncthreads();
      __builtin_amdgcn_s_waitcnt(0x0F70);
      float2 x[P];
      combine(x, tid0);
      {
        const int g = a.ss_c0 / a.ss_avg;
        const bool add = g * a.ss_avg - a.ss_c0 < 0;       // the group began in an earlier call
        float *const dst = a.ss_ring + ((a.ss_pa0 + g * N) & a.ss_mask);
#pragma unroll
        for (int m = 0; m < NB0; m++) {
          const unsigned int own = wk_cur[m] ^ (st ? 0xffffffffu : 0u);
#pragma unroll
          for (int s = 0; s < R0; s++)
            if ((own >> s) & 1u) {
              float *const q = dst + m * T + s * (N / R0);
              float v = acc[m * R0 + s];
              if (add) v = q[(unsigned int)tid0] + v;
              q[(unsigned int)tid0] = v;
            }
        }
      }
      Fft::run(x, lds, tid0);
      if (st) timf2_store<LOG2N, MODE, 1>(a, x, a.pa_first, tid0);
      else timf2_store<LOG2N, MODE, 0>(a, x, a.pa_first, tid0);
      return;
    }
  }
  // items: interleaved over the grid (XCD-aware order), or with SS a run of consecutive transforms per workgroup
  const int stride = SS ? 1 : (int)gridDim.x;
  const int r0 = SS ? (int)blockIdx.x * a.ss_run : 0;
  const int r1 = SS ? min(r0 + a.ss_run, a.batch) : a.batch;
  auto order = [&](int i) { return (!SS && a.xcd) ? xcd_order(i, a.batch) : i; };
  int bi = SS ? r0 : (int)blockIdx.x;
  if (bi < r1) issue(order(bi), STRONG_ONLY ? 1 : 0, tid0);
  __syncthreads();                                       // twiddle tables are in place
  __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): retire the prologue's loads here (see k_fft1)
  if constexpr (STRONG_ONLY) {
#pragma unroll 1
    for (; bi < r1; bi += stride) {
      int tid = tid0;
      asm volatile("" : "+v"(tid));
      const int b = order(bi);
      const int pa = a.pa_first + b * a.step;
      float2 x[P];
      combine(x, tid);
      Fft::run(x, lds, tid);
      pin(x);
      __builtin_amdgcn_sched_barrier(0);
      const int bn = min(bi + stride, r1 - 1);
      issue(order(bn), 1, tid);
      __builtin_amdgcn_sched_barrier(0);
      timf2_store<LOG2N, MODE, 1>(a, x, pa, tid);
    }
    return;
  }
#pragma unroll 1
  for (; bi < r1; bi += stride) {
    // opaque per-iteration copy of the thread index: without it LICM hoists every address and LDS index of the
    // transform out of the loop and parks them in ~200 VGPRs (spills at 1024 threads)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int b = order(bi);
    const int pa = a.pa_first + b * a.step;
    float2 x[P];
    combine(x, tid);
    Fft::run(x, lds, tid);
    pin(x);
    __builtin_amdgcn_sched_barrier(0);                   // keep the loads below out of the transform (register file)
    issue(b, 1, tid);
    __builtin_amdgcn_sched_barrier(0);
    timf2_store<LOG2N, MODE, 0>(a, x, pa, tid);         // no barrier: BlockFftL protects its buffer itself
    combine(x, tid);
    if constexpr (SS) flush_sums(b, r0, r1, tid);
    Fft::run(x, lds, tid);
    pin(x);
    __builtin_amdgcn_sched_barrier(0);
    // unconditional (the last trip re-reads its own transform): a conditional prefetch would keep the old c/pv
    // alive across the transform above
    const int bn = min(bi + stride, r1 - 1);
    issue(order(bn), 0, tid);
    __builtin_amdgcn_sched_barrier(0);
    timf2_store<LOG2N, MODE, 1>(a, x, pa, tid);
  }
}


// =====================================================================================================
// correlation spectrum (fft1_correlation_flag == 1)
// =====================================================================================================
// fft1_c's second sum (fft1.c:4146-4150, 4189-4193): per bin and averaging period 2 X conj(Y), "=" for the period's first transform
// and "+=" after, in transform order -- k_sumsq's group arithmetic on the two channels' exchanged transforms.
__global__ __launch_bounds__(256) void k_corrsum(CorrArgs a)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int g = blockIdx.y;
  const int start = g == 0 ? 0 : g * a.avg - a.c0;
  int count = a.avg - (g == 0 ? a.c0 : 0);
  if (count > a.batch - start) count = a.batch - start;
  const bool accumulate = g == 0 && a.c0 > 0;
  float2 *dst = a.corrsum + ((a.pa0 + g * a.n) & a.sumsq_mask);
  float2 acc = accumulate ? dst[i] : make_float2(0.f, 0.f);
  for (int b = 0; b < count; b++) {
    const size_t t = (size_t)(start + b) * a.n + i;
    const float2 x = a.x[t], y = a.y[t];
    const float re = 2 * (x.x * y.x + x.y * y.y), im = 2 * (x.y * y.x - x.x * y.y);
    if (b == 0 && !accumulate) acc = make_float2(re, im); else { acc.x += re; acc.y += im; }
  }
  dst[i] = acc;
}
// update_fft1_slowsum's part for the correlation sums (fft1.c:4584-4603 with new_fft1_averages, wide_graph.c:1033-1050), one update
// per completed averaging period, replayed per bin in order: inside the rolling refresh window the sliding sum is rebuilt from its
// wg_fft_avg2num periods, outside it takes the newest period and drops the oldest; the grand total adds every period (double).
__global__ __launch_bounds__(64) void k_slowcorr(CorrArgs a)
{
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
  const int mask = a.bufsize - 1, last = a.n - 1;
  float2 slow = a.slowcorr[i];
  double2 tot = a.tot[i];
  int recalc = a.recalc0;
  for (int e = 0; e < a.nupd; e++) {
    const int pa = (a.pa0 + e * a.n) & mask;
    if (recalc == last) recalc = 0;
    const int ia = recalc;
    recalc += a.step; if (recalc > last) recalc = last;
    if (i >= ia && i <= recalc) {
      int p0 = (pa - (a.avg2 - 1) * a.n + a.bufsize) & mask;
      slow = a.corrsum[p0 + i];
      for (int m = 1; m < a.avg2; m++) { p0 = (p0 + a.n) & mask; const float2 v = a.corrsum[p0 + i]; slow.x += v.x; slow.y += v.y; }
    } else {
      const int pb = (pa - a.avg2 * a.n + a.bufsize) & mask;
      const float2 nw = a.corrsum[pa + i], od = a.corrsum[pb + i];
      slow.x += nw.x - od.x; slow.y += nw.y - od.y;
    }
    const float2 nw = a.corrsum[pa + i];
    tot.x += nw.x; tot.y += nw.y;
  }
  a.slowcorr[i] = slow; a.tot[i] = tot;
}
hipError_t launch_corrsum(const CorrArgs &a, hipStream_t st)
{
  const int groups = (a.c0 + a.batch + a.avg - 1) / a.avg;
  hipLaunchKernelGGL(k_corrsum, dim3((a.n + 255) / 256, groups), dim3(256), 0, st, a);
  if (a.nupd > 0) hipLaunchKernelGGL(k_slowcorr, dim3((a.n + 63) / 64), dim3(64), 0, st, a);
  return hipGetLastError();
}

// =====================================================================================================
// fft1 + fft1_c's sums + make_timf2's weak stream in one kernel (fft1_size 16384, sin^2 window, int16 I/Q)
// =====================================================================================================
// The forward transform leaves a thread exactly the bins its back transform starts from (bin = tid mod N/32 in both layouts), so
// the spectrum never has to travel through HBM between the two: per transform 64 KB of samples come in and 96 KB of weak time
// function + power go out, where k_fft1 + k_timf2 moved 128 KB out, 128-256 KB back in and 160 KB out.  512 threads x 32 points
// (two waves per SIMD, 256 VGPRs): a thread holds the new spectrum, the weak part of the previous one (the overlap partner, see
// k_timf2) and fft1_c's running sums at the same time -- what 128 VGPRs at 1024 threads could not.  One set of twiddle tables in
// LDS serves both directions (BlockFftL CONJTAB).  The strong stream is a handful of bins: they are stored to the fft1 ring (all
// bins when keep_spec is set: fft1_float for whoever asks) and k_timf2<.., STRONG_ONLY> transforms them afterwards.
// A workgroup takes a run of consecutive transforms like k_timf2<.., SS>; the weak spectrum of the transform before its run is
// recomputed from the samples, which are still in the ring (one extra forward transform per run).
template <int LOG2N>
__global__ __launch_bounds__((1 << LOG2N) / 32, 1) void k_fft1w(Fft1wArgs a)
{
  constexpr int P = 32;
  using Plan = FftPlan<LOG2N, P>;
  using FftF = BlockFftL<LOG2N, P, +1>;
  using FftB = BlockFftL<LOG2N, P, -1, true>;
  constexpr int N = Plan::N, T = Plan::T, R0 = Plan::R0, RL = Plan::RL, NB0 = P / R0, NBL = P / RL;
  static_assert(RL == 4 && R0 == 16 && (N / RL) % T == 0 && (N / R0) % T == 0, "register maps below");
  __shared__ float2 lds[FftF::LDS_CELLS];
  const int tid0 = threadIdx.x;
  const int r0 = (int)blockIdx.x * a.run, r1 = min(r0 + a.run, a.batch);
  if (r0 >= r1) return;
  FftF::init(lds, a.tw, tid0);
  unsigned int wk_cur[NB0], wk_first[NB0];
#pragma unroll
  for (int m = 0; m < NB0; m++) { wk_cur[m] = a.pack_cur[tid0 + m * T]; wk_first[m] = a.pack_prev[tid0 + m * T]; }
  // register e of the forward transform's output (bin k' = tid + (e / RL) T + (e % RL) N/RL of the bare transform) is fft1 bin
  // kk = (k' + N/2) mod N (DC at N/2) = tid + J T with J = e / RL + (P / RL) ((e % RL + RL / 2) % RL), and the back transform wants
  // bin tid + (m + NB0 s) T in register m R0 + s: the same thread, register (J % NB0) R0 + J / NB0
  auto jof = [](int e) { return e / RL + (P / RL) * ((e % RL + RL / 2) % RL); };
  short2 raw[P];
  float win[P];
  auto fetch = [&](int b, int tid) {                     // samples and window values of transform b (b = -1: the one before the launch)
    const int p0 = a.p0_first + b * a.step;
#pragma unroll
    for (int m = 0; m < NB0; m++)
#pragma unroll
      for (int s = 0; s < R0; s++) {
        const int n = p0 + (tid + m * T) + s * (N / R0);
        raw[m * R0 + s] = ((const short2 *)a.timf1)[(n * a.chan_count + a.chan_index) & a.ring_mask];
        win[m * R0 + s] = (a.window + m * T + s * (N / R0))[(unsigned int)tid];
      }
  };
  // forward transform of the fetched samples, filter correction applied: x[e] = fft1_float bin tid + jof(e) T
  auto forward = [&](float2 (&x)[P], int tid) {
#pragma unroll
    for (int e = 0; e < P; e++) x[e] = make_float2((float)raw[e].x * win[e], -((float)raw[e].y * win[e]));   // Q negated (fft1.c:432-447)
    float2 fc[P];
    constexpr int EARLY = P / 4;
    FftF::run(x, lds, tid, [&]() {
#pragma unroll
      for (int e = 0; e < EARLY; e++) fc[e] = (a.filtercorr + jof(e) * T)[(unsigned int)tid];
    });
#pragma unroll
    for (int e = EARLY; e < P; e++) fc[e] = (a.filtercorr + jof(e) * T)[(unsigned int)tid];
#pragma unroll
    for (int e = 0; e < P; e++) x[e] = cmul(x[e], fc[e]);
  };
  auto weak_bit = [&](const unsigned int (&wk)[NB0], int e) { const int J = jof(e); return (wk[J % NB0] >> (J / NB0)) & 1u; };
  const float sg = (tid0 & 1) ? -1.f : 1.f;              // (-1)^bin: every register of a thread holds bins of the thread's parity
  float2 pw[P];                                          // weak part of the previous transform, back-transform register order
  float acc[P];
#pragma unroll
  for (int e = 0; e < P; e++) { pw[e] = make_float2(0.f, 0.f); acc[e] = 0.f; }
  __syncthreads();                                       // twiddle tables are in place
  if (r0 > 0 || a.have_prev) {
    fetch(r0 - 1, tid0);
    float2 x[P];
    forward(x, tid0);
#pragma unroll
    for (int e = 0; e < P; e++) {
      const int J = jof(e);
      const bool weak = r0 > 0 ? weak_bit(wk_cur, e) : weak_bit(wk_first, e);   // routed with the table in force for that transform
      pw[(J % NB0) * R0 + J / NB0] = weak ? x[e] : make_float2(0.f, 0.f);
    }
  }
  fetch(r0, tid0);
#pragma unroll 1
  for (int b = r0; b < r1; b++) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));                        // keep index math inside the loop (see k_timf2)
    float2 x[P];
    forward(x, tid);
    // fft1_c: sum |X|^2 over the averaging group (fft1.c:4115-4171), same additions in the same order as k_timf2<.., SS>
#pragma unroll
    for (int e = 0; e < P; e++) acc[e] += x[e].x * x[e].x + x[e].y * x[e].y;
    {
      const int gb = b + a.ss_c0, g = gb / a.ss_avg;
      const bool group_end = gb - g * a.ss_avg == a.ss_avg - 1;
      if (group_end || b == r1 - 1) {
        const bool head = g * a.ss_avg - a.ss_c0 < r0;   // began in an earlier run (or an earlier call)
        float *dst = (!head && group_end) ? a.ss_ring + ((a.ss_pa0 + g * N) & a.ss_mask) : a.ss_part + (size_t)(2 * blockIdx.x + (head ? 0 : 1)) * N;
#pragma unroll
        for (int e = 0; e < P; e++) { (dst + jof(e) * T)[(unsigned int)tid] = acc[e]; acc[e] = 0.f; }
      }
    }
    // the spectrum ring: every bin when somebody reads fft1_float, else the strong bins the second pass needs
    {
      float2 *out = a.spec + (size_t)((a.first_nb + b) & a.nb_mask) * N;
#pragma unroll
      for (int e = 0; e < P; e++)
        if (a.keep_spec || !weak_bit(wk_cur, e)) store_stream(&(out + jof(e) * T)[(unsigned int)tid], x[e]);
    }
    // weak stream: S_t + (-1)^k S_{t-1}, both routed with their own tables (k_timf2)
    float2 c[P];
#pragma unroll
    for (int e = 0; e < P; e++) {
      const int J = jof(e), ei = (J % NB0) * R0 + J / NB0;
      const float2 v = weak_bit(wk_cur, e) ? x[e] : make_float2(0.f, 0.f);
      c[ei] = make_float2(v.x + sg * pw[ei].x, v.y + sg * pw[ei].y);
      pw[ei] = v;
    }
    FftB::run(c, lds, tid);
#pragma unroll
    for (int e = 0; e < P; e++)
      if ((e % RL) < RL / 2) asm volatile("" : "+v"(c[e].x), "+v"(c[e].y));      // outputs pinned ahead of the prefetch (k_timf2)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    fetch(min(b + 1, r1 - 1), tid);                      // unconditional: the last trip re-reads its own samples
    __builtin_amdgcn_sched_barrier(0);
    {
      const size_t base = (size_t)((a.pa_first + b * a.step) & a.mask);
      typedef float v2f __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int m = 0; m < NBL; m++)
#pragma unroll
        for (int q = 0; q < RL / 2; q++) {               // first half only: the second half is the next transform's partner
          const float2 v = c[m * RL + q];
          const v2f o = { a.ampfac * v.x, a.ampfac * v.y };
          const size_t at = base + m * T + q * (N / RL);
          __builtin_nontemporal_store(o, reinterpret_cast<v2f *>(a.timf2w + at) + (unsigned int)tid);
          __builtin_nontemporal_store(o.x * o.x + o.y * o.y, (a.pwr + at) + (unsigned int)tid);
        }
    }
  }
}

// =====================================================================================================
// k_fft1v: the job of k_fft1w with both transforms laid out along the machine (round 4)
// =====================================================================================================
// k_fft1w moves every point through the LDS six times per block, each time behind two workgroup barriers, eight waves in lock step.
// Here the index of a transform is split the way the hardware is: 32 registers x 32 lanes x 2^B2 half-waves (B2 = log2 N - 10).
// Cooley-Tukey with n = T n1 + 32 n2 + n3, k = k1 + 32 k2 + T k3 (T = N/32 threads, n1 and n3 five bits, n2 B2 bits):
//   P1  32-point DFT over n1 -- the registers, as the coalesced load x[r] = in[tid + T r] leaves them;        times W_T^(n2 k1)
//   X1  the B2 bits of n2 (which half-wave) trade places with B2 register bits: the one exchange of the transform that crosses waves
//   P2  2^B2-point DFTs over n2
//   X2  the 5 bits of n3 (lanes 0..4) trade places with the registers: a 32 x 32 transpose inside each half-wave -- wave-private LDS,
//       no barrier, the waves drift apart;                                                                    times W_N^(n3 (k1 + 32 k2))
//   P3  32-point DFT over n3: bin k1 + 32 k2 + T k3, k3 in the registers, (k1, k2) = where the thread sits.
// The back transform starts from exactly that arrangement (its most significant digit is in the registers) and runs the mirror image --
// 32-point DFT, wave-private transpose, 32-point DFT, cross-wave exchange, 2^B2-point DFTs -- which ends with sample tid + T r in
// register r: the coalesced store.  Per block: two barriered exchanges and two wave-private ones (k_fft1w: six barriered), six passes
// (eight), five barriers (twelve).  Twiddles: three small tables in LDS (W_N^(m v) for v < T and m in {1,2,3,4,8,12,16}, stored at
// v ^ (v >> 5) so that both transforms read it without bank conflicts; W_T^(m n2); W_1024^(m m1)), combined by products of at most
// three factors.  The exchange buffer is unpadded: every access pattern below is conflict free by construction
// (ds_write_b64: 16 consecutive lanes on 16 consecutive cells; ds_read_b64: 32 lanes on 32 consecutive cells, or on cells q ^ lane).
// Results equal k_fft1w's to float32 rounding (different factorisation); same arguments, same sums, same rings.
// one SGPR base and a 32-bit byte offset in a VGPR: global_load / global_store in their saddr form, one v_add per access at most
// (a 64-bit pointer per access costs an SGPR pair each -- k_fft1v would need ~200 -- or two VALU adds)
template <typename V> __device__ __forceinline__ V gld(const void *base, unsigned int byte_off) { return *reinterpret_cast<const V *>(reinterpret_cast<const char *>(base) + byte_off); }
// LDS cells of the exchanges as single 8-byte accesses the compiler may not pair: ds_read2_b64 / ds_write2_b64 run at half the rate of
// ds_read_b64 and bank per 128 B instead of 256 B (MI355X_MICROARCH.md, LDS)
typedef unsigned long long lds_cell_t;
typedef __attribute__((address_space(3))) volatile lds_cell_t lds_vcell_t;
typedef __attribute__((address_space(3))) char lds_byte_t;
// addressed in BYTES from the start of the buffer: a per-thread base plus (or xor) a constant, nothing to shift per access
__device__ __forceinline__ void lds_put(float2 *lds, int byte, float2 v)
{
  lds_cell_t u; __builtin_memcpy(&u, &v, 8);
  *(lds_vcell_t *)((lds_byte_t *)lds + byte) = u;
}
__device__ __forceinline__ float2 lds_get(const float2 *lds, int byte)
{
  const lds_cell_t u = *(lds_vcell_t *)((lds_byte_t *)lds + byte);
  float2 v; __builtin_memcpy(&v, &u, 8); return v;
}
template <int LOG2N> struct Fft1vGeom {
  static constexpr int N = 1 << LOG2N, T = N / 32, B2 = LOG2N - 10, R2 = 1 << B2, KB = 5 - B2, NKB = 1 << KB;
  static constexpr int TN_ENT = 7;
  static constexpr int TN = N, T1 = TN + TN_ENT * T, TB1 = T1 + 10 * R2, LDS_CELLS = TB1 + 10 * 32;
  static_assert(B2 >= 2 && B2 <= 4, "fft1_size 4096, 8192, 16384");
};
__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
// w^n for n = 4a + b from the table values w^1..w^3 (wb) and w^4, w^8, .. (wa)
__device__ __forceinline__ float2 tw_pow(const float2 *wb, const float2 *wa, int n)
{
  const int a = n >> 2, b = n & 3;
  return a == 0 ? wb[b - 1] : (b == 0 ? wa[a - 1] : cmul(wa[a - 1], wb[b - 1]));
}
// EXP (diagnostics, LRH_FFT1V_EXP): bit 0 no workgroup barriers (timing experiment, wrong results), bit 1 shader-clock stamps at the phase
// boundaries (printed by lrh_make_timf2)
// KEEP: every bin of the spectrum goes to the fft1 ring (a.keep_spec), through one more LDS exchange; else the strong bins only
// REAL: real samples (fft1 version 2, fft1_reherm_dit_one, fft1_re.c:32-131): the pair (x[2n], x[2n+1]) is one complex point, each component with
// its own window value, nothing negated; the N-point transform S is followed by one more exchange -- bin k needs S[N-k], which another thread holds --
// and the even / odd split of k_realsplit; array index = transform index (no N/2 shift).  fft1_direction > 0 only.
template <int LOG2N, bool DW, bool KEEP, int EXP = 0, bool REAL = false>
__global__ __launch_bounds__((1 << LOG2N) / 32, 1) void k_fft1v(Fft1wArgs a)
{
  using G = Fft1vGeom<LOG2N>;
  using Raw = typename std::conditional<DW, int2, short2>::type;
  constexpr int P = 32, N = G::N, T = G::T, R2 = G::R2, KB = G::KB, NKB = G::NKB, HP = P / 2, B2_ = G::B2;
  using F32 = SDft<+1, 32>; using F2 = SDft<+1, R2>; using B32 = SDft<-1, 32>; using B2 = SDft<-1, R2>;
  __shared__ float2 lds[G::LDS_CELLS];
  const int tid0 = threadIdx.x;
  const int r0 = (int)blockIdx.x * a.run, r1 = min(r0 + a.run, a.batch);
  if (r0 >= r1) return;
  for (int idx = tid0; idx < G::TN_ENT * T; idx += T) {
    const int e = idx / T, v = idx % T, mult = e < 3 ? e + 1 : 4 * (e - 2);
    lds[G::TN + e * T + (v ^ (v >> 5))] = tw_dir<+1>(a.tw[mult * v]);
  }
  for (int idx = tid0; idx < 10 * R2; idx += T) {
    const int e = idx / R2, v = idx % R2, mult = e < 3 ? e + 1 : 4 * (e - 2);
    lds[G::T1 + idx] = tw_dir<+1>(a.tw[(mult * v * 32) & (N - 1)]);
  }
  for (int idx = tid0; idx < 320; idx += T) {
    const int e = idx / 32, v = idx % 32, mult = e < 3 ? e + 1 : 4 * (e - 2);
    lds[G::TB1 + idx] = tw_dir<+1>(a.tw[(mult * v * (N / 1024)) & (N - 1)]);
  }
  // where the thread sits after the forward transform: bins kk + T j (fft1_float order, DC at N/2), j in the registers
  auto kk_of = [](int tid) { const int lam = tid & 31; return (tid >> 5) + R2 * (lam & (NKB - 1)) + 32 * (lam >> KB); };
  unsigned int wk_cur[2], wk_first[2];
  { const int kk = kk_of(tid0);
    wk_cur[0] = a.pack_cur[kk]; wk_cur[1] = a.pack_cur[kk + T]; wk_first[0] = a.pack_prev[kk]; wk_first[1] = a.pack_prev[kk + T]; }
  auto weak_bit = [](const unsigned int (&wk)[2], int j) { return (wk[j & 1] >> (j >> 1)) & 1u; };
  auto barrier = []() { if constexpr (!(EXP & 1)) __syncthreads(); };
  // EXP == 2: shader-clock stamps of waves 0 and 4 of workgroup 0, second block of its run, at the phase boundaries
  int nstamp = 0; bool stamping = false;
  auto stamp = [&]() {
    if constexpr ((EXP & 2) != 0) {
      if (stamping && a.stamps && (tid0 == 0 || tid0 == 256) && nstamp < 32) a.stamps[(tid0 ? 32 : 0) + nstamp++] = __builtin_amdgcn_s_memtime();
    }
  };

  Raw raw[P];
  float win[P];
  // (part, parts): the loads are spread over the back transform, a few at a time, so that the memory pipeline never holds a wave up at issue
  auto fetch = [&](int b, int tid, int part = 0, int parts = 1) {
    const int p0 = a.p0_first + b * a.step;
#pragma unroll
    for (int i = part * (P / parts); i < (part + 1) * (P / parts); i++) {   // in the order the first butterflies take them
      const int r = F32::in(i);
      const int n = p0 + tid + r * T;
      raw[r] = ((const Raw *)a.timf1)[(n * a.chan_count + a.chan_index) & a.ring_mask];
    }
  };
  // The kernel runs with the sin^2 window only (make_timf2's overlap form, timf2.c:1003-1026), whose value at sample tid + T r is
  // c sin^2(pi (tid + T r) / N) = c/2 (1 - cos(2 pi tid / N + 2 pi r / 32)): two fused multiply-adds on one complex number per thread,
  // e^(2 pi j tid / N), and the 32nd roots of unity -- no table loads, no 32 registers held over the run.  c is the table's own peak
  // value (make_window's normalisation, fft0.c:812-921); the values equal the table's to float32 rounding.
  const float whalf = REAL ? 0.5f * a.real_peak : 0.5f * a.window[N / 2];
  auto window_at = [&](float2 wrot, int r) {             // wrot = (cos, sin)(2 pi tid / N): the first entry of the LDS table; compile-time r
    const float cr = lrh_cos32(r), sr = lrh_sin32(r);
    return whalf - whalf * (wrot.x * cr - wrot.y * sr);
  };
  // REAL: sample i of the 2N takes the half window's value at min(i, 2N - 1 - i) (fft1_re.c:47-57), c sin^2(pi i' / 2N): for the point
  // n = tid + T r the two components sit at the angles theta + 2 pi r / 32 with theta = 2 pi tid / N (x[2n]) and theta + pi / N (x[2n+1]) in the
  // first half, theta + pi / N and theta + 2 pi / N in the second: three rotations per thread, kept over the run
  // (theta + pi / N and theta + 2 pi / N: the table's e^(j theta) turned by constants -- nothing kept over the run)
  const float cpn1 = __builtin_cosf(3.14159265358979323846f / N), spn1 = __builtin_sinf(3.14159265358979323846f / N);
  const float cpn2 = __builtin_cosf(2 * 3.14159265358979323846f / N), spn2 = __builtin_sinf(2 * 3.14159265358979323846f / N);
  // ... and e^(-j pi kk / N), the thread's part of the split's twiddle e^(-j pi k / N), k = kk + T k3
  float2 wsplit = make_float2(1.f, 0.f);
  if constexpr (REAL) { float sn, cs; sincospif((float)kk_of(tid0) / (float)N, &sn, &cs); wsplit = make_float2(cs, sn); }
  (void)win;
  // LDS byte addresses of the two kinds of exchange (see the header comment).  cross: cell (q2 32 + q) 32 + l for the element that sits in
  // half-wave q2, register q, lane l before the exchange; wave: cell 32 l + (q ^ l) of the half-wave's 1024
  auto cross_wr = [](int t) { return 8 * ((t >> 5) * 1024 + (t & 31)); };
  auto wave_wr = [](int t) { return 8 * ((t >> 5) * 1024 + (t & 31) * 33); };
  auto wave_rd = [](int t) { return 8 * ((t >> 5) * 1024 + (t & 31)); };
  // second stage of a 32-point pass: hands out the finished values group by group, f(index, value)
  auto second32 = [&](auto tag, lrh_v2f (&u)[P], auto f, auto half_way) {
    using D = decltype(tag);
#pragma unroll
    for (int c = 0; c < D::NC; c++) {
      if (c == D::NC / 2) half_way();
      lrh_v2f y[D::ND];
      D::b(u, c, y);
#pragma unroll
      for (int d = 0; d < D::ND; d++) f(D::out(c, d), y[d]);
    }
  };
  // the NKB 2^B2-point transforms of a thread: first stages, then (after `between`) second stages handing out f(kb, index, value)
  auto small_pass = [&](auto tag, lrh_v2f (&u)[P], auto between, auto f) {
    using D = decltype(tag);
#pragma unroll
    for (int kb = 0; kb < NKB; kb++) {
      lrh_v2f t[R2];
#pragma unroll
      for (int i = 0; i < R2; i++) t[i] = u[kb * R2 + i];
      D::a(t);
#pragma unroll
      for (int i = 0; i < R2; i++) u[kb * R2 + i] = t[i];
    }
    between();
#pragma unroll
    for (int kb = 0; kb < NKB; kb++) {
      lrh_v2f t[R2];
#pragma unroll
      for (int i = 0; i < R2; i++) t[i] = u[kb * R2 + i];
#pragma unroll
      for (int c = 0; c < D::NC; c++) {
        lrh_v2f y[D::ND];
        D::b(t, c, y);
#pragma unroll
        for (int d = 0; d < D::ND; d++) f(kb, D::out(c, d), y[d]);
      }
    }
  };

  // forward transform of the fetched samples; hands out f(k3, value): bare-transform bin kk + T k3.  `early` runs before the last pass
  // (the place for loads the caller needs right after it)
  auto forward = [&](int tid, auto after_load, auto early, auto half_way, auto f) {
    lrh_v2f x[P];
    { int t_ = tid; asm volatile("" : "+v"(t_));
      const float2 wrot = lds[G::TN + (t_ ^ (t_ >> 5))];
#pragma unroll
      for (int r = 0; r < P; r++) {
        if constexpr (REAL) {
#pragma clang fp contract(off)                               // (the prologue and the loop are two copies of this code: every product and sum rounded on its own, so that both round alike)
          const float2 wr1 = make_float2(wrot.x * cpn1 - wrot.y * spn1, wrot.y * cpn1 + wrot.x * spn1), wr2 = make_float2(wrot.x * cpn2 - wrot.y * spn2, wrot.y * cpn2 + wrot.x * spn2);
          const float w0 = window_at(r < HP ? wrot : wr1, r), w1 = window_at(r < HP ? wr1 : wr2, r);
          x[r] = lrh_v2f{(float)raw[r].x * w0, (float)raw[r].y * w1};
        } else { const float w = window_at(wrot, r); x[r] = lrh_v2f{(float)raw[r].x * w, -((float)raw[r].y * w)}; }   // Q negated (fft1.c:432-447)
      } }
    after_load();
    stamp();
    F32::a(x);
    stamp();
    barrier();                                           // B_a: every wave is through with the previous block's exchange buffer
    stamp();
    { int t_ = tid; asm volatile("" : "+v"(t_));
      const float2 *tb = lds + G::T1 + (t_ >> 5);
      float2 wb[3], wa[7];
#pragma unroll
      for (int e = 0; e < 3; e++) wb[e] = tb[e * R2];
#pragma unroll
      for (int e = 0; e < 7; e++) wa[e] = tb[(3 + e) * R2];
      const int wr = cross_wr(t_);
      second32(F32(), x, [&](int k1, lrh_v2f v) { lds_put(lds, wr + k1 * 256, to_f2(k1 == 0 ? v : cmul_v(v, to_v(tw_pow(wb, wa, k1))))); }, []() {}); }
    stamp();
    barrier();                                           // B_b
    stamp();
    { int t_ = tid; asm volatile("" : "+v"(t_));
      const int rd0 = 8 * t_, rd1 = 8 * t_ + 8 * 8192;   // (a ds offset reaches 64 KB)
#pragma unroll
      for (int kb = 0; kb < NKB; kb++)
#pragma unroll
        for (int i = 0; i < R2; i++) { const int v = F2::in(i); x[kb * R2 + v] = to_v(lds_get(lds, (v < 8 ? rd0 : rd1) + 8 * ((v & 7) * 1024 + kb * R2 * 32))); } }
    { int t_ = tid; asm volatile("" : "+v"(t_));
      const int wb_ = wave_wr(t_);
      small_pass(F2(), x, [&]() { stamp(); barrier(); stamp(); },          // B_c: the cross-wave reads are done, the half-waves take their cells back
                 [&](int kb, int k2, lrh_v2f v) { lds_put(lds, wb_ ^ (8 * (kb + NKB * k2)), to_f2(v)); }); }
    stamp();
    wave_lds_sync();
    { int t_ = tid; asm volatile("" : "+v"(t_));
      const int rb = wave_rd(t_);
      const int kk = kk_of(t_);
      const float2 *tb = lds + G::TN + (kk ^ (kk >> 5));
      float2 wb[3], wa[3];
#pragma unroll
      for (int e = 0; e < 3; e++) { wb[e] = tb[e * T]; wa[e] = tb[(3 + e) * T]; }
      const float2 w16 = tb[6 * T];
#pragma unroll
      for (int i = 0; i < P; i++) {
        const int n3 = F32::in(i);
        const lrh_v2f v = to_v(lds_get(lds, rb ^ (8 * 33 * n3)));
        if (n3 == 0) x[n3] = v;
        else if (n3 == 16) x[n3] = cmul_v(v, to_v(w16));
        else if (n3 < 16) x[n3] = cmul_v(v, to_v(tw_pow(wb, wa, n3)));
        else x[n3] = cmul_v(v, cmul_v(to_v(w16), to_v(tw_pow(wb, wa, n3 - 16))));
      } }
    if constexpr (!REAL) early();
    stamp();
    F32::a(x);
    stamp();
    if constexpr (!REAL) second32(F32(), x, f, half_way);
    else {
      // S[kk + T k3] is in the registers; bin k wants S[N - k] as well: through the half-wave's own cells (32 k3 + lane: nobody else
      // writes there), read back from the cells of the thread that holds kk' = (T - kk) mod T -- a lane permutation of another half-wave,
      // so both directions touch 32 consecutive cells per half-wave
      lrh_v2f z[P];                                      // the thread's own S[kk + T k3] stays in the registers (x is dead once the second stage has run)
      { int t_ = tid; asm volatile("" : "+v"(t_));
        const int wr = 8 * ((t_ >> 5) * 1024 + (t_ & 31));
        second32(F32(), x, [&](int k3, lrh_v2f v) { z[k3] = v; lds_put(lds, wr + 256 * k3, to_f2(v)); }, []() {}); }
      early();                                           // (the filter table's first half: asked for here, where x is gone)
      barrier();                                         // B_f: every half-wave's S is in its cells
      { int t_ = tid; asm volatile("" : "+v"(t_));
        const int kk = kk_of(t_);
        const int km = (T - kk) & (T - 1);               // kk' of the mirror bin; its owner: kk = hw | (lam & (NKB-1)) R2 ... inverted below
        const int ot = (km & (R2 - 1)) * 32 + ((km >> B2_) & (NKB - 1)) + NKB * (km >> 5);
        const int mir = 8 * ((ot >> 5) * 1024 + (ot & 31));
        const bool kk0 = kk == 0;
#pragma unroll
        for (int i = 0; i < P; i++) {
#pragma clang fp contract(off)                               // (as above: no fused multiply-adds the two copies of this loop could place differently)
          const int k3 = F32::out(i / F32::ND, i % F32::ND);    // the order the complex form hands them out in (the filter table's loads follow it)
          const int m3 = 31 - k3, m0 = (32 - k3) & 31;              // k3 of bin N - k: 31 - k3 (kk != 0), (32 - k3) mod 32 (kk == 0)
          if (i == P / 2) half_way();                      // (the even pairs of the filter table are used up by now)
          const float2 sa = to_f2(z[k3]);
          const float2 sb = lds_get(lds, kk0 ? mir + 256 * m0 : mir + 256 * m3);
          // k_realsplit's arithmetic: k < N/2: (sa, sb) = (S[k], S[N-k]) -> Z_k; k > N/2: the pair the other way round, angle pi (N-k) / N -> Z_(N-k') = conj(E - t)
          const float cr = lrh_cos64(k3), sr = lrh_sin64(k3);        // e^(j pi k3 / 32)
          const float cs = wsplit.x * cr - wsplit.y * sr, sn = wsplit.y * cr + wsplit.x * sr;   // (cos, sin)(pi k / N)
          float2 o;
          if (k3 < HP) {
            const float ex = 0.5f * (sb.x + sa.x), ey = 0.5f * (sb.y - sa.y), dx = 0.5f * (sb.x - sa.x), dy = 0.5f * (sb.y + sa.y);
            const float ox = dy, oy = -dx;
            const float tx = cs * ox + sn * oy, ty = cs * oy - sn * ox;
            o = make_float2(ey + ty, ex + tx);             // (Im Z_k, Re Z_k)
            if (k3 == 0 && kk0) o = make_float2(sa.x - sa.y, sa.x + sa.y);    // (Z_N, Z_0)
          } else {
            const float ex = 0.5f * (sa.x + sb.x), ey = 0.5f * (sa.y - sb.y), dx = 0.5f * (sa.x - sb.x), dy = 0.5f * (sa.y + sb.y);
            const float ox = dy, oy = -dx;
            const float tx = -cs * ox + sn * oy, ty = -cs * oy - sn * ox;      // angle pi - pi k / N
            o = make_float2(-(ey - ty), ex - tx);          // (Im, Re) of conj(E - t)
          }
          f(k3, to_v(o));
        } }
      barrier();                                         // B_g: the cells go back to their half-waves (back transform's first exchange)
    }
    stamp();
  };
  // back transform of the spectrum c[j] (bin kk + T j); hands out f(r, value): sample tid + T r of the transform (r = 0 .. 31)
  auto backward = [&](lrh_v2f (&c)[P], int tid, auto mid, auto before_last, auto f) {
    B32::a(c);
    mid(0);
    stamp();
    wave_lds_sync();
    { int t_ = tid; asm volatile("" : "+v"(t_));
      const int wb_ = wave_wr(t_);
      second32(B32(), c, [&](int m1, lrh_v2f v) { lds_put(lds, wb_ ^ (8 * m1), to_f2(v)); }, []() {}); }
    stamp();
    wave_lds_sync();
    { int t_ = tid; asm volatile("" : "+v"(t_));
      const int rb = wave_rd(t_);
      const float2 *tb = lds + G::TB1 + (t_ & 31);
      float2 wb[3], wa[7];
#pragma unroll
      for (int e = 0; e < 3; e++) wb[e] = tb[e * 32];
#pragma unroll
      for (int e = 0; e < 7; e++) wa[e] = tb[(3 + e) * 32];
#pragma unroll
      for (int i = 0; i < P; i++) {
        const int b = B32::in(i);
        const lrh_v2f v = to_v(lds_get(lds, rb ^ (8 * 33 * b)));
        c[b] = b == 0 ? v : cmul_conj_v(v, to_v(tw_pow(wb, wa, b)));
      } }
    stamp();
    mid(1);
    B32::a(c);
    stamp();
    barrier();                                           // B_d: every half-wave has read its transpose
    stamp();
    mid(2);
    { int t_ = tid; asm volatile("" : "+v"(t_));
      const int wr = cross_wr(t_);
      second32(B32(), c, [&](int m2, lrh_v2f v) { lds_put(lds, wr + m2 * 256, to_f2(v)); }, []() {}); }
    stamp();
    barrier();                                           // B_e
    stamp();
    { int t_ = tid; asm volatile("" : "+v"(t_));
      const int rd0 = 8 * t_, rd1 = 8 * t_ + 8 * 8192;
      const float2 *tb = lds + G::TN + (t_ ^ (t_ >> 5));
      float2 wb[3], wa[3];
#pragma unroll
      for (int e = 0; e < 3; e++) { wb[e] = tb[e * T]; wa[e] = tb[(3 + e) * T]; }
#pragma unroll
      for (int mb = 0; mb < NKB; mb++)
#pragma unroll
        for (int i = 0; i < R2; i++) {
          const int cc = B2::in(i);
          const lrh_v2f v = to_v(lds_get(lds, (cc < 8 ? rd0 : rd1) + 8 * ((cc & 7) * 1024 + mb * R2 * 32)));
          if (cc == 0) { c[mb * R2 + cc] = v; continue; }
          const int e = (cc * mb) & 31;                  // W_N^(cc T mb) = a 32nd root of unity
          const lrh_v2f w = to_v(tw_pow(wb, wa, cc));
          c[mb * R2 + cc] = cmul_conj_v(v, e == 0 ? w : cmulc_v(w, lrh_cos32(e), lrh_sin32(e)));
        } }
    stamp();
    mid(3);
    before_last();
    small_pass(B2(), c, []() {}, [&](int mb, int m3, lrh_v2f v) { f(mb + NKB * m3, v); });
    stamp();
  };

  // The sin^2 overlap (timf2.c:1003-1026: the first half of block t is added onto the second half of block t-1) is carried in the time
  // domain like the reference does: ov = second half of the previous weak back transform, P/2 samples per thread.  (k_fft1w carries the
  // previous weak SPECTRUM and transforms S_t + (-1)^k S_(t-1): twice the registers.)
  lrh_v2f ov[HP];
  float acc[P];
#pragma unroll
  for (int e = 0; e < HP; e++) ov[e] = lrh_v2f{0.f, 0.f};
#pragma unroll
  for (int e = 0; e < P; e++) acc[e] = 0.f;
  __syncthreads();                                       // twiddle tables are in place
  // All workgroups start together and take the same time per block: left alone, every CU of the chip asks the memory system for its
  // samples at the same moment and stores its results at the same moment, and in between the memory idles.  A start delay that differs
  // from workgroup to workgroup spreads the phases over the block time.
  for (int i = (int)(blockIdx.x & 15u) * a.stagger; i > 0; i--) __builtin_amdgcn_s_sleep(32);   // 32 x 64 cycles
  if (r0 > 0 || a.have_prev) {                           // the block before the run once more: its second half is the first block's partner
    fetch(r0 - 1, tid0);
    lrh_v2f c[P];
    forward(tid0, []() {}, []() {}, []() {}, [&](int k3, lrh_v2f v) {
      const int j = REAL ? k3 : k3 ^ 16;
      const bool weak = r0 > 0 ? weak_bit(wk_cur, j) : weak_bit(wk_first, j);   // routed with the table in force for that transform
      c[j] = weak ? cmul_v(v, to_v(gld<float2>(a.filtercorr_v, 8u * (unsigned int)(2 * (tid0 + (j >> 1) * T) + (j & 1))))) : lrh_v2f{0.f, 0.f};
    });
    backward(c, tid0, [](int) {}, []() {}, [&](int r, lrh_v2f v) { if (r >= HP) ov[r - HP] = v; });
  }
  fetch(r0, tid0);
  // Retire the prologue's loads here (see k_fft1): the compiler merges the wait counts of both loop entries, and with the first fetch the
  // youngest memory operation on this path the loop-top wait for the samples would allow only a handful of operations in flight on every
  // trip -- i.e. wait for the previous block's STORES to be acknowledged (measured: 7-10 k cycles of a 48 k block)
  __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0), other counters untouched
#pragma unroll 1
  for (int b = r0; b < r1; b++) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));                        // keep index math inside the loop (see k_timf2)
    stamping = blockIdx.x == 0 && b == r0 + 1;
    stamp();
    const int kk = kk_of(tid);
    float4 fc[P / 2];                                    // filter correction of the bins kk + T j, two per load, on its way behind the last forward pass
    lrh_v2f c[P];
    float2 *const out0 = a.spec + (size_t)((a.first_nb + b) & a.nb_mask) * N;
    // (the last pass hands out k3 = c + 4 d group by group: c = 0, 1 are the even pairs j / 2, c = 2, 3 the odd ones)
    forward(tid, []() {}, [&]() {
#pragma unroll
      for (int jp = 0; jp < P / 2; jp += 2) fc[jp] = gld<float4>(a.filtercorr_v, 16u * (unsigned int)(tid + jp * T));
    }, [&]() {
#pragma unroll
      for (int jp = 1; jp < P / 2; jp += 2) fc[jp] = gld<float4>(a.filtercorr_v, 16u * (unsigned int)(tid + jp * T));
    }, [&](int k3, lrh_v2f xv) {
      const int j = REAL ? k3 : k3 ^ 16;                 // fft1_float bin kk + T j (DC at N/2; real input: natural order)
      const lrh_v2f v = cmul_v(xv, (j & 1) ? lrh_v2f{fc[j >> 1].z, fc[j >> 1].w} : lrh_v2f{fc[j >> 1].x, fc[j >> 1].y});
      // fft1_c: sum |X|^2 over the averaging group (fft1.c:4115-4171), order as k_timf2<.., SS>; the contraction written out, so that every
      // instantiation of this kernel rounds alike (the full and the sparse ring must not differ downstream)
      acc[j] += __builtin_fmaf(v.y, v.y, v.x * v.x);
      const bool weak = weak_bit(wk_cur, j);
      if constexpr (KEEP) { c[j] = v; return; }          // every bin goes to the ring: below, through the LDS
      if (!weak) store_stream(reinterpret_cast<float2 *>(reinterpret_cast<char *>(out0) + 8u * (unsigned int)(kk + j * T)), to_f2(v));   // the strong bins: a handful
      c[j] = weak ? v : lrh_v2f{0.f, 0.f};               // the weak stream's spectrum
    });
    if constexpr (KEEP) {
      // fft1_float in full (cfg.fft1_float_sparse = 0: Linrad's graphs, fft1_mix1_*, the AFC, NET_RXOUT_FFT1 read it).  A thread holds bins
      // kk + T j, of which a wave's lanes cover pairs 16 bins apart: stored from here a wave-instruction would touch 32 cache lines for 16
      // bytes each.  One more trip through the exchange buffer -- cell f ^ ((f >> 5) & 15), free of bank conflicts on the way out and at
      // most 2-way on the way in -- and every store is 64 consecutive bins.
      auto cell = [](int f) { return 8 * (f ^ ((f >> 5) & 15)); };
      barrier();
      { int t_ = tid; asm volatile("" : "+v"(t_));
        const int kk_ = kk_of(t_);
#pragma unroll
        for (int j = 0; j < P; j++) lds_put(lds, cell(kk_ + j * T), to_f2(c[j])); }
      barrier();
      { int t_ = tid; asm volatile("" : "+v"(t_));
#pragma unroll
        for (int r = 0; r < P; r++)
          store_stream(reinterpret_cast<float2 *>(reinterpret_cast<char *>(out0) + 8u * (unsigned int)(t_ + r * T)), lds_get(lds, cell(t_ + r * T))); }
      barrier();                                         // the half-waves take their cells back (back transform, first exchange)
#pragma unroll
      for (int j = 0; j < P; j++) if (!weak_bit(wk_cur, j)) c[j] = lrh_v2f{0.f, 0.f};
    }
    {
      const int gb = b + a.ss_c0, g = gb / a.ss_avg;
      const bool group_end = gb - g * a.ss_avg == a.ss_avg - 1;
      if (group_end || b == r1 - 1) {
        const bool head = g * a.ss_avg - a.ss_c0 < r0;   // began in an earlier run (or an earlier call)
        float *dst = (!head && group_end) ? a.ss_ring + ((a.ss_pa0 + g * N) & a.ss_mask) : a.ss_part + (size_t)(2 * blockIdx.x + (head ? 0 : 1)) * N;
#pragma unroll
        for (int j = 0; j < P; j++) { *reinterpret_cast<float *>(reinterpret_cast<char *>(dst) + 4u * (unsigned int)(kk + j * T)) = acc[j]; acc[j] = 0.f; }
      }
    }
    stamp();
    {
      const size_t base = (size_t)((a.pa_first + b * a.step) & a.mask);
      typedef float v2f __attribute__((ext_vector_type(2)));
      // the next block's samples are asked for ahead of this block's stores: vmcnt retires in order, and a wait of the next forward
      // transform for its samples must not cover a store
      // int16: the next block's samples are asked for in the middle of the back transform (32 registers; a quarter of a block of time for
      // the memory); int32 (64 registers): ahead of its last pass.  Unconditional: the last trip re-reads its own samples
      backward(c, tid, [&](int part) { if constexpr (!DW) fetch(min(b + 1, r1 - 1), tid, part, 4); },
               [&]() { if constexpr (DW) fetch(min(b + 1, r1 - 1), tid); }, [&](int r, lrh_v2f v) {
        if (r >= HP) { ov[r - HP] = v; return; }         // this block's second half waits for the next
        const lrh_v2f o = (v + ov[r]) * a.ampfac;        // first half + the previous block's second half
        const v2f ov_ = { o.x, o.y };
        const unsigned int at = (unsigned int)(tid + r * T);
        __builtin_nontemporal_store(ov_, reinterpret_cast<v2f *>(reinterpret_cast<char *>(a.timf2w + base) + 8u * at));
        __builtin_nontemporal_store(o.x * o.x + o.y * o.y, reinterpret_cast<float *>(reinterpret_cast<char *>(a.pwr + base) + 4u * at));
      });
    }
    stamp();
    stamping = false;
  }
}

// =====================================================================================================
// blanker
// =====================================================================================================
// The reference scan is serial (blank1.c:1023-1086): a run of samples above the limit is cleared, and when it
// ends a guard of i_after samples *ahead* is zeroed too, so later samples are tested against modified data.
// Exact parallel form: a lane owns LRH_BLN_CHUNK samples but starts its replay at the nearest earlier point
// where the serial state is provably clean (G >= clr2 consecutive samples at or below the limit, or the start
// of the call), so every decision equals the serial one.  Decisions go to a bit mask; k_blank_apply zeroes the
// data afterwards (the scan itself only reads), which also keeps lanes from racing on guard samples.
#define LRH_BLN_CHUNK 64
// How far back a lane searches for its clean restart point.  In the normal regime the point is one or two samples away.
// A miss hands the call to the long-run replay below (uncalibrated blanker: parallel, k_blank_runs) or to the one-thread
// serial pass (calibrated); searching further back only pays while a miss is expensive.
#define LRH_BLN_BACK 256
#define LRH_BLN_BACK2 4096

__device__ __forceinline__ void bln_setbit(unsigned int *bits, int p) { atomicOr(&bits[p >> 5], 1u << (p & 31)); }

__device__ __forceinline__ int bln_guards(const BlankArgs &a, float pulmax, float totnoise, int *ib, int *ia)
{
  float t1 = pulmax / totnoise;
  *ib = 0; *ia = 0;
  if (a.clr1 == 0 && a.clr2 == 1) {
    // Without pulse calibration (blank1.c:1013-1014) the general form below reduces to: one sample behind the run when the
    // ratio reaches 2500, none before it.  Exactly: ia = (int)((float)(sqrt((double)t1) / 100) + 0.5) is 1 iff the float
    // quotient is >= 0.5f, i.e. sqrt(t1) >= 50 - 100 * 2^-26 (the tie rounds to even, 0.5f), i.e. t1 >= 2500 - 1.5e-4; the
    // float below 2500 is 2500 - 2.4e-4, so the threshold in float is 2500 itself.  No double-precision square root per run.
    if (t1 > 4) { *ia = t1 >= 2500.0f ? 1 : 0; return 1; }
    return 0;
  }
  if (t1 > 4) {
    if (t1 > 10000) t1 = 10000;                        // 40 dB cap (blank1.c:1056-1057)
    // (int)(clr * (float)(sqrt((double)t1) / 100) + 0.5): the double square root and division are ~150 instructions, and the serial walk
    // pays them per run.  The single-precision value differs from the reference's by a few ulp at most, which changes the truncated
    // result only when clr * x + 0.5 lies within that of an integer: decided in float unless it is that close (then exactly as written).
    const float xf = sqrtf(t1) * 0.01f;
    const float fb = (float)a.clr1 * xf + 0.5f, fa = (float)a.clr2 * xf + 0.5f;
    const float db = fb - floorf(fb), da = fa - floorf(fa);
    if (a.clr2 <= 256 && db > 1e-3f && db < 1.f - 1e-3f && da > 1e-3f && da < 1.f - 1e-3f) { *ib = (int)fb; *ia = (int)fa; return 1; }   // (clr x 4 ulp stays far below the 1e-3 margin)
    t1 = (float)(sqrt((double)t1) / 100);
    *ib = (int)((float)a.clr1 * t1 + 0.5);
    *ia = (int)((float)a.clr2 * t1 + 0.5);
    return 1;
  }
  return 0;
}

#define LRH_BLN_TILE (256 * LRH_BLN_CHUNK)
#define LRH_BLN_WORDS (LRH_BLN_TILE / 32 + 2)

// BACK: how far before a wave's span the powers are staged for the lanes' search of a clean restart point.  LRH_BLN_BACK (256) in the
// scan every call runs; LRH_BLN_BACK2 (4096) in a second launch that returns at once unless a lane of the first gave up, and then redoes
// the call with the long reach (same bits where the first succeeded: the decisions do not depend on where a replay starts).  With the
// limit inside the noise -- the start-up of a calibrated receiver -- a clean point (clr2 samples in a row at or below the limit) is some
// 200 samples away on average and every few hundredth lane finds none within 256: that call used to go to the serial walk as a whole.
template <int BACK>
__global__ __launch_bounds__(256) void k_blank_scan(BlankArgs a)
{
  if constexpr (BACK != LRH_BLN_BACK) { if (!a.st->need_slow || a.debug == 8) return; }
  // Two phases per workgroup (tile of 256 chunks x 64 samples).
  // 1. Cooperative, coalesced: every wave reads its 4096 samples (and the 256 before them, for the clean-point search) as
  //    256-byte rows, one dword per lane, and turns each row into one 64-bit word of "above the limit" bits by ballot.  This
  //    is the only pass over the power ring: 134 MB per 33.5 M samples, every line fetched once.  (Before: a lane walked its
  //    own 64-sample chunk with 16-byte loads at a 256-byte lane stride -- 64 lines per wave instruction, each line requested
  //    in two loop trips; FETCH_SIZE 1.6x the ring span.)
  // 2. Per lane, exact replay of the serial scan (blank1.c:1023-1086) over its own chunk, event driven on the bit words: only
  //    samples above the limit and the sample that ends a run matter, everything else leaves the serial state alone.  The
  //    power of the few samples above the limit is read back from L2 for the run maximum.
  __shared__ unsigned int wbits[LRH_BLN_WORDS];         // decisions for the ring words this tile overlaps
  __shared__ unsigned long long above[4][LRH_BLN_CHUNK + BACK / 64];   // per wave: word k covers positions R0 - BACK + 64 k ..
  // the power of the samples above the limit, compacted per wave in row order (phase 2 needs it for the run maxima and would
  // otherwise wait for L2 once per event): value of bit l of row k sits at rowoff[k] + popcount(bits of the row below l)
  constexpr int VCAP = 512;                               // (with 1024 the workgroup took 22 KB of LDS: seven per CU, and the eighth of every CU ran alone in a second round)
  __shared__ float vals[4][VCAP];
  __shared__ int rowoff[4][LRH_BLN_CHUNK + BACK / 64];
  __shared__ int wg_cnt;
  __shared__ double wg_sum[4];
  if (threadIdx.x == 0) wg_cnt = 0;
  for (int i = threadIdx.x; i < LRH_BLN_WORDS; i += 256) wbits[i] = 0;
  const int qt = blockIdx.x * LRH_BLN_TILE;              // first sequence position of this workgroup's tile
  // ring word that holds the tile's first position; bits of positions inside [w0*32, (w0+WORDS)*32) go to LDS
  const int w0 = ((a.pbeg + qt) & a.mask) >> 5;
  const int nwords_ring = (a.mask + 1) >> 5;
  auto setbit = [&](int q) {
    const int p = (a.pbeg + q) & a.mask;
    const int w = ((p >> 5) - w0 + nwords_ring) & (nwords_ring - 1);
    if (w < LRH_BLN_WORDS) atomicOr(&wbits[w], 1u << (p & 31));
    else atomicOr(&a.mask_bits[p >> 5], 1u << (p & 31));  // guard reaching into a neighbour tile (rare)
  };
  const float nfl = (float)a.st->limit, totnoise = (float)(a.st->noise_floor * (a.chans > 1 ? a.chans : 1));   // blank1.c:1017
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int NBACK = BACK / 64, NROWS = LRH_BLN_CHUNK + NBACK;
  const int base = qt + wv * 64 * LRH_BLN_CHUNK - BACK;          // position of bit 0 of word 0 of this wave
  double s4 = 0;                                         // every-4th-sample power of the wave's own positions before clearing
  int running = 0;                                       // wave-uniform: samples above the limit in the rows so far
  const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
  constexpr int FL = NROWS % 17 == 0 ? 17 : 16;          // rows in flight per trip (NROWS = 68 = 4 x 17; 128 = 8 x 16 with the long reach)
  static_assert(NROWS % FL == 0, "row count");
#pragma unroll 1
  for (int k = 0; k < NROWS; k += FL) {
    float v[FL]; bool ok[FL];
#pragma unroll
    for (int u = 0; u < FL; u++) {
      const int q = base + 64 * (k + u) + lane;
      ok[u] = q >= 1 && q <= a.total;
      v[u] = ok[u] ? a.pwr[(a.pbeg + q) & a.mask] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < FL; u++) {
      const bool up = ok[u] && v[u] > nfl;
      const unsigned long long b = __ballot(up);
      if (lane == 0) { above[wv][k + u] = b; rowoff[wv][k + u] = running; }
      if (up) { const int at = running + __popcll(b & below); if (at < VCAP) vals[wv][at] = v[u]; }
      running += __popcll(b);
      if (k + u >= NBACK && ok[u] && (lane & 3) == 0) s4 += (double)v[u];     // positions 4, 8, .. of the own span (base is a multiple of 64)
    }
  }
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int cb = c * LRH_BLN_CHUNK;
  const int cs = max(cb, 1);
  const int ce = min(cb + LRH_BLN_CHUNK - 1, a.total);
  auto bit = [&](int q) -> bool { const int r = q - base; return (above[wv][r >> 6] >> (r & 63)) & 1ull; };   // base <= q <= ce
  // power of a sample that is above the limit (base <= q <= ce): from the wave's compact table, from the ring when that overflowed
  auto sample = [&](int q) -> float {
    const int r = q - base, l = r & 63;
    const int at = rowoff[wv][r >> 6] + __popcll(above[wv][r >> 6] & (l ? (~0ull >> (64 - l)) : 0ull));
    return at < VCAP ? vals[wv][at] : a.pwr[(a.pbeg + q) & a.mask];
  };
  // first position >= q above the limit, ce + 1 when there is none up to ce
  auto next_above = [&](int q) -> int {
    while (q <= ce) {
      const int r = q - base;
      const unsigned long long w = above[wv][r >> 6] >> (r & 63);
      if (w) { const int n = q + __ffsll((long long)w) - 1; return n <= ce ? n : ce + 1; }
      q += 64 - (r & 63);
    }
    return ce + 1;
  };
  const int G = max(a.clr2, 1);
  int s = 1;
  bool live = cs <= ce;
  if (live) {                                            // nearest earlier point where the serial state is provably clean
    int run = 0, steps = 0, q = cs - 1; bool found = false;
    while (q >= 1) {
      if (bit(q)) run = 0; else if (++run >= G) { found = true; break; }
      q--;
      if (++steps >= BACK) break;
    }
    if (found) s = q + G;
    else if (q >= 1) { if constexpr (BACK == LRH_BLN_BACK) a.st->need_slow = 1; else a.st->need_slow2 = 1; live = false; }  // no clean point in reach: the long-run replay / the second scan / the serial walk takes over
  }
  int cnt = 0;
  if (a.debug == 7) live = false;
  if (live) {
    int ifirst = 0, pk = 0, erase_end = 0, last = -2;
    float pulmax = 0;
    int q = s;
    for (;;) {
      const int n = next_above(q);
      if (ifirst != 0 && n != last + 1) {                // the run ended on the sample behind its last one (not above the limit, <= ce)
        const int qe = last + 1;
        ifirst = 0;
        int ib, ia;
        const int ext = bln_guards(a, pulmax, totnoise, &ib, &ia);
        pulmax = 0;
        if (ext) {
          if (qe >= cs) {
            for (int j = 1; j <= ib; j++) setbit(pk - j);
            for (int j = 0; j < ia; j++) setbit(qe + j);
            cnt += ib + ia;
          }
          erase_end = qe + ia;
        }
      }
      if (n > ce) break;
      if (n >= erase_end) {                              // above the limit and not erased ahead of its test (blank1.c:1030)
        const float v = sample(n);
        if (ifirst == 0) pk = n;
        if (v > pulmax) pulmax = v;
        ifirst++;
        if (n >= cs) { setbit(n); cnt++; }
        last = n;
      }
      q = n + 1;
    }
  }
  // one global atomic per workgroup for the count; sums reduced in a fixed order
  if (cnt) atomicAdd(&wg_cnt, cnt);
  for (int off = 32; off > 0; off >>= 1) s4 += __shfl_xor(s4, off);
  if (lane == 0) wg_sum[wv] = s4;
  __syncthreads();
  if (threadIdx.x == 0) {
    a.counts[blockIdx.x] = wg_cnt;
    reinterpret_cast<double *>(a.partials)[blockIdx.x] = (wg_sum[0] + wg_sum[1]) + (wg_sum[2] + wg_sum[3]);
  }
  // publish the decision words: interior words are owned by this tile alone (the mask is all zero between calls);
  // words within guard reach of either end can also be touched by the neighbours, so they are merged atomically
  const int edge = (max(a.clr1, a.clr2) >> 5) + 2;
  for (int i = threadIdx.x; i < LRH_BLN_WORDS; i += 256) {
    const unsigned int wv2 = wbits[i];
    if (!wv2) continue;
    const int wi = (w0 + i) & (nwords_ring - 1);
    if (i < edge || i >= LRH_BLN_TILE / 32 - edge) atomicOr(&a.mask_bits[wi], wv2); else a.mask_bits[wi] = wv2;
  }
}

// Long runs.  A lane of k_blank_scan gives up when LRH_BLN_BACK samples in a row sit above the limit -- a strong signal the
// selective limiter has not routed away yet, or a limit far below the noise.  Without pulse calibration (clr1 = 0,
// clr2 = 1: guards reach no further than the sample that ends a run, blank1.c:1013-1014, 1058-1086) the serial scan
// decomposes by runs of samples above the limit: every sample of a run is cleared, and the sample that ends it is cleared
// too when the run's maximum is 34 dB over the noise.  The only long-range quantity is that maximum, so the replay is a
// segmented max: chunk summaries in LDS, tile summaries in global memory (k_blank_runs_pre), then every lane decides its
// own 64 samples with the incoming (in run?, maximum so far) looked up backwards (k_blank_runs).  Both kernels return at
// once unless need_slow is set (k_blank_update, the last kernel of the call, takes the flag down); with calibration (clr2 > 1) guards chain runs together and k_blank_serial stays.
__device__ __forceinline__ bool bln_runs_mode(const BlankArgs &a) { return a.clr1 == 0 && a.clr2 == 1; }

struct BlnChunk { float cmax, smax; int any_below, last_above, empty; };
__device__ __forceinline__ BlnChunk bln_chunk_summary(const BlankArgs &a, int cs, int ce, float nfl)
{
  BlnChunk r; r.cmax = 0.f; r.smax = 0.f; r.any_below = 0; r.last_above = 0; r.empty = cs > ce;
  for (int q = cs; q <= ce; q++) {
    const float v = a.pwr[(a.pbeg + q) & a.mask];
    if (v > nfl) { r.cmax = fmaxf(r.cmax, v); r.smax = fmaxf(r.smax, v); r.last_above = 1; }
    else { r.any_below = 1; r.smax = 0.f; r.last_above = 0; }
  }
  return r;
}

__global__ __launch_bounds__(256) void k_blank_runs_pre(BlankArgs a)
{
  if (!a.st->need_slow || !bln_runs_mode(a)) return;
  __shared__ float l_max[256], l_smax[256];
  __shared__ int l_flags[256];
  const int qt = blockIdx.x * LRH_BLN_TILE;
  const int nwords_ring = (a.mask + 1) >> 5;
  const int w0 = ((a.pbeg + qt) & a.mask) >> 5;
  for (int i = threadIdx.x; i < LRH_BLN_WORDS; i += 256) a.mask_bits[(w0 + i) & (nwords_ring - 1)] = 0;   // what the scan left
  if (blockIdx.x == 0 && threadIdx.x == 0) a.st->call_cleared = 0;
  if (threadIdx.x == 0) a.counts[blockIdx.x] = 0;        // the scan's count goes with its decisions
  const int cb = (blockIdx.x * 256 + threadIdx.x) * LRH_BLN_CHUNK;
  const BlnChunk ch = bln_chunk_summary(a, max(cb, 1), min(cb + LRH_BLN_CHUNK - 1, a.total), (float)a.st->limit);
  l_max[threadIdx.x] = ch.cmax; l_smax[threadIdx.x] = ch.smax;
  l_flags[threadIdx.x] = ch.any_below | (ch.last_above << 1) | (ch.empty << 2);
  __syncthreads();
  if (threadIdx.x != 0) return;
  float tmax = 0.f, smax = 0.f; int any = 0, last_above = 0;
  for (int l = 0; l < 256; l++) {
    const int f = l_flags[l];
    if (f & 4) continue;
    tmax = fmaxf(tmax, l_max[l]);
    if (f & 1) { any = 1; smax = l_smax[l]; } else smax = fmaxf(smax, l_max[l]);
    last_above = (f >> 1) & 1;
  }
  a.tiles[blockIdx.x] = make_float4((float)any, tmax, smax, (float)last_above);
}

__global__ __launch_bounds__(256) void k_blank_runs(BlankArgs a)
{
  if (!a.st->need_slow || !bln_runs_mode(a)) return;
  __shared__ float l_max[256], l_smax[256];
  __shared__ int l_flags[256];
  __shared__ float t_max; __shared__ int t_run;
  const float nfl = (float)a.st->limit, totnoise = (float)(a.st->noise_floor * (a.chans > 1 ? a.chans : 1));
  const int cb = (blockIdx.x * 256 + threadIdx.x) * LRH_BLN_CHUNK;
  const int cs = max(cb, 1), ce = min(cb + LRH_BLN_CHUNK - 1, a.total);
  const BlnChunk ch = bln_chunk_summary(a, cs, ce, nfl);
  l_max[threadIdx.x] = ch.cmax; l_smax[threadIdx.x] = ch.smax;
  l_flags[threadIdx.x] = ch.any_below | (ch.last_above << 1) | (ch.empty << 2);
  if (threadIdx.x == 0) {                                  // state at the first sample of the tile
    int run = 0; float m = 0.f;
    for (int t = (int)blockIdx.x - 1; t >= 0; t--) {
      const float4 ti = a.tiles[t];
      if (ti.x != 0.f) { if (ti.w != 0.f) { run = 1; m = fmaxf(m, ti.z); } break; }
      run = 1; m = fmaxf(m, ti.y);                          // every sample of that tile above the limit
    }
    t_run = run; t_max = m;
  }
  __syncthreads();
  if (cs > ce) return;
  int run = 0; float pm = 0.f; bool open = true;            // state at cs: look back over the earlier chunks of the tile
  for (int l = (int)threadIdx.x - 1; l >= 0 && open; l--) {
    const int f = l_flags[l];
    if (f & 4) continue;
    if (f & 1) { if (f & 2) { run = 1; pm = fmaxf(pm, l_smax[l]); } open = false; }
    else { run = 1; pm = fmaxf(pm, l_max[l]); }
  }
  if (open && t_run) { run = 1; pm = fmaxf(pm, t_max); }
  int cnt = 0;
  int wcur = -1; unsigned int wacc = 0;                     // decision bits gathered per 32-sample ring word, one atomic per word
  auto flush = [&]() { if (wacc) atomicOr(&a.mask_bits[wcur], wacc); wacc = 0; };
  for (int q = cs; q <= ce; q++) {
    const int p = (a.pbeg + q) & a.mask;
    if ((p >> 5) != wcur) { flush(); wcur = p >> 5; }
    const float v = a.pwr[p];
    if (v > nfl) { run = 1; pm = fmaxf(pm, v); wacc |= 1u << (p & 31); cnt++; }
    else if (run) {
      run = 0;
      int ib, ia;
      const int ext = bln_guards(a, pm, totnoise, &ib, &ia);
      pm = 0.f;
      if (ext && ia > 0) { wacc |= 1u << (p & 31); cnt += ia; }   // ib = 0, ia <= 1: the ending sample itself
    }
  }
  flush();
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&a.st->call_cleared, cnt);
}

// exact serial replay for the calibrated blanker (clr2 > 1), only when a lane of k_blank_scan could not find a clean restart point
__global__ void k_blank_serial(BlankArgs a)
{
  if (!a.st->need_slow || !(a.st->need_slow2 || a.debug == 8) || bln_runs_mode(a)) return;     // (debug 8: no second scan, tests of this walk)
  a.st->need_slow = 0; a.st->need_slow2 = 0; a.st->slow_calls++;
  for (int i = 0; i < a.ncounts; i++) a.counts[i] = 0;
  for (int q = 1 - a.clr1 - 32; q <= a.total + a.clr2 + 32; q++) a.mask_bits[((a.pbeg + q) & a.mask) >> 5] = 0;
  const float nfl = (float)a.st->limit, totnoise = (float)(a.st->noise_floor * (a.chans > 1 ? a.chans : 1));
  int ifirst = 0, pk = 0, erase_end = 0, cnt = 0; float pulmax = 0;
  for (int q = 1; q <= a.total; q++) {
    const int p = (a.pbeg + q) & a.mask;
    const float v = a.pwr[p];
    if (v > nfl && q >= erase_end) {
      if (ifirst == 0) pk = q;
      if (v > pulmax) pulmax = v;
      ifirst++; a.mask_bits[p >> 5] |= 1u << (p & 31); cnt++;
    } else if (ifirst != 0) {
      ifirst = 0;
      int ib, ia;
      const int ext = bln_guards(a, pulmax, totnoise, &ib, &ia);
      pulmax = 0;
      if (ext) {
        for (int j = 1; j <= ib; j++) { const int r = (a.pbeg + pk - j) & a.mask; a.mask_bits[r >> 5] |= 1u << (r & 31); }
        for (int j = 0; j < ia; j++) { const int r = (a.pbeg + q + j) & a.mask; a.mask_bits[r >> 5] |= 1u << (r & 31); }
        cnt += ib + ia; erase_end = q + ia;
      }
    }
  }
  a.st->call_cleared = cnt;
}

// The same walk by one wave (the form that is launched; k_blank_serial stays as the statement of it and for comparison, LRH_BLN_SERIAL=1).
// The state of the walk (in a run? its start, its maximum; where the last guard ends) is uniform over the lanes; the lanes hold 64
// consecutive powers each step, eight such steps in flight, and the walk moves by runs through the ballot of "above the limit": a run's
// length is a count of trailing ones, its maximum a wave reduction, its mask bits three words formed by shifts and ORed into the (cleared)
// mask without waiting for them.  One lane and one dependent global load per sample took 200 ns per sample -- 6.7 s for a 33.5 M-sample
// call whose limit sits below the noise (the start-up of a calibrated receiver); this form takes a few ns per sample.
__global__ __launch_bounds__(64) void k_blank_serial_wave(BlankArgs a)
{
  if (!a.st->need_slow || !(a.st->need_slow2 || a.debug == 8) || bln_runs_mode(a)) return;
  const int lane = threadIdx.x;
  const int wordmask = ((a.mask + 1) >> 5) - 1;
  for (int i = lane; i < a.ncounts; i += 64) a.counts[i] = 0;
  {
    const int qlo = 1 - a.clr1 - 32, nq = a.total + a.clr2 + 32 - qlo + 1;
    const int w0 = ((a.pbeg + qlo) & a.mask) >> 5;
    for (int k = lane; k <= nq / 32 + 1; k += 64) a.mask_bits[(w0 + k) & wordmask] = 0;
  }
  __threadfence();                                       // the clears are in memory before any bit is ORed in
  const float nfl = (float)a.st->limit, totnoise = (float)(a.st->noise_floor * (a.chans > 1 ? a.chans : 1));
  int ifirst = 0, pk = 0, erase_end = 0, cnt = 0; float pulmax = 0;
  // bit i of `bits` = sequence position q_base + i: three words ORed into the mask, nobody waits for them
  auto or_bits = [&](int q_base, unsigned long long bits) {
    const int p0 = (a.pbeg + q_base) & a.mask, off = p0 & 31, w = p0 >> 5;
    const unsigned long long lo = bits << off, hi = off ? bits >> (64 - off) : 0ull;
    const unsigned int part = lane == 0 ? (unsigned int)lo : (lane == 1 ? (unsigned int)(lo >> 32) : (unsigned int)hi);
    if (lane < 3 && part) atomicOr(&a.mask_bits[(w + lane) & wordmask], part);
  };
  // The bits of the step before, the current step and the next one gather in registers (guards reach a few samples behind a run's start
  // and beyond its end) and leave once per step: atomics on one mask word from run after run queue up in the L2 (a dozen per 32 samples
  // when the limit sits in the noise: 120 ns per sample).  What falls outside the window -- the guard before a run that began steps ago --
  // goes out directly.
  unsigned long long acc[3] = { 0, 0, 0 };                // positions qc - 64 .., qc .., qc + 64 ..
  int qc = 1;
  auto mark = [&](int q_first, int n) {                  // sequence positions q_first .. q_first + n - 1
    int r = q_first - (qc - 64);
    if (r < 0 || r + n > 192) {
      for (int j = 0; j < n; j += 64) { const int m = n - j < 64 ? n - j : 64; or_bits(q_first + j, m == 64 ? ~0ull : (1ull << m) - 1); }
      return;
    }
#pragma unroll
    for (int w = 0; w < 3; w++) {
      const int lo = r > 64 * w ? r : 64 * w, hi = r + n < 64 * w + 64 ? r + n : 64 * w + 64;
      if (lo < hi) acc[w] |= ((hi - lo == 64) ? ~0ull : (1ull << (hi - lo)) - 1) << (lo - 64 * w);
    }
  };
  constexpr int DEPTH = 8;
  for (int q0 = 1; q0 <= a.total; q0 += 64 * DEPTH) {
    float v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) { const int q = q0 + 64 * d + lane; v[d] = q <= a.total ? a.pwr[(a.pbeg + q) & a.mask] : 0.f; }
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      if (q0 + 64 * d > a.total) break;
      if (q0 + 64 * d != qc) {                            // the window moves on by one step
        if (acc[0]) or_bits(qc - 64, acc[0]);
        acc[0] = acc[1]; acc[1] = acc[2]; acc[2] = 0; qc += 64;
      }
      const int valid = a.total - qc + 1 < 64 ? a.total - qc + 1 : 64;
      const unsigned long long hot = __ballot(v[d] > nfl);
      if (hot == 0 && ifirst == 0) continue;
      int pos = 0;
      while (pos < valid) {
        if (ifirst == 0) {                                // the next sample that starts a run: above the limit, at or behind the last guard's end
          unsigned long long m = hot & (~0ull << pos);
          const int skip = erase_end - qc;
          if (skip >= 64) m = 0; else if (skip > 0) m &= ~0ull << skip;
          if (m == 0) break;
          pos = __ffsll((long long)m) - 1;
          pk = qc + pos;
        }
        const unsigned long long rest = ~(hot >> pos);    // first zero above pos = the run's length within this step
        int len = rest ? __ffsll((long long)rest) - 1 : 64 - pos;
        if (len > valid - pos) len = valid - pos;
        if (len > 0) {
          const unsigned long long seg = (len == 64 ? ~0ull : (1ull << len) - 1) << pos;
          if (len <= 12) {                                  // a short run: its samples one by one out of the lanes (uniform index)
            for (int i = pos; i < pos + len; i++) { const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[d]), i)); if (x > pulmax) pulmax = x; }
          } else {
            float mx = ((seg >> lane) & 1) ? v[d] : 0.f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            mx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mx)));   // the same in every lane: keep the walk's state scalar
            if (mx > pulmax) pulmax = mx;
          }
          ifirst += len; cnt += len; acc[1] |= seg;
          pos += len;
        }
        if (pos >= valid) break;                          // the run goes on in the next step (or the call ends inside it: no guards then)
        // the sample at pos ends the run
        ifirst = 0;
        int ib, ia;
        const int ext = bln_guards(a, pulmax, totnoise, &ib, &ia);
        pulmax = 0;
        if (ext) {
          if (ib > 0) mark(pk - ib, ib);
          if (ia > 0) mark(qc + pos, ia);
          cnt += ib + ia; erase_end = qc + pos + ia;
        }
        pos++;
      }
    }
  }
#pragma unroll
  for (int w = 0; w < 3; w++) if (acc[w]) or_bits(qc - 64 + 64 * w, acc[w]);
  if (lane == 0) { a.st->need_slow = 0; a.st->need_slow2 = 0; a.st->slow_calls++; a.st->call_cleared = cnt; }
}

// ---- the same walk, tile-parallel ---------------------------------------------------------------------------------------------
// The walk's state between two samples is small: either "inside a run" (its start, its maximum so far) or "outside", and then only the
// end of the last guard matters (samples before it are skipped).  Outside a run at position z with no guard reaching z, the future does
// not depend on the past at all.  So:
//   1. k_blank_walk_spec, one wave per tile of LRH_BLN_WTILE positions: walk the tile as if it were entered outside a run with no guard
//      pending; keep where that walk was busy (inside a run, on the sample that ends it, inside the guard behind it) and the state it
//      leaves the tile in.  No decisions are written.
//   2. k_blank_walk_chain, one wave, tiles in order: the state the call really enters each tile in.  Entered clean, the tile leaves as
//      the speculative walk left it.  Otherwise (a run or a guard crosses the boundary) the true walk is followed into the tile until it
//      stands outside a run, past its guard, on a position where the speculative walk was not busy: from there on the two are the same
//      walk.  With the limit in the noise -- a run every few samples, the case that made the one-wave walk take 2.6 s for 33.5 M
//      samples -- that happens within the first run or two; a tile that is one long run is crossed in 64 steps of 64.
//   3. k_blank_walk_final, one wave per tile: the walk again from the true entry state, this time writing the decision bits and
//      counting (run by run, as the serial walk counts: guards that overlap earlier decisions count again, blank1.c:1049-1083).
// Same decisions, same count as k_blank_serial_wave (tests/test_gpu_fullsize.py: both against the oracle, and against each other).
struct BlnWalk { int in_run, pk, erase_end; float pulmax; };
// MODE 0: speculative (busy words out), 1: chase (busy words in; true = merged with the speculative walk), 2: final (mask bits, count)
template <int MODE>
__device__ __forceinline__ bool bln_walk(const BlankArgs &a, int q_first, int q_last, BlnWalk &s, unsigned long long *busy, int &cnt_out)
{
  const int lane = threadIdx.x & 63;
  const int wordmask = ((a.mask + 1) >> 5) - 1;
  const float nfl = (float)a.st->limit, totnoise = (float)(a.st->noise_floor * (a.chans > 1 ? a.chans : 1));
  int ifirst = s.in_run, pk = s.pk, erase_end = s.erase_end, cnt = 0; float pulmax = s.pulmax;
  auto or_bits = [&](int q_base, unsigned long long bits) {
    const int p0 = (a.pbeg + q_base) & a.mask, off = p0 & 31, w = p0 >> 5;
    const unsigned long long lo = bits << off, hi = off ? bits >> (64 - off) : 0ull;
    const unsigned int part = lane == 0 ? (unsigned int)lo : (lane == 1 ? (unsigned int)(lo >> 32) : (unsigned int)hi);
    if (lane < 3 && part) atomicOr(&a.mask_bits[(w + lane) & wordmask], part);
  };
  unsigned long long acc[3] = { 0, 0, 0 };
  int qc = q_first;
  auto mark = [&](int qf, int n) {
    if constexpr (MODE != 2) return;
    int r = qf - (qc - 64);
    if (r < 0 || r + n > 192) {
      for (int j = 0; j < n; j += 64) { const int m = n - j < 64 ? n - j : 64; or_bits(qf + j, m == 64 ? ~0ull : (1ull << m) - 1); }
      return;
    }
#pragma unroll
    for (int w = 0; w < 3; w++) {
      const int lo = r > 64 * w ? r : 64 * w, hi = r + n < 64 * w + 64 ? r + n : 64 * w + 64;
      if (lo < hi) acc[w] |= ((hi - lo == 64) ? ~0ull : (1ull << (hi - lo)) - 1) << (lo - 64 * w);
    }
  };
  auto low = [](int n) -> unsigned long long { return n >= 64 ? ~0ull : (n <= 0 ? 0ull : (1ull << n) - 1); };
  constexpr int DEPTH = 8;
  bool merged = false;
  for (int q0 = q_first; q0 <= q_last && !merged; q0 += 64 * DEPTH) {
    float v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) { const int q = q0 + 64 * d + lane; v[d] = q <= q_last ? a.pwr[(a.pbeg + q) & a.mask] : 0.f; }
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      if (q0 + 64 * d > q_last || merged) break;
      if (q0 + 64 * d != qc) {
        if constexpr (MODE == 2) { if (acc[0]) or_bits(qc - 64, acc[0]); acc[0] = acc[1]; acc[1] = acc[2]; acc[2] = 0; }
        qc += 64;
      }
      const int step = (qc - q_first) >> 6;
      const int valid = q_last - qc + 1 < 64 ? q_last - qc + 1 : 64;
      const unsigned long long hot = __ballot(v[d] > nfl);
      unsigned long long bw = 0;                           // MODE 0: this step's busy bits; MODE 1: the speculative walk's
      if constexpr (MODE == 0) bw = low(erase_end - qc);   // the guard of a run that ended in an earlier step
      if constexpr (MODE == 1) bw = busy[step];
      int pos = 0;
      if (!(hot == 0 && ifirst == 0)) while (pos < valid) {
        if (ifirst == 0) {
          const int skip = erase_end - qc;
          const int c = skip > pos ? skip : pos;          // outside a run and past the guard from position c of this step on
          if (c >= 64) break;
          const unsigned long long m = hot & (~0ull << c);
          if constexpr (MODE == 1) {
            const unsigned long long fr = ~bw & (~0ull << c) & low(valid);
            const int h = m ? __ffsll((long long)m) - 1 : 64, z = fr ? __ffsll((long long)fr) - 1 : 64;
            if (z <= h && z < valid) { merged = true; break; }
          }
          if (m == 0) break;
          pos = __ffsll((long long)m) - 1;
          if (pos >= valid) break;
          pk = qc + pos;
        }
        const unsigned long long rest = ~(hot >> pos);
        int len = rest ? __ffsll((long long)rest) - 1 : 64 - pos;
        if (len > valid - pos) len = valid - pos;
        if (len > 0) {
          const unsigned long long seg = (len == 64 ? ~0ull : (1ull << len) - 1) << pos;
          if (len <= 12) {
            for (int i = pos; i < pos + len; i++) { const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[d]), i)); if (x > pulmax) pulmax = x; }
          } else {
            float mx = ((seg >> lane) & 1) ? v[d] : 0.f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            mx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mx)));
            if (mx > pulmax) pulmax = mx;
          }
          ifirst = 1; cnt += len;
          if constexpr (MODE == 2) acc[1] |= seg;
          if constexpr (MODE == 0) bw |= seg;
          pos += len;
        }
        if (pos >= valid) break;
        ifirst = 0;
        int ib, ia;
        const int ext = bln_guards(a, pulmax, totnoise, &ib, &ia);
        pulmax = 0;
        if constexpr (MODE == 0) bw |= 1ull << pos;       // the sample that ends the run
        if (ext) {
          if (ib > 0) mark(pk - ib, ib);
          if (ia > 0) mark(qc + pos, ia);
          cnt += ib + ia; erase_end = qc + pos + ia;
          if constexpr (MODE == 0) bw |= low(pos + ia) & ~low(pos);
        }
        pos++;
      }
      else if constexpr (MODE == 1) {                      // nothing above the limit here: still outside a run; has the guard ended where the other walk is free?
        const int skip = erase_end - qc;
        if (skip < 64) { const unsigned long long fr = ~bw & (~0ull << (skip > 0 ? skip : 0)) & low(valid); if (fr) merged = true; }
      }
      if constexpr (MODE == 0) { if (lane == 0) busy[step] = bw; }
    }
  }
  if constexpr (MODE == 2) {
#pragma unroll
    for (int w = 0; w < 3; w++) if (acc[w]) or_bits(qc - 64 + 64 * w, acc[w]);
  }
  s.in_run = ifirst; s.pk = pk; s.erase_end = erase_end; s.pulmax = pulmax;
  cnt_out = cnt;
  return merged;
}
__device__ __forceinline__ bool bln_walk_wanted(const BlankArgs &a) { return a.st->need_slow && (a.st->need_slow2 || a.debug == 8) && !bln_runs_mode(a); }
__device__ __forceinline__ int4 bln_pack(const BlnWalk &s) { return make_int4(s.in_run, s.pk, s.erase_end, __float_as_int(s.pulmax)); }
__device__ __forceinline__ BlnWalk bln_unpack(int4 v) { BlnWalk s; s.in_run = v.x; s.pk = v.y; s.erase_end = v.z; s.pulmax = __int_as_float(v.w); return s; }

__global__ __launch_bounds__(64) void k_blank_walk_spec(BlankArgs a)
{
  if (!bln_walk_wanted(a)) return;
  const int lane = threadIdx.x, t = blockIdx.x;
  const int q_first = 1 + t * LRH_BLN_WTILE, q_last = min(q_first + LRH_BLN_WTILE - 1, a.total);
  const int wordmask = ((a.mask + 1) >> 5) - 1;
  if (t == 0) for (int i = lane; i < a.ncounts; i += 64) a.counts[i] = 0;      // the scan's counts go with its decisions
  {                                                     // ... and so do its bits: this tile's span of the mask (and the margins the guards reach, at the ends)
    const int qlo = t == 0 ? 1 - a.clr1 - 32 : q_first, qhi = t == (int)gridDim.x - 1 ? a.total + a.clr2 + 32 : q_last;
    const int w0 = ((a.pbeg + qlo) & a.mask) >> 5, nw = (qhi - qlo + 1) / 32 + 2;
    for (int k = lane; k < nw; k += 64) a.mask_bits[(w0 + k) & wordmask] = 0;
  }
  BlnWalk s = { 0, 0, 0, 0.f };
  int cnt;
  bln_walk<0>(a, q_first, q_last, s, a.wbusy + (size_t)t * (LRH_BLN_WTILE / 64), cnt);
  if (lane == 0) a.wstate[t] = bln_pack(s);
}
__global__ __launch_bounds__(64) void k_blank_walk_chain(BlankArgs a)
{
  if (!bln_walk_wanted(a)) return;
  BlnWalk s = { 0, 0, 0, 0.f };
  for (int t = 0; t < a.nwt; t++) {
    const int q_first = 1 + t * LRH_BLN_WTILE, q_last = min(q_first + LRH_BLN_WTILE - 1, a.total);
    if (threadIdx.x == 0) a.wstate[a.nwt + t] = bln_pack(s);
    if (!s.in_run && s.erase_end <= q_first) { s = bln_unpack(a.wstate[t]); continue; }
    int cnt;
    if (bln_walk<1>(a, q_first, q_last, s, a.wbusy + (size_t)t * (LRH_BLN_WTILE / 64), cnt)) s = bln_unpack(a.wstate[t]);
  }
  if (threadIdx.x == 0) a.st->call_cleared = 0;
}
__global__ __launch_bounds__(64) void k_blank_walk_final(BlankArgs a)
{
  if (!bln_walk_wanted(a)) return;
  const int t = blockIdx.x;
  const int q_first = 1 + t * LRH_BLN_WTILE, q_last = min(q_first + LRH_BLN_WTILE - 1, a.total);
  BlnWalk s = bln_unpack(a.wstate[a.nwt + t]);
  int cnt = 0;
  bln_walk<2>(a, q_first, q_last, s, nullptr, cnt);
  if (threadIdx.x == 0 && cnt) atomicAdd(&a.st->call_cleared, cnt);
}

// one thread per 32-sample mask word: zero the flagged samples (weak I/Q + power), reset the word, and report how much
// every-4th-sample power was removed (the noise statistic of blank1.c:1493-1497 is taken after clearing)
__global__ __launch_bounds__(256) void k_blank_apply(BlankArgs a, int first_word, int nwords, int word_mask)
{
  __shared__ double red[4];
  const int w = blockIdx.x * 256 + threadIdx.x;
  double removed = 0;
  if (w < nwords) {
    const int wi = (first_word + w) & word_mask;
    unsigned int bits = a.mask_bits[wi];
    if (bits) {
      a.mask_bits[wi] = 0;
      while (bits) {
        const int bpos = __ffs(bits) - 1; bits &= bits - 1;
        const int p = wi * 32 + bpos;
        const int q = (p - a.pbeg) & a.mask;              // sequence position 1..total (guards may fall outside)
        if ((q & 3) == 0 && q >= 4 && q <= a.total) removed += (double)a.pwr[p];
        a.pwr[p] = 0;
        if (a.own) a.own[p] = 0;
        a.timf2w[p] = make_float2(0.f, 0.f);              // weak part only (blank1.c:1043-1045)
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) removed += __shfl_xor(removed, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = removed;
  __syncthreads();
  if (threadIdx.x == 0) reinterpret_cast<double *>(a.partials)[a.npartials + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// every-4th-sample power sum (blank1.c:1493-1497) as fixed-order partial sums
__global__ __launch_bounds__(256) void k_blank_stats(BlankArgs a)
{
  __shared__ double red[256];
  const int per = (a.nstat + a.npartials - 1) / a.npartials;
  const int j0 = blockIdx.x * per, j1 = min(j0 + per, a.nstat);
  double acc = 0;
  const float *ring = a.own ? a.own : a.pwr;            // two coupled channels: this channel's own power (blank1.c:1512-1523)
  for (int j = j0 + threadIdx.x; j < j1; j += 256) acc += (double)ring[(a.pbeg + 4 * (j + 1)) & a.mask];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) reinterpret_cast<double *>(a.partials)[blockIdx.x] = red[0];
}

// scalar bookkeeping of blank1.c:1472-1601 (1 channel); host supplies everything that does not depend on data
__global__ void k_blank_update(BlankArgs a)
{
  // partial sums in a fixed order: lane l adds partials l, l+64, ..., then a butterfly over the wave
  __shared__ double wsum[4];
  double tot = 0;
  for (int i = threadIdx.x; i < a.npartials; i += 256) tot += reinterpret_cast<double *>(a.partials)[i];
  for (int i = threadIdx.x; i < a.nremoved; i += 256) tot -= reinterpret_cast<double *>(a.partials)[a.npartials + i];
  int ncl = 0;                                           // samples the scan cleared, per tile
  for (int i = threadIdx.x; i < a.ncounts; i += 256) ncl += a.counts[i];
  __shared__ int wcnt[4];
  for (int off = 32; off > 0; off >>= 1) { tot += __shfl_xor(tot, off); ncl += __shfl_xor(ncl, off); }
  if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = tot; wcnt[threadIdx.x >> 6] = ncl; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  tot = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
  ncl = (wcnt[0] + wcnt[1]) + (wcnt[2] + wcnt[3]);
  BlankState *s = a.st;
  if (s->need_slow) { s->need_slow = 0; s->need_slow2 = 0; s->slow_calls++; }   // the long-run replay / the tile-parallel walk served this call (k_blank_serial resets the flags itself)
  float t1;
  if (a.phase == 2) {                                    // second half of a coupled call: both channels' means have arrived
    s->despiked_pwrinc[0] += a.xstat[0];                  // blank1.c:1538-1541
    s->despiked_pwrinc[1] += a.xstat[1];
    t1 = a.xstat[0] + a.xstat[1];
  } else {
  if (a.debug == 3) {
    double sa = 0, sr = 0;
    for (int i = 0; i < a.npartials; i++) sa += reinterpret_cast<double *>(a.partials)[i];
    for (int i = 0; i < a.nremoved; i++) sr += reinterpret_cast<double *>(a.partials)[a.npartials + i];
    printf("blank_update: npartials %d nremoved %d all %.1f removed %.1f tot %.1f m %d cleared %d total %d\n", a.npartials, a.nremoved, sa, sr, tot, a.m, s->call_cleared, a.total);
  }
  const int cleared = s->call_cleared + ncl;            // long-run / serial replay + the scan's tiles
  s->call_cleared = 0;
  s->last_cleared = cleared;
  s->cleared_acc += cleared;
  s->last_fitted = a.fitted; s->last_rejected = a.rejected; s->fitted_acc += a.fitted;
  int k = a.m - cleared - a.fitted; if (k < a.m / 25) k = a.m / 25; k = (k + 2) / 4; if (k < 1) k = 1;
  t1 = (float)tot;
  t1 /= k; if (t1 < 10) t1 = 10;
  if (a.phase == 1) { a.xstat[0] = a.xstat[1] = 0.f; a.xstat[a.own_slot & 1] = t1; return; }   // the partner's half comes by exchange
  s->despiked_pwrinc[0] += t1;
  }
  if (!a.do_update) return;
  s->despiked_pwr[0] = s->despiked_pwrinc[0] / (a.interval * a.lowlevel_fraction);
  s->despiked_pwr[1] = s->despiked_pwrinc[1] / (a.interval * a.lowlevel_fraction);
  float crate = (float)(100. * (double)(float)s->fitted_acc / (double)a.blanker_points);
  if (crate > 99) crate = 99;
  s->clever_rate = crate;
  float rate = (float)(100. * (double)(float)s->cleared_acc / (double)a.blanker_points);
  if (rate > 99) rate = 99;
  s->stupid_rate = rate;
  int nf = (int)((s->despiked_pwr[0] + s->despiked_pwr[1]) / (a.chans > 1 ? a.chans : 1));
  if (a.mode == 1) {
    if (rate > 20) {
      if (nf < 30) nf = 30;
      t1 = (float)(0.01 * pow((double)rate - 20.0, 2.));
      if (t1 > 10) t1 = 10;
      nf = (int)((float)nf * (1 + t1));
    } else {
      nf = (int)(((float)((a.avgnum - 1) * nf) + t1) / (float)a.avgnum);
    }
    s->limit = (unsigned int)((float)nf * a.factor);
  }
  s->noise_floor = nf;
  if (a.clever_mode == 1) s->clever_limit = (unsigned int)((float)nf * a.clever_factor);      // blank1.c:1587-1590
  s->despiked_pwrinc[0] = 1; s->despiked_pwrinc[1] = 1;
  s->cleared_acc = 0; s->fitted_acc = 0;
}

// =====================================================================================================
// fft2 (one workgroup per transform, N2 <= 16384)
// =====================================================================================================
// LDS (exchange + twiddle tables) admits three 256-thread workgroups per CU at N = 4096: three waves per SIMD, 168 VGPRs
__host__ __device__ constexpr int fft2_min_waves(int log2n) { return log2n == 12 ? 3 : (log2n == 13 ? 2 : 1); }
template <int LOG2N, bool FUSED>
__global__ __launch_bounds__(fft_threads(LOG2N), fft2_min_waves(LOG2N)) void k_fft2(Fft2Args a)
{
  constexpr int P = points_per_thread(LOG2N);
  using Plan = FftPlan<LOG2N, P>;
  using Fft = BlockFftL<LOG2N, P, +1>;
  constexpr int N = Plan::N, T = Plan::T, R0 = Plan::R0, RL = Plan::RL;
  static_assert(P == R0, "one first-pass butterfly per thread: element s is sample tid + s*N/R0");
  constexpr int H = R0 / 2;                              // elements of the first half of the transform
  __shared__ float2 lds[Fft::LDS_CELLS];
  const int tid0 = threadIdx.x;
  Fft::init(lds, a.tw, tid0);
  // A workgroup takes a.run consecutive transforms.  With the 50 % overlap of the windowed fft2 (step = N/2) the
  // second half of transform t is the first half of t+1 and belongs to the same thread (sample tid + s N/R0 with
  // s >= R0/2), so its weak+strong sum stays in registers and every timf2 sample is fetched once per run instead of
  // twice per transform pair.
  const int g = a.xcd ? xcd_order(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  int t_first = g * a.run, t_end = min(t_first + a.run, a.batch);
  constexpr bool fused = FUSED;
  bool ps_continue = false, ps_complete = false;
  if (fused) {                                           // group arithmetic of k_powersum2
    t_first = g == 0 ? 0 : g * a.ps_avgnum - a.ps_counter;
    int count = a.ps_avgnum - (g == 0 ? a.ps_counter : 0);
    ps_complete = count <= a.batch - t_first;
    if (!ps_complete) count = a.batch - t_first;
    t_end = t_first + count;
    ps_continue = g == 0 && a.ps_counter > 0;
  }
  const bool overlap = a.step * 2 == N;
  // Load order as in k_fft1: what the next transform needs first (its fresh half, the window values) is requested
  // BEFORE the stores of the current one, so the stores drain underneath the next transform instead of in front of
  // its first wait (vmcnt retires in issue order).
  float2 keep[H];                                        // raw weak+strong sum of the half shared with the next transform
  float2 fw[H], fs[H];                                   // weak / strong samples of the fresh half, in flight
  float win[P];
  auto load_half = [&](int px, int h, float2 (&dw)[H], float2 (&ds)[H], int tid) {
#pragma unroll
    for (int s = 0; s < H; s++) {
      const int r = (px + tid + (h * H + s) * (N / R0)) & a.mask;
      dw[s] = a.timf2w[r]; ds[s] = a.timf2s[r];
    }
  };
  auto load_window = [&](int tid) {
#pragma unroll
    for (int s = 0; s < P; s++) win[s] = a.window[tid + s * (N / R0)];
  };
  if (t_first < t_end) {
    load_half(a.px_first + t_first * a.step, 0, fw, fs, tid0);
#pragma unroll
    for (int s = 0; s < H; s++) keep[s] = make_float2(fw[s].x + fs[s].x, fw[s].y + fs[s].y);   // weak + strong (fft2.c:100-105)
    load_half(a.px_first + t_first * a.step, 1, fw, fs, tid0);
    load_window(tid0);
  }
  float acc[FUSED ? P : 1];                              // running sum |X|^2 of the averaging group (fused mode)
  if constexpr (FUSED) {
#pragma unroll
    for (int m = 0; m < P / RL; m++)
#pragma unroll
      for (int q = 0; q < RL; q++) acc[m * RL + q] = ps_continue ? a.ps_in[(tid0 + m * T) + q * (N / RL)] : 0.f;
  }
  __syncthreads();                                       // twiddle tables are in place
  __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): retire the prologue's loads (see k_fft1)
#pragma unroll 1
  for (int b = t_first; b < t_end; b++) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));                        // keep index math inside the loop
    const int px = a.px_first + b * a.step;
    float2 x[P];
#pragma unroll
    for (int s = 0; s < H; s++) {
      const float2 fr = make_float2(fw[s].x + fs[s].x, fw[s].y + fs[s].y);
      x[s] = make_float2(win[s] * keep[s].x, win[s] * keep[s].y);
      x[H + s] = make_float2(win[H + s] * fr.x, win[H + s] * fr.y);
      keep[s] = fr;
    }
    Fft::run(x, lds, tid);
    // pin the outputs, then prefetch for the next transform (the last trip re-reads its own samples: harmless)
#pragma unroll
    for (int e = 0; e < P; e++) asm volatile("" : "+v"(x[e].x), "+v"(x[e].y));
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const int pn = b + 1 < t_end ? px + a.step : px;
    if (!overlap) {
      load_half(pn, 0, fw, fs, tid);
      __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
      for (int s = 0; s < H; s++) keep[s] = make_float2(fw[s].x + fs[s].x, fw[s].y + fs[s].y);
    }
    load_half(pn, 1, fw, fs, tid);
    load_window(tid);
    __builtin_amdgcn_sched_barrier(0);
    const int na = (a.first_na + b) & a.na_mask;
    float2 *out = a.out + (size_t)na * N;
    float *pw = a.power + (size_t)na * N;
    if (a.keep_lo <= 0 && a.keep_hi >= N) {             // every bin reaches the ring (uniform: no per-lane predicate on this path)
#pragma unroll
      for (int m = 0; m < P / RL; m++)
#pragma unroll
        for (int q = 0; q < RL; q++) store_stream(&out[(tid + m * T) + q * (N / RL)], x[m * RL + q]);
    } else {                                             // cfg.fft2_float_sparse: the band mix1 will cut out
#pragma unroll
      for (int m = 0; m < P / RL; m++)
#pragma unroll
        for (int q = 0; q < RL; q++) { const int k = (tid + m * T) + q * (N / RL); if (k >= a.keep_lo && k < a.keep_hi) store_stream(&out[k], x[m * RL + q]); }
    }
#pragma unroll
    for (int m = 0; m < P / RL; m++)
#pragma unroll
      for (int q = 0; q < RL; q++) {
        const int k = (tid + m * T) + q * (N / RL);
        const float2 v = x[m * RL + q];
        const float p2 = v.x * v.x + v.y * v.y;
        if constexpr (FUSED) acc[m * RL + q] = (b == t_first && !ps_continue) ? p2 : acc[m * RL + q] + p2;   // "=" then "+=" (fft2.c:655-670)
        else pw[k] = p2;
      }
  }
  if constexpr (FUSED) {
#pragma unroll
    for (int m = 0; m < P / RL; m++)
#pragma unroll
      for (int q = 0; q < RL; q++) {
        const int k = (tid0 + m * T) + q * (N / RL);
        if (ps_complete) a.wf_scratch[(size_t)g * N + k] = acc[m * RL + q];
        if (g == (int)gridDim.x - 1) a.ps_out[k] = acc[m * RL + q];
      }
  }
}

// fft2_power_float on demand (export of LRH_RING_FFT2_POWER when the fused path does not keep the ring)
__global__ __launch_bounds__(256) void k_power_of(const float2 *src, float *dst, size_t n)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const float2 v = src[i]; dst[i] = v.x * v.x + v.y * v.y; }
}
hipError_t launch_power_of(const float2 *src, float *dst, size_t n, hipStream_t st)
{
  hipLaunchKernelGGL(k_power_of, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n);
  return hipGetLastError();
}

// ---- fft2 for N2 > 16384: four-step through an HBM scratch -----------------------------------------------------
// n = NB*n1 + n2, k = k1 + NA*k2:  X[k] = sum_n2 [ w_N^(n2 k1) * sum_n1 z[NB n1 + n2] w_NA^(n1 k1) ] w_NB^(n2 k2).
// Both steps transform along the strided index of a 16-wide tile, threads laid out tile-column-fastest, so every
// global access is a full 128-byte line (16 x float2).  Step A transposes its result through LDS so that step B
// finds scratch[n2][k1] with k1 contiguous.
#define LRH_TILE 16
__host__ __device__ constexpr int sub_ppt(int log2l) { return log2l >= 9 ? 8 : 4; }   // 64 threads per sub-transform

// A workgroup takes a.run consecutive transforms of its column tile: the step twiddles are the same for all of
// them (registers), the window values come from the cache, and with the 50 % overlap of the windowed fft2 (step = N/2) the second half of transform t
// (n1 >= NA/2, which is element s >= R0/2 of the same thread) is the first half of t+1, so its weak+strong sum stays
// in registers and a transform after the first costs half the loads.
template <int LA, int LB>
__global__ __launch_bounds__(1024, LA <= 8 ? 8 : 4) void k_fft2_cols(Fft2BigArgs a)
{
  constexpr int P = sub_ppt(LA);
  using Plan = FftPlan<LA, P>;
  constexpr int NA = 1 << LA, NB = 1 << LB, T = Plan::T, R0 = Plan::R0, RL = Plan::RL, H = R0 / 2;
  constexpr int CS = Plan::LDS_CELLS + 1;               // odd column stride: tile columns land on different banks
  __shared__ float2 lds[LRH_TILE * CS];
  const int c0 = threadIdx.x & (LRH_TILE - 1), l0 = threadIdx.x >> 4;
  const int n20 = blockIdx.x * LRH_TILE + c0;
  const int t_first = blockIdx.y * a.run, t_end = min(t_first + a.run, a.batch);
  const bool overlap = a.step * 2 == NA * NB;
  float2 twk[P], keep[P / 2], pfw[P / 2], pfs[P / 2];
#pragma unroll
  for (int m = 0; m < P / RL; m++)
#pragma unroll
    for (int q = 0; q < RL; q++) {                       // w_N^(n2 k1), e^{+j}: conjugate of the forward table
      const float2 w = a.tw_big[(n20 * ((l0 + m * T) + q * (NA / RL))) & (NA * NB - 1)];
      twk[m * RL + q] = make_float2(w.x, -w.y);
    }
#pragma unroll 1
  for (int b = t_first; b < t_end; b++) {
    // opaque per-iteration copies of the lane coordinates: keeps the address arithmetic inside the loop (see k_timf2)
    int l = l0, n2 = n20, c = c0;
    asm volatile("" : "+v"(l), "+v"(n2), "+v"(c));
    float2 *col = lds + c * CS;
    const int px = a.px_first + b * a.step;
    const bool reuse = overlap && b > t_first;
    float2 x[P];
#pragma unroll
    for (int m = 0; m < P / R0; m++)
#pragma unroll
      for (int s = 0; s < R0; s++) {
        float2 raw;
        if (s < H && reuse) raw = keep[m * H + s];
        else if (s >= H && reuse) raw = make_float2(pfw[m * H + s - H].x + pfs[m * H + s - H].x, pfw[m * H + s - H].y + pfs[m * H + s - H].y);
        else {
          const int r = (px + NB * ((l + m * T) + s * (NA / R0)) + n2) & a.mask;
          const float2 vw = load_stream(&a.timf2w[r]), vs = load_stream(&a.timf2s[r]);
          raw = make_float2(vw.x + vs.x, vw.y + vs.y);     // weak + strong (fft2.c:100-105)
        }
        const float w = a.window[NB * ((l + m * T) + s * (NA / R0)) + n2];      // cache hits after the first transform
        x[m * R0 + s] = make_float2(w * raw.x, w * raw.y);
        if (s >= H) keep[m * H + s - H] = raw;
      }
    BlockFft<LA, P, +1>::run(x, col, a.tw_a, l);
    __syncthreads();
    // step twiddle, then park as col[k1]
#pragma unroll
    for (int m = 0; m < P / RL; m++)
#pragma unroll
      for (int q = 0; q < RL; q++) col[(l + m * T) + q * (NA / RL)] = cmul(x[m * RL + q], twk[m * RL + q]);
    // the next transform's new half (its first half is this one's second, kept in registers): in flight across the transposed store
    if (overlap && b + 1 < t_end) {
      const int pxn = px + a.step;
#pragma unroll
      for (int m = 0; m < P / R0; m++)
#pragma unroll
        for (int s = H; s < R0; s++) {
          const int r = (pxn + NB * ((l + m * T) + s * (NA / R0)) + n2) & a.mask;
          pfw[m * H + s - H] = load_stream(&a.timf2w[r]); pfs[m * H + s - H] = load_stream(&a.timf2s[r]);
        }
    }
    __syncthreads();
    // store scratch[n2][k1], k1 fastest: thread t handles k1 = t mod NA of tile row t / NA, 1024/NA rows per sweep
    float2 *sc = a.scratch + (size_t)b * NA * NB + (size_t)blockIdx.x * LRH_TILE * NA;
    for (int e = threadIdx.x; e < LRH_TILE * NA; e += LRH_TILE * T) {
      const int cc = e / NA, k1 = e - cc * NA;
      store_stream(&sc[(size_t)cc * NA + k1], lds[cc * CS + k1]);
    }
    __syncthreads();                                     // the next transform reuses the buffer
  }
}

// Column step of 256 points with 16 points per thread (round 4).  The 4-point form above moves a transform through the LDS four times
// (three exchanges of the radix-4 passes and the transposed store, three barriers) with four loads in flight per lane.  Here
// n1 = 16 a + b, k1 = ka + 16 kb: thread (column c, b) loads a = 0..15 (sixteen loads in flight, each 16-lane group a whole 128-byte
// line), runs the 16-point transform over a in registers, multiplies by w_256^(b ka) -- and the ONE exchange that follows is at the same
// time the transposition the store needs: thread (ka, column c2) reads b = 0..15, runs the transform over b and holds bins
// ka + 16 kb of column c2, so that 16 consecutive lanes store 16 consecutive bins of scratch[n2][k1] (a whole line).  Cell
// (b, ka, c) sits at ((b 16 + ka) 16 + (c ^ ka)): 16 writing lanes (c = 0..15) fill 16 consecutive cells, 32 reading lanes (ka = 0..15 of
// two columns) hit 32 different 8-byte banks.  256 threads, 34 KB of LDS, four workgroups per CU in different phases of their cycle.
template <int LB, int TILE>
__global__ __launch_bounds__(16 * TILE, 64 / TILE) void k_fft2_cols16(Fft2BigArgs a)
{
  static_assert(TILE == 16 || TILE == 32, "tile columns");
  constexpr int NA = 256, NB = 1 << LB, LT = TILE == 16 ? 4 : 5;
  __shared__ float2 xch[16 * 16 * TILE];
  __shared__ float2 twl[NA];                             // e^{+2 pi j k / 256}
  const int tid0 = threadIdx.x;
  if (tid0 < NA) { const float2 w = a.tw_a[tid0]; twl[tid0] = make_float2(w.x, -w.y); }
  const int t_first = blockIdx.y * a.run, t_end = min(t_first + a.run, a.batch);
  const bool overlap = a.step * 2 == NA * NB;
  float2 twk[16], keep[8], pf[8];
  {
    const int n2s = blockIdx.x * TILE + (tid0 >> 4), ka = tid0 & 15;
#pragma unroll
    for (int kb = 0; kb < 16; kb++) {                    // w_N^(n2 k1), e^{+j}: conjugate of the forward table
      const float2 w = a.tw_big[(n2s * (ka + 16 * kb)) & (NA * NB - 1)];
      twk[kb] = make_float2(w.x, -w.y);
    }
  }
  // column index of a cell: 16 writing lanes (consecutive c, one ka) fill 16 consecutive cells; 32 reading lanes (ka = 0..15 of two
  // neighbouring columns) hit 32 different 8-byte banks (with 32 columns the two columns share bit 4: the parity of ka tells them apart)
  auto colx = [](int c, int ka) { return TILE == 16 ? (c ^ ka) : (c ^ ka ^ ((ka & 1) << 4)); };
  __syncthreads();
#pragma unroll 1
  for (int t = t_first; t < t_end; t++) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));                        // keeps the address arithmetic inside the loop (see k_timf2)
    const int c = tid & (TILE - 1), b = tid >> LT;       // load role: column c, samples n1 = 16 a + b
    const int n2 = blockIdx.x * TILE + c;
    const int px = a.px_first + t * a.step;
    const bool reuse = overlap && t > t_first;
    float2 x[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
      float2 raw;
      if (reuse) raw = q < 8 ? keep[q] : pf[q - 8];
      else {
        const int r = (px + NB * (16 * q + b) + n2) & a.mask;
        const float2 vw = load_stream(&a.timf2w[r]), vs = load_stream(&a.timf2s[r]);
        raw = make_float2(vw.x + vs.x, vw.y + vs.y);     // weak + strong (fft2.c:100-105)
      }
      const float w = a.window[NB * (16 * q + b) + n2];   // cache hits after the first transform
      x[q] = make_float2(w * raw.x, w * raw.y);
      if (q >= 8) keep[q - 8] = raw;
    }
    Dft<+1, 16>::run(x);                                  // over a: x[ka]
#pragma unroll
    for (int ka = 1; ka < 16; ka++) x[ka] = cmul(x[ka], twl[(b * ka) & (NA - 1)]);
#pragma unroll
    for (int ka = 0; ka < 16; ka++) xch[((b * 16 + ka) << LT) + colx(c, ka)] = x[ka];
    // the next transform's new half (its first half is this one's second, kept in registers): in flight across the exchange
    if (overlap && t + 1 < t_end) {
      const int pxn = px + a.step;
#pragma unroll
      for (int q = 8; q < 16; q++) {
        const int r = (pxn + NB * (16 * q + b) + n2) & a.mask;
        const float2 vw = load_stream(&a.timf2w[r]), vs = load_stream(&a.timf2s[r]);
        pf[q - 8] = make_float2(vw.x + vs.x, vw.y + vs.y);
      }
    }
    __syncthreads();
    const int ka = tid & 15, c2 = tid >> 4;              // store role: bins ka + 16 kb of column c2
#pragma unroll
    for (int bb = 0; bb < 16; bb++) x[bb] = xch[((bb * 16 + ka) << LT) + colx(c2, ka)];
    Dft<+1, 16>::run(x);                                  // over b: x[kb]
    float2 *sc = a.scratch + (size_t)t * NA * NB + (size_t)blockIdx.x * TILE * NA + (size_t)c2 * NA + ka;
#pragma unroll
    for (int kb = 0; kb < 16; kb++) store_stream(&sc[16 * kb], cmul(x[kb], twk[kb]));
    __syncthreads();                                     // the next transform reuses the buffer
  }
}

// FUSED: blockIdx.y is a waterfall averaging group instead of a transform; the workgroup walks the group's
// transforms and keeps sum |X|^2 of its bins in registers (k_fft2's scheme), so neither the fft2_power ring nor the
// k_powersum2 pass over it is needed.
// PPT: points per thread of a row transform.  sub_ppt(LB) (4 at 256 points: 64 threads a transform, 1024-thread workgroups, two per CU) or,
// at 256 points, 16: 16 threads a transform, 256-thread workgroups -- one exchange instead of three, sixteen loads in flight per lane
// instead of four, and four workgroups per CU in different phases of their load / transform / store cycle.
template <int LA, int LB, bool FUSED, int PPT>
__global__ __launch_bounds__(LRH_TILE * ((1 << LB) / PPT), PPT == 16 ? 4 : (LB <= 8 ? 8 : 4)) void k_fft2_rows(Fft2BigArgs a)    // row length <= 256: two workgroups per CU at 64 VGPRs; 512 (8 points per thread) spills 34 registers at that limit
{
  constexpr int P = PPT;
  using Plan = FftPlan<LB, P>;
  constexpr int NA = 1 << LA, NB = 1 << LB, T = Plan::T, R0 = Plan::R0, RL = Plan::RL;
  constexpr int CS = Plan::LDS_CELLS + 1;
  __shared__ float2 lds[LRH_TILE * CS];
  const int c = threadIdx.x & (LRH_TILE - 1), l = threadIdx.x >> 4;
  const int k1 = blockIdx.x * LRH_TILE + c;
  int t_first = blockIdx.y, t_end = t_first + 1;
  bool ps_continue = false, ps_complete = false;
  const int g = blockIdx.y;
  if constexpr (FUSED) {                                 // group arithmetic of k_powersum2
    t_first = g == 0 ? 0 : g * a.ps_avgnum - a.ps_counter;
    int count = a.ps_avgnum - (g == 0 ? a.ps_counter : 0);
    ps_complete = count <= a.batch - t_first;
    if (!ps_complete) count = a.batch - t_first;
    t_end = t_first + count;
    ps_continue = g == 0 && a.ps_counter > 0;
  }
  float acc[FUSED ? P : 1];
  if constexpr (FUSED) {
#pragma unroll
    for (int m = 0; m < P / RL; m++)
#pragma unroll
      for (int q = 0; q < RL; q++) acc[m * RL + q] = ps_continue ? a.ps_in[k1 + NA * ((l + m * T) + q * (NB / RL))] : 0.f;
  }
#pragma unroll 1
  for (int b = t_first; b < t_end; b++) {
    const float2 *sc = a.scratch + (size_t)b * NA * NB;
    float2 x[P];
#pragma unroll
    for (int m = 0; m < P / R0; m++)
#pragma unroll
      for (int s = 0; s < R0; s++) {
        const int n2 = (l + m * T) + s * (NB / R0);
        x[m * R0 + s] = load_stream(&sc[(size_t)n2 * NA + k1]);
      }
    BlockFft<LB, P, +1>::run(x, lds + c * CS, a.tw_b, l);
    const int na = (a.first_na + b) & a.na_mask;
    float2 *out = a.out + (size_t)na * NA * NB;
    float *pw = a.power + (size_t)na * NA * NB;
    if (a.keep_lo <= 0 && a.keep_hi >= NA * NB) {
#pragma unroll
      for (int m = 0; m < P / RL; m++)
#pragma unroll
        for (int q = 0; q < RL; q++) store_stream(&out[k1 + NA * ((l + m * T) + q * (NB / RL))], x[m * RL + q]);
    } else {
#pragma unroll
      for (int m = 0; m < P / RL; m++)
#pragma unroll
        for (int q = 0; q < RL; q++) { const int k = k1 + NA * ((l + m * T) + q * (NB / RL)); if (k >= a.keep_lo && k < a.keep_hi) store_stream(&out[k], x[m * RL + q]); }
    }
#pragma unroll
    for (int m = 0; m < P / RL; m++)
#pragma unroll
      for (int q = 0; q < RL; q++) {
        const int k2 = (l + m * T) + q * (NB / RL);
        const int k = k1 + NA * k2;
        const float2 v = x[m * RL + q];
        const float p2 = v.x * v.x + v.y * v.y;
        if constexpr (FUSED) acc[m * RL + q] = (b == t_first && !ps_continue) ? p2 : acc[m * RL + q] + p2;   // "=" then "+=" (fft2.c:655-670)
        else pw[k] = p2;
      }
    if constexpr (FUSED) __syncthreads();                // the next transform reuses the exchange buffer
  }
  if constexpr (FUSED) {
#pragma unroll
    for (int m = 0; m < P / RL; m++)
#pragma unroll
      for (int q = 0; q < RL; q++) {
        const int k = k1 + NA * ((l + m * T) + q * (NB / RL));
        if (ps_complete) a.wf_scratch[(size_t)g * NA * NB + k] = acc[m * RL + q];
        if (g == (int)gridDim.y - 1) a.ps_out[k] = acc[m * RL + q];
      }
  }
}

// Row step of 256 points in the layout of k_fft2_cols16 (round 4): 32 neighbouring k1 per workgroup (a 256-byte piece of every scratch row
// per load instruction instead of 128), n2 = 16 q + b with b = thread / 32, the 16-point transform over q in registers, ONE exchange
// (cell (b, ka, c) at (b 16 + ka) 32 + c: writers and readers both sweep consecutive cells), the 16-point transform over b, bins
// k2 = ka + 16 kb of column k1 with ka = thread / 32 -- k1 stays along the lanes, so every store is a 256-byte piece of an output row.
// The next transform's sixteen loads are issued behind the exchange.  Same sums, same order as k_fft2_rows<.., FUSED>.
template <int LA, bool FUSED>
__global__ __launch_bounds__(512, 2) void k_fft2_rows16x(Fft2BigArgs a)
{
  constexpr int NA = 1 << LA, NB = 256, TILE = 32;
  __shared__ float2 xch[16 * 16 * TILE];
  __shared__ float2 twl[NB];
  const int tid0 = threadIdx.x;
  if (tid0 < NB) { const float2 w = a.tw_b[tid0]; twl[tid0] = make_float2(w.x, -w.y); }
  int t_first = blockIdx.y, t_end = t_first + 1;
  bool ps_continue = false, ps_complete = false;
  const int g = blockIdx.y;
  if constexpr (FUSED) {                                 // group arithmetic of k_powersum2
    t_first = g == 0 ? 0 : g * a.ps_avgnum - a.ps_counter;
    int count = a.ps_avgnum - (g == 0 ? a.ps_counter : 0);
    ps_complete = count <= a.batch - t_first;
    if (!ps_complete) count = a.batch - t_first;
    t_end = t_first + count;
    ps_continue = g == 0 && a.ps_counter > 0;
  }
  const int k1 = blockIdx.x * TILE + (tid0 & (TILE - 1));
  float acc[FUSED ? 16 : 1];
  if constexpr (FUSED) {
#pragma unroll
    for (int kb = 0; kb < 16; kb++) acc[kb] = ps_continue ? a.ps_in[k1 + NA * ((tid0 >> 5) + 16 * kb)] : 0.f;
  }
  float2 nx[16];
  if (t_first < t_end) {
    const float2 *sc = a.scratch + (size_t)t_first * NA * NB;
#pragma unroll
    for (int q = 0; q < 16; q++) nx[q] = load_stream(&sc[(size_t)(16 * q + (tid0 >> 5)) * NA + k1]);
  }
  __syncthreads();
#pragma unroll 1
  for (int t = t_first; t < t_end; t++) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int c = tid & (TILE - 1), hi = tid >> 5;       // hi: b when loading, ka when storing
    float2 x[16];
#pragma unroll
    for (int q = 0; q < 16; q++) x[q] = nx[q];
    Dft<+1, 16>::run(x);                                  // over q: x[ka]
#pragma unroll
    for (int ka = 1; ka < 16; ka++) x[ka] = cmul(x[ka], twl[(hi * ka) & (NB - 1)]);
#pragma unroll
    for (int ka = 0; ka < 16; ka++) xch[((hi * 16 + ka) << 5) + c] = x[ka];
    if (t + 1 < t_end) {
      const float2 *sc = a.scratch + (size_t)(t + 1) * NA * NB;
#pragma unroll
      for (int q = 0; q < 16; q++) nx[q] = load_stream(&sc[(size_t)(16 * q + hi) * NA + k1]);
    }
    __syncthreads();
#pragma unroll
    for (int bb = 0; bb < 16; bb++) x[bb] = xch[((bb * 16 + hi) << 5) + c];
    Dft<+1, 16>::run(x);                                  // over b: x[kb], bin k2 = hi + 16 kb
    const int na = (a.first_na + t) & a.na_mask;
    float2 *out = a.out + (size_t)na * NA * NB;
    float *pw = a.power + (size_t)na * NA * NB;
    if (a.keep_lo <= 0 && a.keep_hi >= NA * NB) {
#pragma unroll
      for (int kb = 0; kb < 16; kb++) store_stream(&out[k1 + NA * (hi + 16 * kb)], x[kb]);
    } else {
#pragma unroll
      for (int kb = 0; kb < 16; kb++) { const int k = k1 + NA * (hi + 16 * kb); if (k >= a.keep_lo && k < a.keep_hi) store_stream(&out[k], x[kb]); }
    }
#pragma unroll
    for (int kb = 0; kb < 16; kb++) {
      const float2 v = x[kb];
      const float p2 = v.x * v.x + v.y * v.y;
      if constexpr (FUSED) acc[kb] = (t == t_first && !ps_continue) ? p2 : acc[kb] + p2;   // "=" then "+=" (fft2.c:655-670)
      else pw[k1 + NA * (hi + 16 * kb)] = p2;
    }
    __syncthreads();                                     // the next transform reuses the exchange buffer
  }
  if constexpr (FUSED) {
#pragma unroll
    for (int kb = 0; kb < 16; kb++) {
      const int k = k1 + NA * ((tid0 >> 5) + 16 * kb);
      if (ps_complete) a.wf_scratch[(size_t)g * NA * NB + k] = acc[kb];
      if (g == (int)gridDim.y - 1) a.ps_out[k] = acc[kb];
    }
  }
}

// fft2_powersum_float (fft2.c:655-670): group g = one waterfall averaging period; complete groups are parked in
// wf_scratch for k_waterfall, the last (possibly partial) group is what fft2_powersum_float holds afterwards.
__global__ __launch_bounds__(256) void k_powersum2(Powersum2Args a)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int g = blockIdx.y;
  const int start = g == 0 ? 0 : g * a.avgnum - a.counter;
  int count = a.avgnum - (g == 0 ? a.counter : 0);
  const bool complete = count <= a.count - start;
  if (!complete) count = a.count - start;
  const bool accumulate = g == 0 && a.counter > 0;
  float acc = accumulate ? a.powersum_in[i] : 0.f;
  for (int b = 0; b < count; b++) {
    const float pw = a.power[(size_t)((a.first_na + start + b) & a.na_mask) * a.n + i];
    acc = (b == 0 && !accumulate) ? pw : acc + pw;
  }
  if (complete) a.wf_scratch[(size_t)g * a.n + i] = acc;
  if (g == (int)gridDim.y - 1) a.powersum_out[i] = acc;
}

// Two coupled channels (fft2.c:1622-1640): per transform and bin the TWOCHAN_POWER cross products of the two channels'
// spectra, summed over a waterfall averaging group (fft2_xysum) in the reference's order; group arithmetic of
// k_powersum2.  A completed group leaves the power the two-channel waterfall line shows (fft2.c:1700-1712),
// (x2+y2) + 2 (re_xy^2 + im_xy^2 - x2 y2) / (x2+y2), in `lines` for k_waterfall.  One bin per lane: 8 + 8 bytes in,
// 16 bytes out per transform, all coalesced.
__global__ __launch_bounds__(256) void k_xypower(XyArgs a)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int g = blockIdx.y;
  const int start = g == 0 ? 0 : g * a.avgnum - a.counter;
  int count = a.avgnum - (g == 0 ? a.counter : 0);
  const bool complete = count <= a.batch - start;
  if (!complete) count = a.batch - start;
  const bool accumulate = g == 0 && a.counter > 0;
  float4 acc = accumulate ? a.sum_in[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int AHEAD = 8;                               // loads of that many transforms in flight (the sums themselves stay in transform order)
  for (int b0 = 0; b0 < count; b0 += AHEAD) {
  float2 xs[AHEAD], ys[AHEAD];
#pragma unroll
  for (int u = 0; u < AHEAD; u++)
    if (b0 + u < count) { const size_t t = (size_t)(start + b0 + u) * a.n + i; xs[u] = a.x[t]; ys[u] = a.y[t]; }
#pragma unroll
  for (int u = 0; u < AHEAD; u++) {
    const int b = b0 + u;
    if (b >= count) break;
    const float2 x = xs[u], y = ys[u];
    float4 v;
    v.x = x.x * x.x + x.y * x.y;
    v.y = y.x * y.x + y.y * y.y;
    v.z = -x.x * y.y + x.y * y.x;
    v.w = x.x * y.x + x.y * y.y;
    if (a.xypower) a.xypower[(size_t)((a.first_na + start + b) & a.na_mask) * a.n + i] = v;   // null: cfg.fft2_float_sparse, only the sums are kept
    if (b == 0 && !accumulate) acc = v;
    else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  }
  }
  if (complete) {
    const float t1 = acc.x + acc.y;
    // t1 == 0 (a bin empty in both channels): the reference's 0/0 ends, through the int conversion of a NaN on x86
    // (INT_MIN) and the clamp, at -32767, where a zero power ends too
    a.lines[(size_t)g * a.n + i] = t1 > 0.f ? t1 + 2 * (acc.w * acc.w + acc.z * acc.z - acc.x * acc.y) / t1 : 0.f;
  }
  if (g == (int)gridDim.y - 1) a.sum_out[i] = acc;
}

// one waterfall line, 0.01 dB shorts (fft2.c:707-812); itab[] holds the reference's float-accumulated yfac index
// 1000 log10(power) and the conversion to int the way the reference's host code gets them (fft2.c:707-815 runs on x86): a bin or a group with NO power
// -- silent input, a pixel beyond the end of the spectrum -- is log10(0) = -inf there, which cvttsd2si turns into INT_MIN and the clamp into -32767, and an
// interpolation that meets it carries -inf / NaN on.  The device's double log10 returns a finite -2.065 for 0 (seen: -2065 in those pixels where the
// reference has -32767; found by tests/test_gpu_random_configs.py), and its conversions saturate (+inf -> INT_MAX, NaN -> 0) where x86 answers INT_MIN.
__device__ __forceinline__ double wf_db(float v) { return v == 0.f ? -(double)INFINITY : 1000. * log10((double)v); }
__device__ __forceinline__ int wf_int(double v) { return (v != v || v >= 2147483648. || v < -2147483648.) ? (-2147483647 - 1) : (int)v; }
__global__ __launch_bounds__(256) void k_waterfall(WaterfallArgs a)
{
  const int t = blockIdx.x * 256 + threadIdx.x;
  {
    const int l = blockIdx.y;
    int ptr = a.ptr0 - l * a.npix;
    ptr %= a.wf_size; if (ptr < 0) ptr += a.wf_size;   // update_wg_waterf, fft1.c:104-113
    a.line += ptr; a.ps += (size_t)l * a.line_stride;
  }
  if (a.hx == 1 || a.hp == 1) {
    if (t >= a.npix) return;
    int y = wf_int(wf_db(a.ps[a.first + t] * a.yfac[a.itab[t]]));
    if (y < -32767) y = -32767; if (y > 32767) y = 32767;
    a.line[t] = (int16_t)y;
  } else if (a.hx == 0) {
    // segment t covers pixels t*hp+1 .. t*hp+hp between data points t and t+1
    const int nseg = (a.npix - a.hp + a.hp - 1) / a.hp;      // loop count of fft2.c:752
    const int i1 = a.first + t + 1;
    const bool tail = (t == nseg) && (i1 < a.siz);
    if (t > nseg || (t == nseg && !tail)) return;
    const int i0 = a.first + t;
    float y0 = (float)wf_db(a.ps[i0] * a.yfac[a.itab[t]]);
    if (t == 0) {
      int y = wf_int(wf_db(a.ps[i0] * a.yfac[a.itab[0]]));
      y0 = (float)y;                                        // yval=y with y already an int (fft2.c:745-746)
      if (y < -32767) y = -32767; if (y > 32767) y = 32767;
      a.line[0] = (int16_t)y;
    }
    const float r1 = (float)wf_db(a.ps[i1] * a.yfac[a.itab[t + 1]]);
    const float der = (r1 - y0) / a.hp;
    float yval = y0;
    for (int k = t * a.hp + 1; k <= t * a.hp + a.hp; k++) {
      yval = yval + der;
      int y = wf_int((double)yval);
      if (y < -32767) y = -32767; if (y > 32767) y = 32767;
      if (k < a.npix) a.line[k] = (int16_t)y;
    }
  } else {
    if (t >= a.npix) return;
    int ia = a.first + t * a.hx, ib = ia + a.hx;
    // reference clamps ib to siz once it reaches it (fft2.c:807-809)
    if (ia > a.siz) ia = a.siz; if (ib >= a.siz) ib = a.siz;
    float r2 = 0;
    for (int i = ia; i < ib; i++) { const float r1 = a.ps[i]; if (r1 > r2) r2 = r1; }
    int y = wf_int(wf_db(a.yfac[a.itab[t]] * r2));
    if (y < -32767) y = -32767; if (y > 32767) y = 32767;
    a.line[t] = (int16_t)y;
  }
}

// =====================================================================================================
// mix1
// =====================================================================================================
// gather mix1.size bins around mix1_point (mix1.c:955-983), frequency-domain window (mix1.c:113-135), fftback (fft0.c:481)
template <int LOG2N>
__global__ __launch_bounds__(fft_threads(LOG2N), fft_min_waves(LOG2N)) void k_mix1_back(Mix1Args a)
{
  constexpr int P = points_per_thread(LOG2N);
  using Plan = FftPlan<LOG2N, P>;
  constexpr int N = Plan::N, T = Plan::T, R0 = Plan::R0, RL = Plan::RL;
  __shared__ float2 lds[Plan::LDS_CELLS];
  const int tid = threadIdx.x, b = blockIdx.x;
  const float2 *z = a.fft2 + (size_t)((a.first_nx + b) & a.nx_mask) * a.n2;
  float2 x[P];
#pragma unroll
  for (int m = 0; m < P / R0; m++)
#pragma unroll
    for (int s = 0; s < R0; s++) {
      const int i = (tid + m * T) + s * (N / R0);
      const int off = i < N / 2 ? i : i - N;                // upper half first, then the lower half
      const int bin = (a.points ? a.points[b] : a.point) + off;
      const int d = off < 0 ? -off : off;
      const float w = a.fqwin[d == 0 ? N / 2 - 1 : N / 2 - d];
      float2 v = make_float2(0.f, 0.f);
      // upper half is cut at the band edge, lower half at bin 0 (mix1.c:956-958, 971-973)
      if (off >= 0 ? (bin < a.lim_hi) : (bin >= 0)) v = z[bin];
      x[m * R0 + s] = make_float2(v.x * w, v.y * w);
    }
  BlockFft<LOG2N, P, -1>::run(x, lds, a.tw, tid);
  float2 *o = a.scratch + (size_t)b * N;
#pragma unroll
  for (int m = 0; m < P / RL; m++)
#pragma unroll
    for (int q = 0; q < RL; q++) o[(tid + m * T) + q * (N / RL)] = x[m * RL + q];
}

// rotate + overlap into timf3 (mix1.c:141-195); phases follow the host's float recursion (chunk starts uploaded)
__global__ __launch_bounds__(256) void k_mix1_out(Mix1OutArgs a, int batch)
{
  const int half = a.xover > 0 ? a.block : (a.overlap ? a.nm / 2 : a.nm);
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= half) return;
  const int pos = (a.pa_first + b * a.block + i) & a.mask2;
  if (!a.selected) { a.timf3[pos] = make_float2(0.f, 0.f); return; }     // mix1_clear
  if (a.xover > 0) {
    // Crossover windows (mix1.c:196-262, dfq = 0; mix2.c:177-216 without the rotation): sample i of the block comes from back-transform
    // point i + k0; the first xover samples blend with the raw tail parked by the previous transform, the rest is divided by the
    // window (win = 1/window, walked up to the centre and down again), and xover raw points are parked beyond.
    const int X = a.xover, k0 = a.im / 2 - X / 2;
    float t3 = 0.f, t4 = 1.f, r1 = 0.f;
    if (a.rotate) {
      const float2 inc = a.ph_inc[b];
      const float2 st = a.ph_start[(size_t)b * a.nchunks + i / LRH_PH_CHUNK];
      float t1 = st.x; r1 = st.y;
      for (int j = 0; j < (i & (LRH_PH_CHUNK - 1)); j++) { t1 += inc.x; r1 += inc.y; }
      t3 = (float)sin((double)t1); t4 = (float)cos((double)t1);
    }
    const float2 nw = a.scratch[(size_t)b * a.nm + i + k0];
    if (i < X) {
      const float2 raw = (b == 0) ? a.timf3[pos] : a.scratch[(size_t)(b - 1) * a.nm + a.block + i + k0];
      const float w1 = a.sin2win[i], w2 = a.cos2win[i];
      if (a.rotate) {
        const float r3 = (float)sin((double)r1), r4 = (float)cos((double)r1);
        const float a1 = w2 * raw.x, a2 = w2 * raw.y;
        a.timf3[pos] = make_float2(r4 * a1 - r3 * a2 + (t4 * nw.x - t3 * nw.y) * w1,
                                   r4 * a2 + r3 * a1 + (t4 * nw.y + t3 * nw.x) * w1);
      } else a.timf3[pos] = make_float2(raw.x * w2 + nw.x * w1, raw.y * w2 + nw.y * w1);       // mix2.c:187-188
    } else {
      const int ib = a.block / 2 + 1 + X / 2;              // complex index where the window walk turns (mix1.c:226, mix2.c:191)
      const int j = i < ib ? k0 + i : k0 + 2 * ib - 2 - i;
      const float rw = a.win[j];
      if (a.rotate) a.timf3[pos] = make_float2((t4 * nw.x - t3 * nw.y) * rw, (t4 * nw.y + t3 * nw.x) * rw);
      else a.timf3[pos] = make_float2(nw.x * rw, nw.y * rw);
    }
    if (b == batch - 1 && i < X)                           // raw tail for the next call (mix1.c:253-261, mix2.c:208-215)
      a.timf3[(a.pa_first + batch * a.block + i) & a.mask2] = a.scratch[(size_t)b * a.nm + a.block + i + k0];
    return;
  }
  const float2 nw = a.scratch[(size_t)b * a.nm + i];
  if (!a.rotate) {                                           // mix2: baseb_raw += first half, raw second half parked
    if (!a.overlap) { a.timf3[pos] = nw; return; }
    const float2 old = (b == 0) ? a.timf3[pos] : a.scratch[(size_t)(b - 1) * a.nm + half + i];
    a.timf3[pos] = make_float2(old.x + nw.x, old.y + nw.y);
    if (b == batch - 1) a.timf3[(a.pa_first + batch * a.block + i) & a.mask2] = a.scratch[(size_t)b * a.nm + half + i];
    return;
  }
  // the same float additions as the reference's recursion (mix1.c:172-186), restarted every LRH_PH_CHUNK samples
  const float2 inc = a.ph_inc[b];
  const float2 st = a.ph_start[(size_t)b * a.nchunks + i / LRH_PH_CHUNK];
  float t1 = st.x, r1 = st.y;
  for (int j = 0; j < (i & (LRH_PH_CHUNK - 1)); j++) { t1 += inc.x; r1 += inc.y; }
  const float t3 = (float)sin((double)t1), t4 = (float)cos((double)t1);
  if (!a.overlap) {
    a.timf3[pos] = make_float2(t4 * nw.x - t3 * nw.y, t4 * nw.y + t3 * nw.x);
    return;
  }
  const float2 old = (b == 0) ? a.timf3[pos] : a.scratch[(size_t)(b - 1) * a.nm + half + i];
  const float r3 = (float)sin((double)r1), r4 = (float)cos((double)r1);
  a.timf3[pos] = make_float2(r4 * old.x - r3 * old.y + t4 * nw.x - t3 * nw.y,
                             r4 * old.y + r3 * old.x + t4 * nw.y + t3 * nw.x);
  if (b == batch - 1)                                        // raw second half parked at the next block (mix1.c:188-194)
    a.timf3[(a.pa_first + batch * a.block + i) & a.mask2] = a.scratch[(size_t)b * a.nm + half + i];
}

// make_fft3_all, transform part (fft3.c:240-283): window, e^{+j} transform (no conjugation), DC moved to N/2
template <int LOG2N>
__global__ __launch_bounds__(fft_threads(LOG2N), fft_min_waves(LOG2N)) void k_fft3(Fft3Args a)
{
  constexpr int P = points_per_thread(LOG2N);
  using Plan = FftPlan<LOG2N, P>;
  constexpr int N = Plan::N, T = Plan::T, R0 = Plan::R0, RL = Plan::RL;
  __shared__ float2 lds[Plan::LDS_CELLS];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int px = a.px_first + b * a.step;
  float2 x[P];
#pragma unroll
  for (int m = 0; m < P / R0; m++)
#pragma unroll
    for (int s = 0; s < R0; s++) {
      const int idx = (tid + m * T) + s * (N / R0);
      const float2 v = a.timf3[(px + idx) & a.mask];
      const float w = a.window[idx];
      x[m * R0 + s] = make_float2(v.x * w, v.y * w);
    }
  BlockFft<LOG2N, P, +1>::run(x, lds, a.tw, tid);
  float2 *out = a.out + (size_t)((a.first_slot + b) & a.slot_mask) * N;
#pragma unroll
  for (int m = 0; m < P / RL; m++)
#pragma unroll
    for (int q = 0; q < RL; q++) {
      const int k = (tid + m * T) + q * (N / RL);
      out[(k + N / 2) & (N - 1)] = x[m * RL + q];
    }
}

// fft3_mix2 mixer_mode 1 (mix2.c:145-157): mix2.size bins around fft3_size/2 times bg_filterfunc, fftback
template <int LOG2N>
__global__ __launch_bounds__(fft_threads(LOG2N), fft_min_waves(LOG2N)) void k_mix2_back(Mix2Args a)
{
  constexpr int P = points_per_thread(LOG2N);
  using Plan = FftPlan<LOG2N, P>;
  constexpr int N = Plan::N, T = Plan::T, R0 = Plan::R0, RL = Plan::RL;
  __shared__ float2 lds[Plan::LDS_CELLS];
  const int tid = threadIdx.x, b = blockIdx.x;
  const float2 *z = a.fft3 + (size_t)((a.first_slot + b) & a.slot_mask) * a.n3;
  float2 x[P];
#pragma unroll
  for (int m = 0; m < P / R0; m++)
#pragma unroll
    for (int s = 0; s < R0; s++) {
      const int i = (tid + m * T) + s * (N / R0);
      const int bin = a.n3 / 2 + (i < N / 2 ? i : i - N);     // positive offsets first, then -N/2..-1
      const float2 v = a.pol ? a.pol[(size_t)b * N + (bin - a.n3 / 2 + N / 2)] : z[bin];
      const float w = a.filt[bin];
      x[m * R0 + s] = make_float2(v.x * w, v.y * w);
    }
  BlockFft<LOG2N, P, -1>::run(x, lds, a.tw, tid);
  float2 *o = a.scratch + (size_t)b * N;
#pragma unroll
  for (int m = 0; m < P / RL; m++)
#pragma unroll
    for (int q = 0; q < RL; q++) o[(tid + m * T) + q * (N / RL)] = x[m * RL + q];
}

// fft3_mix2 with bg.mixer_mode = 2 (mix2.c:217-246): baseband sample k (1 .. new_points) of a transform is a symmetric FIR on timf3 centred
// 1 - pts + fft3_size - fft3_new_points + k resamp samples behind timf3_py, the taps added from the centre outwards in the reference's order
__global__ __launch_bounds__(64) void k_mix2_fir(Mix2FirArgs a)
{
  const int k = blockIdx.x * 64 + threadIdx.x + 1, b = blockIdx.y;
  if (k > a.nm2new) return;
  const int c0 = a.py_first + b * a.step + 1 - a.pts + a.n3 - a.m3 + k * a.resamp, h = a.pts / 2;
  const float2 x0 = a.timf3[c0 & a.mask];
  float t1 = x0.x * a.fir[h], t2 = x0.y * a.fir[h];
  for (int i = h - 1, j = 1; i >= 0; i--, j++) {
    const float2 xp = a.timf3[(c0 + j) & a.mask], xm = a.timf3[(c0 - j) & a.mask];
    const float f = a.fir[i];
    t1 += (xp.x + xm.x) * f; t2 += (xp.y + xm.y) * f;
  }
  a.baseb[(a.pa_first + b * a.nm2new + k - 1) & a.bmask] = make_float2(t1, t2);
}
hipError_t launch_mix2_fir(const Mix2FirArgs &a, int batch, hipStream_t st)
{
  hipLaunchKernelGGL(k_mix2_fir, dim3((a.nm2new + 63) / 64, batch), dim3(64), 0, st, a);
  return hipGetLastError();
}

// Two coupled channels, polarisation transform of fft3_mix2 (mix2.c:340-343, 377-380): this channel's bins times its
// complex weight in A and in B; the two channels' shares are summed by the caller's all-reduce.
__global__ __launch_bounds__(256) void k_pol(PolArgs a)
{
  const int j = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (j >= a.nm) return;
  const float2 v = a.fft3[(size_t)((a.first_slot + b) & a.slot_mask) * a.n3 + (a.n3 / 2 - a.nm / 2 + j)];
  const size_t o = (size_t)b * a.nm + j;
  a.out[o] = make_float2(a.wa.x * v.x - a.wa.y * v.y, a.wa.x * v.y + a.wa.y * v.x);
  a.out[(size_t)a.batch * a.nm + o] = make_float2(a.wb.x * v.x - a.wb.y * v.y, a.wb.x * v.y + a.wb.y * v.x);
}

// weak-signal power per block of released timf2 data (wcw.c:84-113), one workgroup per block
__global__ __launch_bounds__(256) void k_blockpower(BlockpowerArgs a)
{
  __shared__ double red[256];
  const int base = a.first + blockIdx.x * a.block;
  double acc = 0;
  for (int i = threadIdx.x; i < a.block; i += 256) { const float2 v = a.timf2w[(base + i) & a.mask]; acc += (double)(v.x * v.x + v.y * v.y); }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) a.out[(a.out_first + blockIdx.x) & a.out_mask] = (float)red[0];
}

// =====================================================================================================
// launchers
// =====================================================================================================
#define LRH_DISPATCH(KERNEL, LOG2N, LO, HI, ...)                                                             \
  switch (LOG2N) {                                                                                           \
    case 3: if constexpr (LO <= 3) { KERNEL(3, __VA_ARGS__); } break;                                                  \
    case 4: if constexpr (LO <= 4) { KERNEL(4, __VA_ARGS__); } break;                                                  \
    case 5: if constexpr (LO <= 5) { KERNEL(5, __VA_ARGS__); } break;                                                  \
    case 6: KERNEL(6, __VA_ARGS__); break;                                                                   \
    case 7: KERNEL(7, __VA_ARGS__); break;                                                                   \
    case 8: KERNEL(8, __VA_ARGS__); break;                                                                   \
    case 9: KERNEL(9, __VA_ARGS__); break;                                                                   \
    case 10: KERNEL(10, __VA_ARGS__); break;                                                                 \
    case 11: KERNEL(11, __VA_ARGS__); break;                                                                 \
    case 12: KERNEL(12, __VA_ARGS__); break;                                                                 \
    case 13: KERNEL(13, __VA_ARGS__); break;                                                                 \
    case 14: KERNEL(14, __VA_ARGS__); break;                                                                 \
    default: return hipErrorInvalidValue;                                                                    \
  }

#define LRH_LAUNCH_FFT1_V(L, DW, SK, a, batch, st) \
  hipLaunchKernelGGL((k_fft1<L, DW, SK>), dim3(fftl_grid<L>(batch, a.spare_cus)), dim3(fft1_threads(L)), 0, st, a)
#define LRH_LAUNCH_FFT1(L, a, batch, st)                                                    \
  do {                                                                                      \
    const bool sk = a.shift_i != 0 || a.shift_q != 0;                                       \
    if (a.real && !a.dword) hipLaunchKernelGGL((k_fft1<L, false, false, true>), dim3(fftl_grid<L>(batch, a.spare_cus)), dim3(fft1_threads(L)), 0, st, a); \
    else if (a.real) hipLaunchKernelGGL((k_fft1<L, true, false, true>), dim3(fftl_grid<L>(batch, a.spare_cus)), dim3(fft1_threads(L)), 0, st, a); \
    else if (!a.dword && !sk) LRH_LAUNCH_FFT1_V(L, false, false, a, batch, st);                  \
    else if (!a.dword) LRH_LAUNCH_FFT1_V(L, false, true, a, batch, st);                     \
    else if (!sk) LRH_LAUNCH_FFT1_V(L, true, false, a, batch, st);                          \
    else LRH_LAUNCH_FFT1_V(L, true, true, a, batch, st);                                    \
  } while (0)
#define LRH_LAUNCH_TIMF2(L, a, batch, st)                                                                   \
  do {                                                                                                      \
    if (a.ss_ring) {                                                                                        \
      a.ss_run = (batch + fftl_grid<L>(batch, a.spare_cus) - 1) / fftl_grid<L>(batch, a.spare_cus);                                   \
      if (a.ss_run >= batch) a.ss_part = nullptr;      /* one workgroup: the sums go straight to the ring, no join */           \
      a.ss_split = batch == 1 && L >= 12;                                                                         \
      hipLaunchKernelGGL((k_timf2<L, 1, true>), dim3(a.ss_split ? 2 : (batch + a.ss_run - 1) / a.ss_run), dim3(fft1_threads(L)), 0, st, a); \
    }                                                                                                       \
    else if (a.mode == 1) hipLaunchKernelGGL((k_timf2<L, 1, false>), dim3(fftl_grid<L>(batch, a.spare_cus)), dim3(fft1_threads(L)), 0, st, a);      \
    else if (a.mode == 0) hipLaunchKernelGGL((k_timf2<L, 0, false>), dim3(fftl_grid<L>(batch, a.spare_cus)), dim3(fft1_threads(L)), 0, st, a); \
    else hipLaunchKernelGGL((k_timf2<L, 2, false>), dim3(fftl_grid<L>(batch, a.spare_cus)), dim3(fft1_threads(L)), 0, st, a);                  \
  } while (0)
#define LRH_LAUNCH_FFT2(L, a, batch, st)                                                                              \
  do {                                                                                                                \
    if constexpr (L <= LRH_FFT2_FUSED_MAXLOG) {                                                                       \
      if (a.ps_avgnum > 0) {                                                                                          \
        hipLaunchKernelGGL((k_fft2<L, true>), dim3((a.ps_counter + batch + a.ps_avgnum - 1) / a.ps_avgnum),           \
                           dim3(FftPlan<L, points_per_thread(L)>::T), 0, st, a);                                      \
        break;                                                                                                        \
      }                                                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL((k_fft2<L, false>), dim3((batch + a.run - 1) / a.run), dim3(FftPlan<L, points_per_thread(L)>::T), 0, st, a); \
  } while (0)
#define LRH_LAUNCH_MIX1(L, a, batch, st) \
  hipLaunchKernelGGL((k_mix1_back<L>), dim3(batch), dim3(FftPlan<L, points_per_thread(L)>::T), 0, st, a)

// workgroups that fit on the chip at once for a transform kernel (LDS and thread limits, 256 CUs)
// spare_cus: compute units left without a workgroup of a kernel that takes a whole unit's LDS, spread over the XCDs (workgroup i goes
// to XCD i mod 8).  A one-workgroup side kernel with a large LDS footprint of its own (the limiter) otherwise waits for the whole
// persistent kernel to end before it can start.
static int persistent_grid(int lds_bytes, int threads, int batch, int spare_cus = 0)
{
  int per_cu = 160 * 1024 / lds_bytes; if (per_cu > 2048 / threads) per_cu = 2048 / threads; if (per_cu > 8) per_cu = 8; if (per_cu < 1) per_cu = 1;
  const int g = (256 - (per_cu == 1 ? spare_cus : 0)) * per_cu;
  return batch < g ? batch : g;
}
template <int L> static int fftl_grid(int batch, int spare_cus = 0) { return persistent_grid(8 * BlockFftL<L, points_fft1(L), 1>::LDS_CELLS, fft1_threads(L), batch, spare_cus); }

hipError_t launch_fft1(int log2n, const Fft1Args &a, int batch, hipStream_t st)
{
  LRH_DISPATCH(LRH_LAUNCH_FFT1, log2n, 6, 14, a, batch, st);
  return hipGetLastError();
}
#define LRH_TIMF2_GRID(L, out, batch) out = fftl_grid<L>(batch)
int timf2_grid(int log2n, int batch)
{
  int g = 0;
  switch (log2n) {
    case 6: LRH_TIMF2_GRID(6, g, batch); break; case 7: LRH_TIMF2_GRID(7, g, batch); break; case 8: LRH_TIMF2_GRID(8, g, batch); break;
    case 9: LRH_TIMF2_GRID(9, g, batch); break; case 10: LRH_TIMF2_GRID(10, g, batch); break; case 11: LRH_TIMF2_GRID(11, g, batch); break;
    case 12: LRH_TIMF2_GRID(12, g, batch); break; case 13: LRH_TIMF2_GRID(13, g, batch); break; case 14: LRH_TIMF2_GRID(14, g, batch); break;
    default: break;
  }
  return g;
}
// With `ss` (mode 1 only) the launch also produces fft1_c's power sums of the same transforms: see k_timf2<.., SS>.
hipError_t launch_timf2(int log2n, const Timf2Args &a0, int batch, hipStream_t st, const SumsqArgs *ss, float *ss_part, int *ss_run)
{
  Timf2Args a = a0; a.batch = batch;
  a.ss_ring = nullptr;
  if (ss) {
    if (a.mode != 1 || ss->batch != batch || ss->first_nb != a.first_nb || ss->n != (1 << log2n) || !ss_part) return hipErrorInvalidValue;
    a.ss_ring = ss->sumsq; a.ss_part = ss_part; a.ss_mask = ss->sumsq_mask; a.ss_avg = ss->avg; a.ss_c0 = ss->c0; a.ss_pa0 = ss->pa0;
  }
  LRH_DISPATCH(LRH_LAUNCH_TIMF2, log2n, 6, 14, a, batch, st);
  if (ss && ss_run) *ss_run = a.ss_part ? a.ss_run : 0;        // 0: nothing left for k_sumsq_join
  return hipGetLastError();
}
// k_fft1w + the strong-only pass of k_timf2 (fft1_size 16384): `run` out = transforms per workgroup (for k_sumsq_join)
hipError_t launch_fft1w(const Fft1wArgs &a0, hipStream_t st, int *run)
{
  Fft1wArgs a = a0;
  const int grid = persistent_grid(8 * BlockFftL<14, 32, 1>::LDS_CELLS, 512, a.batch, a.spare_cus);
  a.run = (a.batch + grid - 1) / grid;
  if (run) *run = a.run;
  hipLaunchKernelGGL((k_fft1w<14>), dim3((a.batch + a.run - 1) / a.run), dim3(512), 0, st, a);
  return hipGetLastError();
}
// k_fft1v: fft1_size 4096 / 8192 / 16384, int16 or int32 I/Q
hipError_t launch_fft1v(int log2n, bool dword, bool real, const Fft1wArgs &a0, hipStream_t st, int *run)
{
  Fft1wArgs a = a0;
  const int lds = 8 * (log2n == 14 ? Fft1vGeom<14>::LDS_CELLS : (log2n == 13 ? Fft1vGeom<13>::LDS_CELLS : Fft1vGeom<12>::LDS_CELLS));
  const int threads = (1 << log2n) / 32;
  // 235+ VGPRs: two waves per SIMD, i.e. 512 threads per CU whatever the LDS would allow
  int per_cu = 160 * 1024 / lds; if (per_cu > 512 / threads) per_cu = 512 / threads; if (per_cu < 1) per_cu = 1;
  int grid = (256 - (per_cu == 1 ? a.spare_cus : 0)) * per_cu; if (grid > a.batch) grid = a.batch;
  if (a.max_wg > 0 && grid > a.max_wg) grid = a.max_wg;
  a.run = (a.batch + grid - 1) / grid;
  if (run) *run = a.run;
  const dim3 g((a.batch + a.run - 1) / a.run), t(threads);
  static const int exp_ = getenv("LRH_FFT1V_EXP") ? atoi(getenv("LRH_FFT1V_EXP")) : 0;
  static const int stagger_ = getenv("LRH_V_STAGGER") ? atoi(getenv("LRH_V_STAGGER")) : 0;
  a.stagger = a.run >= 4 ? stagger_ : 0;
  if (exp_ && log2n == 14 && !dword && !real) {
    if (exp_ == 1) hipLaunchKernelGGL((k_fft1v<14, false, false, 1>), g, t, 0, st, a);
    else if (a.keep_spec) hipLaunchKernelGGL((k_fft1v<14, false, true, 2>), g, t, 0, st, a);
    else hipLaunchKernelGGL((k_fft1v<14, false, false, 2>), g, t, 0, st, a);
    return hipGetLastError();
  }
  if (real) {
    switch (log2n * 4 + (dword ? 2 : 0) + (a.keep_spec ? 1 : 0)) {
      case 56: hipLaunchKernelGGL((k_fft1v<14, false, false, 0, true>), g, t, 0, st, a); break;
      case 57: hipLaunchKernelGGL((k_fft1v<14, false, true, 0, true>), g, t, 0, st, a); break;
      case 58: hipLaunchKernelGGL((k_fft1v<14, true, false, 0, true>), g, t, 0, st, a); break;
      case 59: hipLaunchKernelGGL((k_fft1v<14, true, true, 0, true>), g, t, 0, st, a); break;
      case 52: hipLaunchKernelGGL((k_fft1v<13, false, false, 0, true>), g, t, 0, st, a); break;
      case 53: hipLaunchKernelGGL((k_fft1v<13, false, true, 0, true>), g, t, 0, st, a); break;
      case 54: hipLaunchKernelGGL((k_fft1v<13, true, false, 0, true>), g, t, 0, st, a); break;
      case 55: hipLaunchKernelGGL((k_fft1v<13, true, true, 0, true>), g, t, 0, st, a); break;
      case 48: hipLaunchKernelGGL((k_fft1v<12, false, false, 0, true>), g, t, 0, st, a); break;
      case 49: hipLaunchKernelGGL((k_fft1v<12, false, true, 0, true>), g, t, 0, st, a); break;
      case 50: hipLaunchKernelGGL((k_fft1v<12, true, false, 0, true>), g, t, 0, st, a); break;
      case 51: hipLaunchKernelGGL((k_fft1v<12, true, true, 0, true>), g, t, 0, st, a); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (log2n * 4 + (dword ? 2 : 0) + (a.keep_spec ? 1 : 0)) {
    case 56: hipLaunchKernelGGL((k_fft1v<14, false, false>), g, t, 0, st, a); break;
    case 57: hipLaunchKernelGGL((k_fft1v<14, false, true>), g, t, 0, st, a); break;
    case 58: hipLaunchKernelGGL((k_fft1v<14, true, false>), g, t, 0, st, a); break;
    case 59: hipLaunchKernelGGL((k_fft1v<14, true, true>), g, t, 0, st, a); break;
    case 52: hipLaunchKernelGGL((k_fft1v<13, false, false>), g, t, 0, st, a); break;
    case 53: hipLaunchKernelGGL((k_fft1v<13, false, true>), g, t, 0, st, a); break;
    case 54: hipLaunchKernelGGL((k_fft1v<13, true, false>), g, t, 0, st, a); break;
    case 55: hipLaunchKernelGGL((k_fft1v<13, true, true>), g, t, 0, st, a); break;
    case 48: hipLaunchKernelGGL((k_fft1v<12, false, false>), g, t, 0, st, a); break;
    case 49: hipLaunchKernelGGL((k_fft1v<12, false, true>), g, t, 0, st, a); break;
    case 50: hipLaunchKernelGGL((k_fft1v<12, true, false>), g, t, 0, st, a); break;
    case 51: hipLaunchKernelGGL((k_fft1v<12, true, true>), g, t, 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_timf2_strong(int log2n, const Timf2Args &a0, int batch, hipStream_t st)
{
  Timf2Args a = a0; a.batch = batch; a.ss_ring = nullptr;
  // the direct sum first (it takes the launch when at most LRH_SD_KMAX bins are routed strong; LRH_SD=0: the transform kernel always)
  const char *sde_ = getenv("LRH_SD"); const int sd_ = sde_ ? atoi(sde_) : 1;      // (read per launch: tests/test_gpu_fused.py flips it)
  a.sd_kmax = sd_ ? LRH_SD_KMAX : 0;
  if (a.sd_kmax) { const hipError_t e = launch_timf2_sd(log2n, a, st); if (e != hipSuccess) return e; }
  switch (log2n) {
    case 14: hipLaunchKernelGGL((k_timf2<14, 1, false, true>), dim3(fftl_grid<14>(batch, a.spare_cus)), dim3(fft1_threads(14)), 0, st, a); break;
    case 13: hipLaunchKernelGGL((k_timf2<13, 1, false, true>), dim3(fftl_grid<13>(batch, a.spare_cus)), dim3(fft1_threads(13)), 0, st, a); break;
    case 12: hipLaunchKernelGGL((k_timf2<12, 1, false, true>), dim3(fftl_grid<12>(batch, a.spare_cus)), dim3(fft1_threads(12)), 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_sumsq_join(const SumsqArgs &a, const float *part, int run, hipStream_t st)
{
  hipLaunchKernelGGL(k_sumsq_join, dim3((a.n + 255) / 256, (a.batch + run - 1) / run + 1), dim3(256), 0, st, a, part, run);
  return hipGetLastError();
}
hipError_t launch_fft2(int log2n, const Fft2Args &a0, int batch, hipStream_t st)
{
  Fft2Args a = a0; a.batch = batch;
  // consecutive transforms per workgroup: one run per resident workgroup slot (tables and prologue amortised, a
  // fraction (run+1)/(2 run) of the overlapped reads left), single transforms when the batch is small
  {
    int lds = 0, threads = 0;
    switch (log2n) {
#define LRH_FFT2_GEOM(L) case L: lds = 8 * BlockFftL<L, points_per_thread(L), 1>::LDS_CELLS; threads = fft_threads(L); break;
      LRH_FFT2_GEOM(6) LRH_FFT2_GEOM(7) LRH_FFT2_GEOM(8) LRH_FFT2_GEOM(9) LRH_FFT2_GEOM(10) LRH_FFT2_GEOM(11)
      LRH_FFT2_GEOM(12) LRH_FFT2_GEOM(13) LRH_FFT2_GEOM(14)
#undef LRH_FFT2_GEOM
      default: return hipErrorInvalidValue;
    }
    const int cap = persistent_grid(lds, threads, 1 << 30);
    int run = a0.run > 0 ? a0.run : (batch + cap - 1) / cap;     // a0.run: tuning knob LRH_FFT2_RUN (read by lrh_open)
    a.run = run < 1 ? 1 : (run > 16 ? 16 : run);
  }
  LRH_DISPATCH(LRH_LAUNCH_FFT2, log2n, 6, 14, a, batch, st);
  return hipGetLastError();
}
template <int LA, int LB> static void launch_fft2_big_t(const Fft2BigArgs &a0, int batch, hipStream_t st, int steps)   // steps: 1 column step, 2 row step, 3 both
{
  Fft2BigArgs a = a0; a.batch = batch;
  {
    // transforms per workgroup: enough workgroups left to fill the chip a few times over
    const int tiles = (1 << LB) / LRH_TILE;
    int run = a0.run > 0 ? a0.run : batch * tiles / 512;         // a0.run: tuning knob LRH_FFT2_COLS_RUN (read by lrh_open)
    a.run = run < 1 ? 1 : (run > 32 ? 32 : run);
  }
  // columns of 256 points: 16 points per thread (LRH_FFT2_COLS_P16=0: the 4-point form, for comparison)
  static const int cols16 = getenv("LRH_FFT2_COLS_P16") ? atoi(getenv("LRH_FFT2_COLS_P16")) : 2;   // 2: 32 columns per workgroup (256-byte pieces of a row), 1: 16 columns
  static const int cols16_wgs = getenv("LRH_FFT2_COLS_WGS") ? atoi(getenv("LRH_FFT2_COLS_WGS")) : 1024;    // workgroups aimed at (four of 256 threads fit a CU)
  if ((steps & 1) && LA == 8 && cols16) {
    const int tile = cols16 == 2 ? 32 : 16, tiles = (1 << LB) / tile;
    int run = a0.run > 0 ? a0.run : batch * tiles / (cols16 == 2 ? cols16_wgs / 2 : cols16_wgs);
    a.run = run < 1 ? 1 : (run > 32 ? 32 : run);
    if (cols16 == 2) hipLaunchKernelGGL((k_fft2_cols16<LB, 32>), dim3(tiles, (batch + a.run - 1) / a.run), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((k_fft2_cols16<LB, 16>), dim3(tiles, (batch + a.run - 1) / a.run), dim3(256), 0, st, a);
  }
  else
  if (steps & 1) hipLaunchKernelGGL((k_fft2_cols<LA, LB>), dim3((1 << LB) / LRH_TILE, (batch + a.run - 1) / a.run), dim3(LRH_TILE * ((1 << LA) / sub_ppt(LA))), 0, st, a);
  if (!(steps & 2)) return;
  // rows of 256 points: 16 points per thread (k_fft2_rows' PPT; LRH_FFT2_ROWS_P16=0: the 4-point form, for comparison)
  static const int rows16 = getenv("LRH_FFT2_ROWS_P16") ? atoi(getenv("LRH_FFT2_ROWS_P16")) : 2;   // 2: k_fft2_rows16x, 1: k_fft2_rows<.., 16>, 0: the 4-point form
  if constexpr (LB == 8) {
    if (rows16 == 2) {                                   // 32 neighbouring bins per workgroup, one exchange, loads of the next transform behind it
      if (a.ps_avgnum > 0)
        hipLaunchKernelGGL((k_fft2_rows16x<LA, true>), dim3((1 << LA) / 32, (a.ps_counter + batch + a.ps_avgnum - 1) / a.ps_avgnum), dim3(512), 0, st, a);
      else
        hipLaunchKernelGGL((k_fft2_rows16x<LA, false>), dim3((1 << LA) / 32, batch), dim3(512), 0, st, a);
      return;
    }
    if (rows16) {
      if (a.ps_avgnum > 0)
        hipLaunchKernelGGL((k_fft2_rows<LA, LB, true, 16>), dim3((1 << LA) / LRH_TILE, (a.ps_counter + batch + a.ps_avgnum - 1) / a.ps_avgnum),
                           dim3(LRH_TILE * ((1 << LB) / 16)), 0, st, a);
      else
        hipLaunchKernelGGL((k_fft2_rows<LA, LB, false, 16>), dim3((1 << LA) / LRH_TILE, batch), dim3(LRH_TILE * ((1 << LB) / 16)), 0, st, a);
      return;
    }
  }
  if (a.ps_avgnum > 0)
    hipLaunchKernelGGL((k_fft2_rows<LA, LB, true, sub_ppt(LB)>), dim3((1 << LA) / LRH_TILE, (a.ps_counter + batch + a.ps_avgnum - 1) / a.ps_avgnum),
                       dim3(LRH_TILE * ((1 << LB) / sub_ppt(LB))), 0, st, a);
  else
    hipLaunchKernelGGL((k_fft2_rows<LA, LB, false, sub_ppt(LB)>), dim3((1 << LA) / LRH_TILE, batch), dim3(LRH_TILE * ((1 << LB) / sub_ppt(LB))), 0, st, a);
}
// ---- fft1 / timf2 for fft1_size 32768: four-step through an HBM scratch (same tiling as k_fft2_cols / k_fft2_rows) -------------
// fft1: x[n] = (I w, -Q w)[n], n = NB n1 + n2; X[k1 + NA k2] = sum_n2 [ w_N^(n2 k1) sum_n1 x[NB n1 + n2] w_NA^(n1 k1) ] w_NB^(n2 k2), e^{+j}.
// TILE columns per workgroup: 16 (4 points per thread) or, for int16 samples, 32 with 8 points per thread -- 32 adjacent
// short2 are a whole 128-byte line, 16 are half of one (the other half is fetched again by the neighbouring workgroup)
template <int LA, int LB, bool DW, int TILE>
__global__ __launch_bounds__(1024, 4) void k_fft1_cols(Fft1BigArgs g)
{
  using Raw = typename std::conditional<DW, int2, short2>::type;
  const Fft1Args &a = g.f;
  constexpr int P = (TILE << LA) / 1024;
  using Plan = FftPlan<LA, P>;
  constexpr int NA = 1 << LA, NB = 1 << LB, T = Plan::T, R0 = Plan::R0, RL = Plan::RL;
  constexpr int CS = Plan::LDS_CELLS + 1;
  static_assert(TILE * T == 1024, "1024 threads");
  __shared__ float2 lds[TILE * CS];
  // a workgroup takes g.run consecutive blocks of its columns: window values and twiddles of its points do not depend on the block and
  // are fetched once (they were two of the three loads per point), the next block's samples travel while the current one is transformed
  const int c = threadIdx.x & (TILE - 1), l = threadIdx.x / TILE;
  const int n2 = blockIdx.x * TILE + c;
  const int b0 = blockIdx.y * g.run, b1 = min(b0 + g.run, a.batch);
  float wv[P]; float2 tw[P];
#pragma unroll
  for (int m = 0; m < P / R0; m++)
#pragma unroll
    for (int s = 0; s < R0; s++) wv[m * R0 + s] = a.window[NB * ((l + m * T) + s * (NA / R0)) + n2];
#pragma unroll
  for (int m = 0; m < P / RL; m++)
#pragma unroll
    for (int q = 0; q < RL; q++) { const float2 w = g.tw_big[(n2 * ((l + m * T) + q * (NA / RL))) & (NA * NB - 1)]; tw[m * RL + q] = make_float2(w.x, -w.y); }
  Raw raw[P];
  auto fetch = [&](int b) {
    const int p0 = a.p0_first + b * a.step;
#pragma unroll
    for (int m = 0; m < P / R0; m++)
#pragma unroll
      for (int s = 0; s < R0; s++) {
        const int i = NB * ((l + m * T) + s * (NA / R0)) + n2;
        raw[m * R0 + s] = ((const Raw *)a.timf1)[((p0 + i) * a.chan_count + a.chan_index) & a.ring_mask];
      }
  };
  if (b0 < b1) fetch(b0);
  for (int b = b0; b < b1; b++) {
    float2 x[P];
#pragma unroll
    for (int j = 0; j < P; j++) x[j] = make_float2((float)raw[j].x * wv[j], -((float)raw[j].y * wv[j]));   // Q negated before the e^{+j} transform (fft1.c:432-447)
    if (b + 1 < b1) fetch(b + 1);
    float2 *col = lds + c * CS;
    BlockFft<LA, P, +1>::run(x, col, g.tw_a, l);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < P / RL; m++)
#pragma unroll
      for (int q = 0; q < RL; q++) col[(l + m * T) + q * (NA / RL)] = cmul(x[m * RL + q], tw[m * RL + q]);
    __syncthreads();
    float2 *sc = g.scratch + (size_t)b * NA * NB + (size_t)blockIdx.x * TILE * NA;
    for (int e = threadIdx.x; e < TILE * NA; e += TILE * T) {
      const int cc = e / NA, k1 = e - cc * NA;
      store_stream(&sc[(size_t)cc * NA + k1], lds[cc * CS + k1]);
    }
    __syncthreads();
  }
}
template <int LA, int LB>
__global__ __launch_bounds__(1024) void k_fft1_rows(Fft1BigArgs g)
{
  const Fft1Args &a = g.f;
  constexpr int P = sub_ppt(LB);
  using Plan = FftPlan<LB, P>;
  constexpr int NA = 1 << LA, NB = 1 << LB, N = NA * NB, T = Plan::T, R0 = Plan::R0, RL = Plan::RL;
  constexpr int CS = Plan::LDS_CELLS + 1;
  __shared__ float2 lds[LRH_TILE * CS];
  const int c = threadIdx.x & (LRH_TILE - 1), l = threadIdx.x >> 4;
  const int k1 = blockIdx.x * LRH_TILE + c, b = blockIdx.y;
  const float2 *sc = g.scratch + (size_t)b * NA * NB;
  float2 x[P];
#pragma unroll
  for (int m = 0; m < P / R0; m++)
#pragma unroll
    for (int s = 0; s < R0; s++) x[m * R0 + s] = sc[(size_t)((l + m * T) + s * (NB / R0)) * NA + k1];
  BlockFft<LB, P, +1>::run(x, lds + c * CS, g.tw_b, l);
  float2 *out = a.out + (size_t)((a.first_nb + b) & a.nb_mask) * N;
#pragma unroll
  for (int m = 0; m < P / RL; m++)
#pragma unroll
    for (int q = 0; q < RL; q++) {
      const int k = k1 + NA * ((l + m * T) + q * (NB / RL));
      int kk = (k + N / 2) & (N - 1);                     // DC at N/2 (make_permute mode 1, fft0.c:1196-1204)
      float2 v = x[m * RL + q];
      if (a.direction < 0) { kk = (N - kk) & (N - 1); v = make_float2(v.y, v.x); }   // fft1.c:3660-3679
      store_stream(&out[kk], cmul(v, a.filtercorr[kk]));
    }
}
// timf2, sin^2 overlap (see k_timf2): out_t[n] = ampfac * DFT_{e^-j}( S_t + (-1)^k S_{t-1} )[n], n < N/2, per stream.
// k = NB i1 + i2 in, n = o1 + NA o2 out; blockIdx.z = stream (0 weak, 1 strong); routing bits dense, one per bin.
template <int LA, int LB>
__global__ __launch_bounds__(1024, 4) void k_timf2_cols(Timf2BigArgs g)
{
  const Timf2Args &a = g.t;
  constexpr int P = sub_ppt(LA);
  using Plan = FftPlan<LA, P>;
  constexpr int NA = 1 << LA, NB = 1 << LB, N = NA * NB, T = Plan::T, R0 = Plan::R0, RL = Plan::RL;
  constexpr int CS = Plan::LDS_CELLS + 1;
  __shared__ float2 lds[LRH_TILE * CS];
  const int c = threadIdx.x & (LRH_TILE - 1), l = threadIdx.x >> 4;
  const int i2 = blockIdx.x * LRH_TILE + c, b = blockIdx.y, st = blockIdx.z;
  const float2 *cur = a.spec + (size_t)((a.first_nb + b) & a.nb_mask) * N;
  const float2 *prv = a.spec + (size_t)((a.first_nb + b - 1) & a.nb_mask) * N;
  const unsigned int *pk_prev = b == 0 ? a.pack_prev : a.pack_cur;
  const float sg = (i2 & 1) ? -1.f : 1.f;                // (-1)^k, k = NB i1 + i2, NB even
  float2 x[P];
#pragma unroll
  for (int m = 0; m < P / R0; m++)
#pragma unroll
    for (int s = 0; s < R0; s++) {
      const int k = NB * ((l + m * T) + s * (NA / R0)) + i2;
      const unsigned int wc = (a.pack_cur[k >> 5] >> (k & 31)) & 1u, wp = (pk_prev[k >> 5] >> (k & 31)) & 1u;   // 1: weak (liminfo == 0, timf2.c:50)
      float2 v = make_float2(0.f, 0.f), pv = make_float2(0.f, 0.f);
      if (wc != (unsigned int)st) v = cur[k];
      if (wp != (unsigned int)st) pv = prv[k];
      x[m * R0 + s] = make_float2(v.x + sg * pv.x, v.y + sg * pv.y);
    }
  float2 *col = lds + c * CS;
  BlockFft<LA, P, -1>::run(x, col, g.tw_a, l);
  __syncthreads();
#pragma unroll
  for (int m = 0; m < P / RL; m++)
#pragma unroll
    for (int q = 0; q < RL; q++) {
      const int o1 = (l + m * T) + q * (NA / RL);
      col[o1] = cmul(x[m * RL + q], g.tw_big[(i2 * o1) & (N - 1)]);
    }
  __syncthreads();
  float2 *sc = g.scratch + ((size_t)b * 2 + st) * NA * NB + (size_t)blockIdx.x * LRH_TILE * NA;
  for (int e = threadIdx.x; e < LRH_TILE * NA; e += LRH_TILE * T) {
    const int cc = e / NA, o1 = e - cc * NA;
    sc[(size_t)cc * NA + o1] = lds[cc * CS + o1];
  }
}
template <int LA, int LB, int PPT>
__global__ __launch_bounds__(LRH_TILE * ((1 << LB) / PPT)) void k_timf2_rows(Timf2BigArgs g)     // PPT: see k_fft2_rows
{
  const Timf2Args &a = g.t;
  constexpr int P = PPT;
  using Plan = FftPlan<LB, P>;
  constexpr int NA = 1 << LA, NB = 1 << LB, T = Plan::T, R0 = Plan::R0, RL = Plan::RL;
  constexpr int CS = Plan::LDS_CELLS + 1;
  __shared__ float2 lds[LRH_TILE * CS];
  const int c = threadIdx.x & (LRH_TILE - 1), l = threadIdx.x >> 4;
  const int o1 = blockIdx.x * LRH_TILE + c, b = blockIdx.y, st = blockIdx.z;
  const float2 *sc = g.scratch + ((size_t)b * 2 + st) * NA * NB;
  float2 x[P];
#pragma unroll
  for (int m = 0; m < P / R0; m++)
#pragma unroll
    for (int s = 0; s < R0; s++) x[m * R0 + s] = sc[(size_t)((l + m * T) + s * (NB / R0)) * NA + o1];
  BlockFft<LB, P, -1>::run(x, lds + c * CS, g.tw_b, l);
  const int pa = a.pa_first + b * a.step;
#pragma unroll
  for (int m = 0; m < P / RL; m++)
#pragma unroll
    for (int q = 0; q < RL; q++) {
      const int o2 = (l + m * T) + q * (NB / RL);
      if (o2 >= NB / 2) continue;                          // first half of the block only (n < N/2)
      const int r = (pa + o1 + NA * o2) & a.mask;
      const float2 v = x[m * RL + q];
      const float2 o = make_float2(a.ampfac * v.x, a.ampfac * v.y);
      if (st == 0) { store_stream(&a.timf2w[r], o); __builtin_nontemporal_store(o.x * o.x + o.y * o.y, &a.pwr[r]); }   // weak power only (timf2.c:1010-1012)
      else store_stream(&a.timf2s[r], o);
    }
}
template <int LA, int LB> static void launch_timf2_rows(const Timf2BigArgs &a, int batch, hipStream_t st)
{
  static const int p16 = getenv("LRH_TIMF2_ROWS_P16") ? atoi(getenv("LRH_TIMF2_ROWS_P16")) : 1;   // 16 points per thread (0: sub_ppt, for comparison)
  if (p16) hipLaunchKernelGGL((k_timf2_rows<LA, LB, 16>), dim3((1 << LA) / LRH_TILE, batch, 2), dim3(LRH_TILE * ((1 << LB) / 16)), 0, st, a);
  else hipLaunchKernelGGL((k_timf2_rows<LA, LB, sub_ppt(LB)>), dim3((1 << LA) / LRH_TILE, batch, 2), dim3(LRH_TILE * ((1 << LB) / sub_ppt(LB))), 0, st, a);
}
template <int LA, int LB> static void launch_fft1_big_t(const Fft1BigArgs &a, int batch, hipStream_t st, int steps)
{
  const dim3 gc((1 << LB) / LRH_TILE, batch), gr((1 << LA) / LRH_TILE, batch);
  if (steps & 1) {
    Fft1BigArgs c = a; c.f.batch = batch;
    const int tiles = (1 << LB) / (a.f.dword ? LRH_TILE : 32);
    int run = batch * tiles / 1024; if (run < 1) run = 1; if (run > 16) run = 16;      // ~4 workgroups per CU over the launch
    c.run = run;
    const dim3 gcr(tiles, (batch + run - 1) / run);
    if (a.f.dword) hipLaunchKernelGGL((k_fft1_cols<LA, LB, true, LRH_TILE>), gcr, dim3(1024), 0, st, c);
    else hipLaunchKernelGGL((k_fft1_cols<LA, LB, false, 32>), gcr, dim3(1024), 0, st, c);
  }
  if (steps & 2) hipLaunchKernelGGL((k_fft1_rows<LA, LB>), gr, dim3(LRH_TILE * ((1 << LB) / sub_ppt(LB))), 0, st, a);
}
hipError_t launch_fft1_big(int log2n, const Fft1BigArgs &a, int batch, hipStream_t st, int steps)
{
  if ((log2n != 15 && log2n != 16) || a.f.real || a.f.shift_i || a.f.shift_q) return hipErrorInvalidValue;
  if (log2n == 15) launch_fft1_big_t<8, 7>(a, batch, st, steps);
  else launch_fft1_big_t<8, 8>(a, batch, st, steps);      // 65536: the reference's maximum without the second fft (fft0.c:1162-1169)
  return hipGetLastError();
}

// ---- fft1_size 32768: row step of fft1 + fft1_c's sums + column step of timf2 (both streams) -----------------------------------------
// The row step of fft1 leaves X[k1 + NA k2] for fixed k1; timf2's column step wants S[NB i1 + i2] for fixed i2.  With DC moved to N/2
// (a multiple of NB) bin k1 + NA k2 has i2 = k1 mod NB, i1 = 2 k2 + (k1 >= NB) (+ NA/2): the two rows k1 = i2 and i2 + NB together ARE
// column i2.  A workgroup therefore takes sixteen columns i2 = thirty-two rows, transforms the rows (4 points per thread), applies the
// filter correction, writes the spectrum (once: nothing reads it back on this path), adds |X|^2 into the running sums of the averaging
// period, turns the tile through LDS and runs the sixteen 256-point column transforms of the weak and then of the strong stream on
// S_t + (-1)^k S_{t-1} with the predecessor's bins kept in registers.  It walks a run of whole averaging periods (so the sums need no
// joining) and starts each run by transforming the block before it once more for the predecessor (the launch's first block takes it
// from the ring).  Saved against the separate kernels per transform of 32768 points: the spectrum read three times (k_sumsq, cur and
// prev of k_timf2_cols), 786 KB of 3.1 MB.
template <int LA, int LB>
__global__ __launch_bounds__(1024, 4) void k_fft1r_t2c(Fft1rT2cArgs g)
{
  const Fft1Args &f = g.f1.f; const Timf2Args &t = g.t2.t; const SumsqArgs &ss = g.ss;
  constexpr int PR = sub_ppt(LB), PC = sub_ppt(LA);
  using PlanR = FftPlan<LB, PR>; using PlanC = FftPlan<LA, PC>;
  constexpr int NA = 1 << LA, NB = 1 << LB, N = NA * NB;
  constexpr int TR = PlanR::T, TC = PlanC::T, CSR = PlanR::LDS_CELLS + 1, CSC = PlanC::LDS_CELLS + 1;
  constexpr int ROWS = 2 * LRH_TILE, TS = NA + 1;
  static_assert(ROWS * TR == 1024 && LRH_TILE * TC == 1024 && TC == 64, "1024 threads: 32 rows x 32, 16 columns x one wave");
  constexpr int XCH = ROWS * CSR > LRH_TILE * CSC ? ROWS * CSR : LRH_TILE * CSC;
  __shared__ float2 xch[XCH];                             // exchange buffer of the row transforms, then of the column transforms
  __shared__ float2 tile[LRH_TILE * TS];                  // the spectrum tile on its way from rows to columns: [column][i1]
  const int i2b = blockIdx.x * LRH_TILE;
  // rows: thread (rr, lr) -- rr & 15 = column of the tile, rr >> 4 = which of its two rows; columns: thread (cc, lc).
  // The coordinates are re-derived from an opaque copy of the thread index in every trip of the block loop: otherwise LICM hoists
  // the address arithmetic of every unrolled load and store out of it and the kernel spills (see k_timf2)
  int rr, lr, k1, cc, lc, i2;
  auto coords = [&]() {
    int t_ = threadIdx.x; asm volatile("" : "+v"(t_));
    rr = t_ & (ROWS - 1); lr = t_ / ROWS; k1 = i2b + (rr & (LRH_TILE - 1)) + (rr / LRH_TILE) * NB;
    cc = t_ / TC; lc = t_ & (TC - 1); i2 = i2b + cc;          // a column = one wave: its transform needs no workgroup barrier
  };
  coords();
  const float sg = ((threadIdx.x / TC) & 1) ? -1.f : 1.f;   // (-1)^k, k = NB i1 + i2 (i2b is even)
  // the run: groups of the averaging periods like k_sumsq (group 0 continues a period begun by an earlier launch when c0 > 0)
  const int ngroups = (ss.c0 + ss.batch + ss.avg - 1) / ss.avg;
  const int g_first = blockIdx.y * g.groups_per_run, g_end = min(g_first + g.groups_per_run, ngroups);
  if (g_first >= g_end) return;
  float2 pv[PC];                                          // the predecessor's bins of this thread's column points
  // what does not change from block to block is fetched once per run: filter correction and spectrum places of this thread's row
  // bins, routing bits and twiddles of its column points -- inside the loop they were ~3.5 us of exposed global latency per block
  float2 fc[PR];
  auto kk_of = [&](int j) -> int {                        // spectrum place of row output j: DC at N/2 (make_permute mode 1, fft0.c:1196-1204)
    return (k1 + NA * ((lr + (j / PlanR::RL) * TR) + (j % PlanR::RL) * (NB / PlanR::RL)) + N / 2) & (N - 1);
  };
#pragma unroll
  for (int j = 0; j < PR; j++) fc[j] = f.filtercorr[kk_of(j)];
  unsigned int wcm = 0, wpm0 = 0;                         // bit j: point j is weak in the current table / in the table before this launch (timf2.c:50)
#pragma unroll
  for (int m = 0; m < PC / PlanC::R0; m++)
#pragma unroll
    for (int s = 0; s < PlanC::R0; s++) {
      const int j = m * PlanC::R0 + s, k = NB * ((lc + m * TC) + s * (NA / PlanC::R0)) + i2;
      wcm |= ((t.pack_cur[k >> 5] >> (k & 31)) & 1u) << j; wpm0 |= ((t.pack_prev[k >> 5] >> (k & 31)) & 1u) << j;
    }
  float2 twc[PC];
#pragma unroll
  for (int m = 0; m < PC / PlanC::RL; m++)
#pragma unroll
    for (int q = 0; q < PlanC::RL; q++) twc[m * PlanC::RL + q] = g.t2.tw_big[(i2 * ((lc + m * TC) + q * (NA / PlanC::RL))) & (N - 1)];
  // row inputs of a block
  auto fetch = [&](int b, float2 (&x)[PR]) {
    const float2 *sc = g.f1.scratch + (size_t)b * N;
#pragma unroll
    for (int m = 0; m < PR / PlanR::R0; m++)
#pragma unroll
      for (int s = 0; s < PlanR::R0; s++) x[m * PlanR::R0 + s] = sc[(size_t)((lr + m * TR) + s * (NB / PlanR::R0)) * NA + k1];
  };
  // the row step on x: this thread's bins of the spectrum in v[]
  auto rows = [&](float2 (&x)[PR]) {                        // in place (the register file is what limits this kernel)
    BlockFft<LB, PR, +1>::run(x, xch + rr * CSR, g.f1.tw_b, lr);
#pragma unroll
    for (int j = 0; j < PR; j++) x[j] = cmul(x[j], fc[j]);
  };
  auto to_tile = [&](const float2 (&v)[PR]) {
    __syncthreads();                                      // the tile's previous readers are through
#pragma unroll
    for (int j = 0; j < PR; j++) { const int kk = kk_of(j); tile[(kk & (NB - 1)) % LRH_TILE * TS + (kk >> LB)] = v[j]; }
    __syncthreads();
  };
  const int b_first = g_first == 0 ? 0 : g_first * ss.avg - ss.c0;
  int b_last = g_end * ss.avg - ss.c0; if (b_last > ss.batch) b_last = ss.batch;          // one past the run's last block
  float2 xn[PR];
  if (b_first > 0) {                                      // the block before the run, for its spectrum only
    float2 x[PR];
    fetch(b_first - 1, x);
    fetch(b_first, xn);
    rows(x);
    to_tile(x);
#pragma unroll
    for (int m = 0; m < PC / PlanC::R0; m++)
#pragma unroll
      for (int s = 0; s < PlanC::R0; s++) pv[m * PlanC::R0 + s] = tile[cc * TS + (lc + m * TC) + s * (NA / PlanC::R0)];
  } else {
    fetch(0, xn);
    const float2 *prv = t.spec + (size_t)((t.first_nb - 1) & t.nb_mask) * N;
#pragma unroll
    for (int m = 0; m < PC / PlanC::R0; m++)
#pragma unroll
      for (int s = 0; s < PlanC::R0; s++) pv[m * PlanC::R0 + s] = prv[NB * ((lc + m * TC) + s * (NA / PlanC::R0)) + i2];
  }
  for (int gi = g_first; gi < g_end; gi++) {
    const int start = gi == 0 ? 0 : gi * ss.avg - ss.c0;
    int count = ss.avg - (gi == 0 ? ss.c0 : 0);
    if (count > ss.batch - start) count = ss.batch - start;
    const bool accumulate = gi == 0 && ss.c0 > 0;
    float *dst = ss.sumsq + ((ss.pa0 + gi * N) & ss.sumsq_mask);
    float acc[PR];
    for (int bi = 0; bi < count; bi++) {
      const int b = start + bi;
      coords();
      {
        float2 x[PR];
#pragma unroll
        for (int j = 0; j < PR; j++) x[j] = xn[j];
        if (b + 1 < b_last) fetch(b + 1, xn);            // the next block's inputs travel while this one is worked on
        rows(x);
        float2 *out = f.out + (size_t)((f.first_nb + b) & f.nb_mask) * N;
        const bool keep = g.keep_spec || b == ss.batch - 1;
#pragma unroll
        for (int j = 0; j < PR; j++) {
          const int kk = kk_of(j);
          if (keep) store_stream(&out[kk], x[j]);
          const float p2 = x[j].x * x[j].x + x[j].y * x[j].y;
          if (bi == 0) acc[j] = accumulate ? dst[kk] + p2 : p2; else acc[j] += p2;   // "=" then "+=" (fft1.c:4126-4169)
        }
        to_tile(x);
      }
      // columns: S_t + (-1)^k S_{t-1}, each masked with the routing table that was in force for it
      const unsigned int wpm = b == 0 ? wpm0 : wcm;
#pragma unroll 1
      for (int st = 0; st < 2; st++) {
        float2 x2[PC];
#pragma unroll
        for (int m = 0; m < PC / PlanC::R0; m++)
#pragma unroll
          for (int s = 0; s < PlanC::R0; s++) {
            const int j = m * PlanC::R0 + s;
            const float2 cur = tile[cc * TS + (lc + m * TC) + s * (NA / PlanC::R0)];
            const bool tc = ((wcm >> j) & 1u) != (unsigned int)st, tp = ((wpm >> j) & 1u) != (unsigned int)st;
            x2[j] = make_float2((tc ? cur.x : 0.f) + sg * (tp ? pv[j].x : 0.f), (tc ? cur.y : 0.f) + sg * (tp ? pv[j].y : 0.f));
          }
        // the wave's own column: transform, twiddle, and out to the scratch straight from the registers (a register's outputs
        // o1 = lc + const are 512 bytes of consecutive lanes) -- no LDS hand-over, no workgroup barrier
        BlockFft<LA, PC, -1, true>::run(x2, xch + cc * CSC, g.t2.tw_a, lc);
        float2 *sc2 = g.t2.scratch + ((size_t)b * 2 + st) * N + (size_t)i2 * NA;
#pragma unroll
        for (int m = 0; m < PC / PlanC::RL; m++)
#pragma unroll
          for (int q = 0; q < PlanC::RL; q++) store_stream(&sc2[(lc + m * TC) + q * (NA / PlanC::RL)], cmul(x2[m * PlanC::RL + q], twc[m * PlanC::RL + q]));
        BlockFft<LA, PC, -1, true>::sync();               // the exchange buffer is free again (the wave's next transform)
      }
      __syncthreads();                                    // (the row step of the next block reuses the exchange buffer of all waves)
#pragma unroll
      for (int m = 0; m < PC / PlanC::R0; m++)
#pragma unroll
        for (int s = 0; s < PlanC::R0; s++) pv[m * PlanC::R0 + s] = tile[cc * TS + (lc + m * TC) + s * (NA / PlanC::R0)];
    }
    if (count > 0) {
#pragma unroll
      for (int j = 0; j < PR; j++) dst[kk_of(j)] = acc[j];
    }
  }
}
hipError_t launch_fft1r_t2c(const Fft1rT2cArgs &a0, int batch, hipStream_t st)
{
  constexpr int LA = 8, LB = 7;
  Fft1rT2cArgs a = a0; a.t2.t.batch = batch;
  const int ngroups = (a.ss.c0 + a.ss.batch + a.ss.avg - 1) / a.ss.avg;
  const int tiles = (1 << LB) / LRH_TILE;
  // 120 VGPRs x 1024 threads: one workgroup per CU is resident, so the launch is one wave of workgroups, one per CU when the batch
  // is long enough -- any other count leaves a partly filled last wave (656 workgroups ran as 3 waves at 85 %)
  int cus = 256; { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
  const int runs_per_tile = cus / tiles > 0 ? cus / tiles : 1;
  int gpr = (ngroups + runs_per_tile - 1) / runs_per_tile; // a run costs one extra row step
  if (gpr < 1) gpr = 1;
  a.groups_per_run = gpr;
  hipLaunchKernelGGL((k_fft1r_t2c<LA, LB>), dim3(tiles, (ngroups + gpr - 1) / gpr), dim3(1024), 0, st, a);
  launch_timf2_rows<LA, LB>(a.t2, batch, st);
  return hipGetLastError();
}
hipError_t launch_timf2_big(int log2n, const Timf2BigArgs &a0, int batch, hipStream_t st)
{
  if (log2n != 15 || a0.t.mode != 1) return hipErrorInvalidValue;
  constexpr int LA = 8, LB = 7;
  Timf2BigArgs a = a0; a.t.batch = batch;
  hipLaunchKernelGGL((k_timf2_cols<LA, LB>), dim3((1 << LB) / LRH_TILE, batch, 2), dim3(LRH_TILE * ((1 << LA) / sub_ppt(LA))), 0, st, a);
  launch_timf2_rows<LA, LB>(a, batch, st);
  return hipGetLastError();
}

hipError_t launch_fft2_big(int log2n, const Fft2BigArgs &a, int batch, hipStream_t st, int steps)
{
  switch (log2n) {
    case 15: launch_fft2_big_t<8, 7>(a, batch, st, steps); break;
    case 16: launch_fft2_big_t<8, 8>(a, batch, st, steps); break;
    case 17: launch_fft2_big_t<9, 8>(a, batch, st, steps); break;
    case 18: launch_fft2_big_t<9, 9>(a, batch, st, steps); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_mix1_back(int log2n, const Mix1Args &a, int batch, hipStream_t st)
{
  LRH_DISPATCH(LRH_LAUNCH_MIX1, log2n, 3, 14, a, batch, st);
  return hipGetLastError();
}
#define LRH_LAUNCH_FFT3(L, a, batch, st) hipLaunchKernelGGL((k_fft3<L>), dim3(batch), dim3(fft_threads(L)), 0, st, a)
#define LRH_LAUNCH_MIX2(L, a, batch, st) hipLaunchKernelGGL((k_mix2_back<L>), dim3(batch), dim3(fft_threads(L)), 0, st, a)
hipError_t launch_fft3(int log2n, const Fft3Args &a, int batch, hipStream_t st)
{
  LRH_DISPATCH(LRH_LAUNCH_FFT3, log2n, 6, 14, a, batch, st);
  return hipGetLastError();
}
hipError_t launch_mix2_back(int log2n, const Mix2Args &a, int batch, hipStream_t st)
{
  LRH_DISPATCH(LRH_LAUNCH_MIX2, log2n, 3, 14, a, batch, st);
  return hipGetLastError();
}
hipError_t launch_mix1_out(const Mix1OutArgs &a, int batch, hipStream_t st)
{
  const int half = a.xover > 0 ? a.block : (a.overlap ? a.nm / 2 : a.nm);
  hipLaunchKernelGGL(k_mix1_out, dim3((half + 255) / 256, batch), dim3(256), 0, st, a, batch);
  return hipGetLastError();
}
hipError_t launch_blockpower(const BlockpowerArgs &a, int nblocks, hipStream_t st)
{
  hipLaunchKernelGGL(k_blockpower, dim3(nblocks), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_sumsq(const SumsqArgs &a, hipStream_t st)
{
  const int ngroups = (a.c0 + a.batch + a.avg - 1) / a.avg;
  hipLaunchKernelGGL(k_sumsq, dim3((a.n + 255) / 256, ngroups), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_slowsum(const SlowsumArgs &a, hipStream_t st)
{
  hipLaunchKernelGGL(k_slowsum, dim3((a.n + 63) / 64), dim3(64), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_powersum2(const Powersum2Args &a, hipStream_t st)
{
  const int ngroups = (a.counter + a.count + a.avgnum - 1) / a.avgnum;
  hipLaunchKernelGGL(k_powersum2, dim3((a.n + 255) / 256, ngroups), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_pol(const PolArgs &a, hipStream_t st)
{
  hipLaunchKernelGGL(k_pol, dim3((a.nm + 255) / 256, a.batch), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_xypower(const XyArgs &a, hipStream_t st)
{
  const int ngroups = (a.counter + a.batch + a.avgnum - 1) / a.avgnum;
  hipLaunchKernelGGL(k_xypower, dim3((a.n + 255) / 256, ngroups), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_waterfall(const WaterfallArgs &a0, int nlines, hipStream_t st)
{
  // more lines than the ring holds: the older ones share ring rows with the newest and would race with them, the reference's
  // sequential update_wg_waterf leaves the newest; only those are converted
  WaterfallArgs a = a0;
  const int ring_lines = a.npix > 0 ? a.wf_size / a.npix : nlines;
  if (nlines > ring_lines) {
    const int skip = nlines - ring_lines;
    a.ps += (size_t)skip * a.line_stride;
    a.ptr0 = (int)(((long long)a.ptr0 - (long long)skip * a.npix) % a.wf_size); if (a.ptr0 < 0) a.ptr0 += a.wf_size;
    nlines = ring_lines;
  }
  int work = a.npix;
  if (!(a.hx == 1 || a.hp == 1) && a.hx == 0) work = (a.npix - a.hp + a.hp - 1) / a.hp + 1;
  hipLaunchKernelGGL(k_waterfall, dim3((work + 255) / 256, nlines), dim3(256), 0, st, a);
  return hipGetLastError();
}
// exchange buffer <-> ring span of the coupled two-channel blanker: x[q-1] = ring[(pbeg+q) & mask], q = 1..count
__global__ __launch_bounds__(256) void k_span_copy(float *x, float *ring, int pbeg, int count, int mask, int to_ring)
{
  // four independent elements per thread, a wave's four accesses each a whole 256-byte row
  const int i0 = blockIdx.x * 1024 + threadIdx.x;
  float v[4];
#pragma unroll
  for (int u = 0; u < 4; u++) { const int i = i0 + 256 * u; if (i < count) v[u] = to_ring ? x[i] : ring[(pbeg + 1 + i) & mask]; }
#pragma unroll
  for (int u = 0; u < 4; u++) { const int i = i0 + 256 * u; if (i < count) { if (to_ring) ring[(pbeg + 1 + i) & mask] = v[u]; else x[i] = v[u]; } }
}
hipError_t launch_span_copy(float *x, float *ring, int pbeg, int count, int mask, int to_ring, hipStream_t st)
{
  if (count > 0) hipLaunchKernelGGL(k_span_copy, dim3((count + 1023) / 1024), dim3(256), 0, st, x, ring, pbeg, count, mask, to_ring);
  return hipGetLastError();
}
// the same for complex samples, x[i] = ring[(first + i) & mask]: the weak samples the two-channel linear blanker exchanges
__global__ __launch_bounds__(256) void k_span_copy2(float2 *x, float2 *ring, int first, int count, int mask, int to_ring)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  if (to_ring) ring[(first + i) & mask] = x[i]; else x[i] = ring[(first + i) & mask];
}
hipError_t launch_span_copy2(float2 *x, float2 *ring, int first, int count, int mask, int to_ring, hipStream_t st)
{
  if (count > 0) hipLaunchKernelGGL(k_span_copy2, dim3((count + 255) / 256), dim3(256), 0, st, x, ring, first, count, mask, to_ring);
  return hipGetLastError();
}
// =====================================================================================================
// linear ("clever") blanker: first_noise_blanker's pulse search / fit / subtract (blank1.c:765-1003)
// =====================================================================================================
// The reference walks the span once, sample by sample, and every pulse it handles changes the data the walk continues on
// (subtracted samples, flags), so the ORDER of the events is part of the result.  What is parallel: finding the next sample
// above the limit (bit words, 4096 samples per step), loading the neighbourhood of a pulse, the per-sample arithmetic of the
// fit.  One wave runs the reference's control flow with all lanes in step (every branch below is wave-uniform), a pulse's
// neighbourhood (+-128 samples: refpul_size <= 256) sits in LDS, sums run in the reference's order, and contraction to fma is
// off so that the threshold decisions see the reference's roundings.
// k_clever_prep: candidate bits (power above the limit), the flag clear over exactly the span (blank1.c:768-774) and the clear of the
// undo log's bits (span and margins); phase 1 runs only after a violation, behind k_clever_restore.
constexpr int CLV_RBLOCKS_MAX = 1024;                    // blocks of the region kernels (their counts live in reg_ctl[8 ..])
__global__ __launch_bounds__(256) void k_clever_prep(CleverArgs a)
{
  if (a.phase == 1 && a.reg_ctl[1] == 0) return;
  const int lane = threadIdx.x & 63, wmask = ((a.mask + 1) >> 6) - 1;
  const int first_word = a.pbeg >> 6, nwords = ((a.pbeg & 63) + a.total + 64) >> 6;
  const float nfl = (float)a.st->clever_limit;
  if (blockIdx.x == 0 && threadIdx.x == 0) { a.st->clever_out[0] = (a.pbeg + a.total) & a.mask; a.st->clever_out[1] = 0; a.st->clever_out[2] = 0; if (a.phase == 0) { a.st->clever_out[3] = 0; a.reg_ctl[3] = 0; } }
  { const int mw = (a.bk_margin + 63) / 64 + 1;           // words of the margins either side
    for (int w = blockIdx.x * 256 + threadIdx.x; w < nwords + 2 * mw; w += gridDim.x * 256) a.logged[(first_word - mw + w) & wmask] = 0ull; }
  for (int w = (blockIdx.x * 256 + threadIdx.x) >> 6; w < nwords; w += gridDim.x * 4) {
    const int pos = (((first_word + w) << 6) + lane) & a.mask;
    const int o = (pos - a.pbeg) & a.mask;
    const bool in = o <= a.total;
    const float v = in ? a.pwr[pos] : 0.f;
    const bool hot = in && v > nfl;
    if (in) a.flag[pos] = 0;
    const unsigned long long b = __ballot(hot);
    if (lane == 0) a.cand[(first_word + w) & wmask] = b;
  }
}
// after a violation: every sample the parallel pass rewrote goes back to what the call started with
__global__ __launch_bounds__(256) void k_clever_restore(CleverArgs a)
{
  if (a.reg_ctl[1] == 0) return;
  const int n = a.reg_ctl[3];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int pos = a.bk_pos[i];
    a.pwr[pos] = a.bk_pwr[i]; a.timf2w[pos] = a.bk_tf[i];
    if (a.twochan) { a.pwr_own[pos] = a.bk_pwo[i]; a.timf2y[pos] = a.bk_ty[i]; }
  }
}

// Regions: a candidate starts a region when no candidate lies within `gap` samples before it.  gap >= 64, so only the lowest set
// bit of a word can start one.  Ordered list of the starts (offsets from pbeg): every block counts the starts in its stretch of words
// (k_clever_count), then places them behind the blocks before it (k_clever_regions).
struct CleverWords {
  const CleverArgs &a; int wmask, first_word, nwords, back;
  __device__ CleverWords(const CleverArgs &a_) : a(a_), wmask(((a_.mask + 1) >> 6) - 1), first_word(a_.pbeg >> 6), nwords(((a_.pbeg & 63) + a_.total + 64) >> 6), back((a_.gap + 63) / 64 + 1) {}
  __device__ int start_of(int w) const                    // offset of a region start in word w, or -1
  {
    const unsigned long long v = a.cand[(first_word + w) & wmask];
    if (!v) return -1;
    const int pos = ((first_word + w) << 6) + __ffsll((long long)v) - 1;
    for (int k = 1; k <= back && w - k >= 0; k++) {
      const unsigned long long u = a.cand[(first_word + w - k) & wmask];
      if (u) { const int prev = ((first_word + w - k) << 6) + 63 - __clzll((long long)u); if (pos - prev < a.gap) return -1; break; }
    }
    return pos - a.pbeg;                                  // not masked: words count up from pbeg's word, so this is the offset
  }
  // every thread takes `per` consecutive words, threads and blocks in the order of the words
  __device__ int per() const { return (nwords + (int)gridDim.x * 256 - 1) / ((int)gridDim.x * 256); }
};
__device__ __forceinline__ int clv_block_sum(int v, int *sh)        // sum over the 256 threads of a block
{
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const int tot = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return tot;
}
__global__ __launch_bounds__(256) void k_clever_count(CleverArgs a)
{
  __shared__ int sh[4];
  const CleverWords cw(a);
  const int per = cw.per(), w0 = (blockIdx.x * 256 + threadIdx.x) * per, w1 = min(w0 + per, cw.nwords);
  int n = 0;
  for (int w = w0; w < w1; w++) if (cw.start_of(w) >= 0) n++;
  const int tot = clv_block_sum(n, sh);
  if (threadIdx.x == 0) a.reg_ctl[8 + blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void k_clever_regions(CleverArgs a)
{
  __shared__ int sh[4], cnt[256];
  const CleverWords cw(a);
  const int per = cw.per(), w0 = (blockIdx.x * 256 + threadIdx.x) * per, w1 = min(w0 + per, cw.nwords);
  int before = 0;                                         // regions of the blocks before this one
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) before += a.reg_ctl[8 + b];
  before = clv_block_sum(before, sh);
  int n = 0;
  for (int w = w0; w < w1; w++) if (cw.start_of(w) >= 0) n++;
  cnt[threadIdx.x] = n;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {              // inclusive scan
    const int v = threadIdx.x >= off ? cnt[threadIdx.x - off] : 0;
    __syncthreads();
    cnt[threadIdx.x] += v;
    __syncthreads();
  }
  int at = before + cnt[threadIdx.x] - n;
  for (int w = w0; w < w1; w++) { const int o = cw.start_of(w); if (o >= 0) { if (at < a.max_regions) a.reg_start[at] = o; at++; } }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) {
    const int tot = before + cnt[255];
    a.reg_ctl[0] = min(tot, a.max_regions); a.reg_ctl[1] = (tot > a.max_regions || a.force_serial) ? 1 : 0; a.reg_ctl[2] = a.total;
  }
}

// extents must stay apart; the last region's stopping point is the call's
__global__ __launch_bounds__(1024) void k_clever_check(CleverArgs a)
{
  const int n = a.reg_ctl[0];
  int bad = 0;
  for (int r = threadIdx.x; r + 1 < n; r += 1024) if (a.reg_ext[2 * r + 1] >= a.reg_ext[2 * (r + 1)]) bad = 1;
  bad = __syncthreads_or(bad);
  if (threadIdx.x == 0) {
    if (bad) a.reg_ctl[1] = 1;
    const int violated = a.reg_ctl[1];
    if (!violated) a.st->clever_out[0] = (a.pbeg + a.reg_ctl[2]) & a.mask;
    a.st->clever_out[3] = violated;                        // the host reads this with the resume point and issues the replay if it must
  }
}

// 64 consecutive ring positions from pos0 (lane 0's): the lanes' bits laid onto the one or two 64-bit words of a per-sample bitmap they
// fall in -- one atomic per word instead of one per sample (atomics on the same word queue up behind each other in the L2)
struct ClvWordPair { int wa, wb; unsigned long long ma, mb; };
__device__ __forceinline__ ClvWordPair clv_words(int pos0, unsigned long long bits, int wmask)
{
  ClvWordPair r; const int sh = pos0 & 63;
  r.wa = (pos0 >> 6) & wmask; r.wb = (r.wa + 1) & wmask;
  r.ma = bits << sh; r.mb = sh ? bits >> (64 - sh) : 0ull;
  return r;
}
__device__ __forceinline__ unsigned long long clv_bcast64(unsigned long long v)     // lane 0's value in every lane
{
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v), hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
// value of lane l (wave-uniform l) in every lane
__device__ __forceinline__ float clv_lane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// sum of lds[first], lds[first + step], ... (count terms) added one after the other in that order, every lane the same result: the
// lanes fetch 64 terms at a time and the additions run on registers (a chain of LDS round trips otherwise)
__device__ __forceinline__ float clv_ordered_sum(const float *lds, int first, int step, int count, float acc, int lane)
{
#pragma clang fp contract(off)
  for (int c0 = 0; c0 < count; c0 += 64) {
    const int n = min(64, count - c0);
    const float v = lane < n ? lds[first + (c0 + lane) * step] : 0.f;
    for (int j = 0; j < n; j++) acc += clv_lane(v, j);
  }
  return acc;
}

__global__ __launch_bounds__(64) void k_clever(CleverArgs a)
{
#pragma clang fp contract(off)
  constexpr int W = 128;                                  // half width of the pulse neighbourhood held in LDS
  __shared__ float s_pw[2 * W + 8], s_old[2 * W + 8], s_spw[256], s_in[2 * 2 * W + 8], s_avg[8];
  __shared__ float2 s_tf[2 * W + 8], s_ty[2 * W + 8], s_otf[2 * W + 8], s_oty[2 * W + 8];
  __shared__ unsigned char s_fl[2 * W + 8], s_sfl[256];
  // two coupled channels: X is channel 0, Y channel 1, whichever of them this context owns
  float2 *const ring_x = (a.twochan && a.chan) ? a.timf2y : a.timf2w, *const ring_y = a.twochan ? (a.chan ? a.timf2w : a.timf2y) : nullptr;
  const int lane = threadIdx.x, mask = a.mask, total = a.total, R = a.R, pwid = a.pwid, rs = a.rs;
  const int wmask = ((mask + 1) >> 6) - 1, first_word = a.pbeg >> 6, nwords = ((a.pbeg & 63) + total + 64) >> 6;
  BlankState *s = a.st;
  const float nfl = (float)s->clever_limit;
  const float sizlim = (float)(0.1 * (double)s->noise_floor);
  if (lane < 8) s_avg[lane] = 0.f;
  // offsets o count from pbeg (o = total is blnk_pend); they may go below 0 / above total where the reference's pointers do
  auto POS = [&](int o) { return (a.pbeg + o) & mask; };
  auto next_candidate = [&](int o_start) -> int {         // blank1.c:781-794: first sample >= o_start above the limit and not flagged
    if (o_start >= total) return total;
    const int pos0 = POS(o_start);
    int w = pos0 >> 6;
    bool first = true;
    for (;;) {
      const int rel = (w + lane - first_word) & wmask;
      unsigned long long v = rel < nwords ? __hip_atomic_load(&a.cand[(w + lane) & wmask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      if (first && lane == 0) v &= ~0ull << (pos0 & 63);
      first = false;
      const unsigned long long any = __ballot(v != 0ull);
      if (any) {
        const int L = __ffsll((long long)any) - 1;
        const unsigned int lo = __shfl((unsigned int)v, L), hi = __shfl((unsigned int)(v >> 32), L);
        const unsigned long long vv = ((unsigned long long)hi << 32) | lo;
        const int pos = (((w + L) & wmask) << 6) + (__ffsll((long long)vv) - 1);
        const int o = (pos - a.pbeg) & mask;
        return o < total ? o : total;
      }
      w += 64;
      if (((w - first_word) & wmask) >= nwords || ((w - first_word) & wmask) < 64) return total;
    }
  };
  auto stage_search = [&](int base) {
    __syncthreads();
    for (int i = lane; i < 256; i += 64) { const int pos = POS(base + i); s_spw[i] = a.pwr[pos]; s_sfl[i] = a.flag[pos]; }
    __syncthreads();
  };
  // phase 0: one region per wave (work list); phase 1: the whole span by the first wave, only when the extents collided
  if (a.phase == 1 && (a.reg_ctl[1] == 0 || blockIdx.x != 0)) return;
  if (a.phase == 0 && a.reg_ctl[1] != 0) return;          // already known to go to the one-wave replay (too many regions, forced): nothing to undo
  const int nreg = a.phase == 1 ? 1 : a.reg_ctl[0];
  for (int reg = blockIdx.x; reg < nreg; reg += gridDim.x) {
  const int r_begin = a.phase == 1 ? 0 : a.reg_start[reg];
  // A region ends short of the next one's start by the reach of a subtraction: the next region rewrites samples up to half the largest
  // fit size BEFORE its first candidate, and a residue there that exceeds the limit is a candidate the reference's walk never sees (it
  // lies behind the walk by the time the pulse is handled) -- so it is nobody's: not the next region's (behind its start) and not this
  // one's.  This region's own candidates cannot lie there: the gap to the next start is wider than any reach.
  const int r_end = (a.phase == 1 || reg + 1 >= nreg) ? total : a.reg_start[reg + 1] - (a.bln_size[a.largest] / 2 + 1);
  int ext_lo = r_begin, ext_hi = r_begin;
  int pf = r_begin, fitted = 0, rejected = 0;
  const long long t_begin = a.reg_dbg ? (long long)wall_clock64() : 0;
  for (;;) {
    // (the ring samples and flags this wave has rewritten come back in program order; the candidate words it changes with atomics are
    // read with atomic loads -- no fence: a device-scope fence writes the L2 back, ~100 us per pulse)
    pf = next_candidate(pf);
    if (pf >= r_end) { pf = r_end; break; }
    // ---- the maximum that stays the maximum for blnfit_range samples (blank1.c:795-824)
    // the walk "o++; new maximum -> m = R; m--" until m == 0 or o == total, 64 positions at a time: running maximum and the place of
    // its last rise by wave scans (strict >: the first of equal values keeps the place, as in the walk)
    int o = pf - 1, p_max = pf, m = R, base = pf;
    float powermax = 10.f;
    stage_search(base);
    while (o != total && m > 0) {
      if (o + 1 - base > 256 - 64) { base = o + 1; stage_search(base); }
      const int oj = o + 1 + lane;
      const bool valid = oj <= total;
      const float v = (valid && s_sfl[oj - base] < 64) ? s_spw[oj - base] : -1.f;     // powers are >= 0: -1 never rises above anything
      float run = v;                                      // inclusive prefix maximum over the lanes
      for (int d = 1; d < 64; d <<= 1) { const float u = __shfl_up(run, d, 64); if (lane >= d) run = fmaxf(run, u); }
      float before = __shfl_up(run, 1, 64);               // maximum of everything before this position, the walk's powermax included
      before = lane == 0 ? powermax : fmaxf(before, powermax);
      int last = v > before ? lane : -1;                  // lane of the last rise at or before this position
      for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(last, d, 64); if (lane >= d) last = max(last, u); }
      const int mj = last >= 0 ? R - 1 - (lane - last) : m - (lane + 1);
      const unsigned long long stop = __ballot(valid && (mj <= 0 || oj == total));
      const int js = stop ? __ffsll((long long)stop) - 1 : 63;
      const int last_s = __shfl(last, js, 64);
      powermax = fmaxf(powermax, __shfl(run, js, 64));
      if (last_s >= 0) p_max = o + 1 + last_s;
      m = __shfl(mj, js, 64);
      o += js + 1;
    }
    ext_hi = max(ext_hi, o);
    if (m > 0) break;                                     // too close to the end of the span: next call
    bool no_pulse = false;
    auto FLAG = [&](int oo) -> int { return (oo >= base && oo < base + 256) ? (int)s_sfl[oo - base] : (int)a.flag[POS(oo)]; };
    if (FLAG(p_max - 1) >= 64) { pf = p_max; no_pulse = true; }
    else {
      if (p_max + 1 == total) break;
      if (FLAG(p_max + 1) >= 64) no_pulse = true;
    }
    if (no_pulse) {                                       // blank1.c:833-856: next to a region already handled: walk on while the power does not rise
      while (pf != total) {
        const float v = a.pwr[POS(pf)];
        if (!(v <= powermax || a.flag[POS(pf)] > 64)) break;
        powermax = v; pf++;
      }
      ext_hi = max(ext_hi, pf); ext_lo = min(ext_lo, p_max - 1);
      if (pf == total) break;
      continue;
    }
    // what the fit can read or change around p_max: the shells of the largest fit size, the phase window, the subtracted span
    { const int wn = max(a.bln_size[a.largest] / 2, pwid) + 1; ext_lo = min(ext_lo, p_max - wn); ext_hi = max(ext_hi, p_max + wn); }
    // ---- neighbourhood of the pulse into LDS
    __syncthreads();
    for (int i = lane; i <= 2 * W; i += 64) {
      const int pos = POS(p_max - W + i); s_pw[i] = a.pwr[pos]; s_tf[i] = ring_x[pos]; s_fl[i] = a.flag[pos];
      if (a.twochan) s_ty[i] = ring_y[pos];
    }
    __syncthreads();
    // ---- unresolved multiple pulses? mean power of the shells between successive fit sizes (blank1.c:909-951)
    int bln_no = 0, ia = W + 1, ib = W - 1, k = 2;
    for (;;) {
      powermax = s_pw[W];
      float t1 = powermax * a.bln_rest[bln_no];
      if (t1 < sizlim) break;
      { const int cnt = a.bln_size[bln_no] > k ? (a.bln_size[bln_no] - k + 1) / 2 : 0;    // pairs s_pw[ia] + s_pw[ib], added in the walk's order
        t1 = 0.f;
        for (int c0 = 0; c0 < cnt; c0 += 64) {
          const int n = min(64, cnt - c0);
          const float pr = lane < n ? s_pw[ia + c0 + lane] + s_pw[ib - c0 - lane] : 0.f;
          for (int j = 0; j < n; j++) t1 += clv_lane(pr, j);
        }
        ia += cnt; ib -= cnt; k += 2 * cnt; }
      s_avg[bln_no] = t1 / powermax;
      bln_no++;
      if (bln_no > a.largest) break;
    }
    bln_no--;
    __syncthreads();
    while (bln_no >= 0 && s_avg[bln_no] > a.bln_avgmax[bln_no]) bln_no--;
    // ---- subtract_onechan_pulse (blank1.c:36-232); two coupled channels: get_pulse_pol, transform_timf2_pol, subtract_twochan_pulse (:232-609)
    float rv = -1.f;
    if (bln_no >= 0) {
      const int sub = a.bln_size[bln_no];
      bool go = true;
      float pc1 = 1.f, pc2 = 0.f, pc3 = 0.f;
      if (a.twochan) {
        // polarisation of the pulse from the channels' powers and cross product around the peak: sums in the reference's order, every lane the same
        float x2 = 0.f, y2 = 0.f, re_xy = 0.f, im_xy = 0.f;
        for (int i = 0; i <= 2 * pwid; i++) {
          const float2 x = s_tf[W - pwid + i], y = s_ty[W - pwid + i];
          x2 += x.x * x.x + x.y * x.y; y2 += y.x * y.x + y.y * y.y;
          re_xy += x.x * y.x + x.y * y.y; im_xy += x.y * y.x - x.x * y.y;
        }
        float t1 = x2 + y2;
        x2 /= t1; y2 /= t1; re_xy /= t1; im_xy /= t1;
        const float t2 = re_xy * re_xy + im_xy * im_xy;
        const float noi2 = x2 * y2 - t2;
        if ((double)noi2 > 0.15) go = false;              // the two channels do not carry one common signal here
        else {
          const float x2s = x2 - noi2, y2s = y2 - noi2;
          if (x2s > 0.f) {
            pc1 = (float)sqrt((double)x2s);
            if (y2s > 0.f && t2 > 0.f) {
              const float sina = (float)sqrt((double)y2s);
              pc2 = (float)((double)(sina * re_xy) / sqrt((double)t2)); pc3 = (float)((double)(sina * im_xy) / sqrt((double)t2));
              t1 = (float)sqrt((double)(pc1 * pc1 + pc2 * pc2 + pc3 * pc3));
              pc1 /= t1; pc2 /= t1; pc3 /= t1;
            } else { if (x2 > y2) { pc1 = 1.f; pc2 = 0.f; } else { pc1 = 0.f; pc2 = 1.f; } pc3 = 0.f; }
          } else { pc1 = 0.f; pc2 = 1.f; pc3 = 0.f; }
        }
      }
      for (int i = lane; i <= 2 * pwid; i += 64) {
        float2 t = s_tf[W - pwid + i];
        if (a.twochan) {                                  // the one signal that carries the pulse
          const float2 y = s_ty[W - pwid + i];
          t = make_float2(pc1 * t.x + pc2 * y.x - pc3 * y.y, pc1 * t.y + pc2 * y.y + pc3 * y.x);
        }
        const int kk = rs - 2 * pwid + 2 * i;
        const float t3 = a.phasefunc[kk], t4 = a.phasefunc[kk + 1];
        s_in[2 * i] = t.x * t3 + t.y * t4; s_in[2 * i + 1] = t.y * t3 - t.x * t4;
      }
      __syncthreads();
      const int imax = pwid;
      float c1 = 0.f, c2 = 0.f;
      for (int i = imax - 1; i <= imax + 1; i++) { const float t1 = s_in[2 * i], t2 = s_in[2 * i + 1], t3 = sqrtf(t1 * t1 + t2 * t2); c1 += t3 * t1; c2 += t3 * t2; }
      float t1 = c1 * c1 + c2 * c2;
      if (a.twochan ? sqrtf(t1) < 4.f : t1 < 32.f) go = false;       // blank1.c:274-275 / :101-103
      float t3 = 0.f, t4 = 0.f;
      if (go) {
        t1 = sqrtf(t1); c1 /= t1; c2 /= t1;
        __syncthreads();
        for (int i = lane; i <= 2 * pwid; i += 64) {
          const float x = s_in[2 * i], y = s_in[2 * i + 1];
          s_in[2 * i] = c1 * x + c2 * y; s_in[2 * i + 1] = c1 * y - c2 * x;
        }
        __syncthreads();
        for (int i = 0; i <= 2 * pwid; i++) { t3 += s_in[2 * i] * s_in[2 * i]; t4 += s_in[2 * i + 1] * s_in[2 * i + 1]; }
        if ((double)t4 > 0.25 * (double)t3) go = false;   // too much power off the pulse's phase
      }
      if (go) {
        t4 = s_in[2 * imax - 2] - s_in[2 * imax + 2];
        t3 = 2 * (s_in[2 * imax - 2] + s_in[2 * imax + 2] - 2 * s_in[2 * imax]);
        if (t3 == 0.f) { go = false; rv = -2.f; }
      }
      if (go) {
        // peak position within the sample by a parabola, then the reference pulse for it
        if (a.twochan) { t4 /= 2 * t3; t4 = t4 < 0 ? (float)(-sqrt((double)-t4)) : (float)sqrt((double)t4); }
        else { t4 /= t3; if (t4 < 0) t4 = (float)(-sqrt(0.5) * sqrt((double)-t4)); else t4 = (float)(sqrt(0.5) * sqrt((double)t4)); }
        int j = (int)(LRH_MAX_REFPULSES_K * ((double)t4 + 0.5) + 0.5);
        if (j < 0) j = 0;
        if (j >= LRH_MAX_REFPULSES_K) j = LRH_MAX_REFPULSES_K - 1;
        const int mrp = 2 * a.pulindex[j] * rs;
        { const float af = a.amp_dev ? *a.amp_dev : s->amp_factor;   // liminfo_amplitude_factor (blank1.c:143-144 / :323-324)
          if (a.twochan) { c1 = c1 * s_in[2 * imax] * af; c2 = c2 * s_in[2 * imax] * af; } else { c1 *= s_in[2 * imax] * af; c2 *= s_in[2 * imax] * af; } }
        __syncthreads();
        for (int jj = lane; jj <= sub; jj += 64) {
          const int q = W - sub / 2 + jj, kk = rs - sub + 2 * jj;
          const float r1 = a.refpulse[mrp + kk], r2 = a.refpulse[mrp + kk + 1];
          const float2 x = s_tf[q];
          s_old[q] = s_pw[q];
          s_otf[q] = x; if (a.twochan) s_oty[q] = s_ty[q];         // as staged: what the undo log keeps
          if (a.twochan) {
            const float2 y = s_ty[q];
            const float re_a = c1 * r1 - c2 * r2, im_a = c1 * r2 + c2 * r1;
            const float re_x = x.x - pc1 * re_a, im_x = x.y - pc1 * im_a;
            const float re_y = y.x - (pc2 * re_a + pc3 * im_a), im_y = y.y - (pc2 * im_a - pc3 * re_a);
            s_tf[q] = make_float2(re_x, im_x); s_ty[q] = make_float2(re_y, im_y);
            s_pw[q] = re_x * re_x + im_x * im_x + re_y * re_y + im_y * im_y;
          } else {
            const float re = x.x - c1 * r1 + c2 * r2, im = x.y - c1 * r2 - c2 * r1;
            s_tf[q] = make_float2(re, im); s_pw[q] = re * re + im * im;
          }
        }
        __syncthreads();
        t3 = clv_ordered_sum(s_old, W - sub / 2, 1, 2 * (sub / 2) + 1, 0.f, lane);
        t4 = clv_ordered_sum(s_pw, W - sub / 2, 1, 2 * (sub / 2) + 1, 0.f, lane);
        rv = t4 / t3;
        if (rv > 0.5f) {                                  // the fit removed too little: put the samples back (one channel: with the reference's signs, see the oracle)
          __syncthreads();
          for (int jj = lane; jj <= sub; jj += 64) {
            const int q = W - sub / 2 + jj, kk = rs - sub + 2 * jj;
            const float r1 = a.refpulse[mrp + kk], r2 = a.refpulse[mrp + kk + 1];
            const float2 x = s_tf[q];
            if (a.twochan) {
              const float2 y = s_ty[q];
              const float re_a = c1 * r1 - c2 * r2, im_a = c1 * r2 + c2 * r1;
              const float re_x = x.x + pc1 * re_a, im_x = x.y + pc1 * im_a;
              const float re_y = y.x + (pc2 * re_a + pc3 * im_a), im_y = y.y + (pc2 * im_a - pc3 * re_a);
              s_tf[q] = make_float2(re_x, im_x); s_ty[q] = make_float2(re_y, im_y);
              s_pw[q] = re_x * re_x + im_x * im_x + re_y * re_y + im_y * im_y;
            } else {
              const float re = x.x + c1 * r1 + c2 * r2, im = x.y + c1 * r2 - c2 * r1;
              s_tf[q] = make_float2(re, im); s_pw[q] = re * re + im * im;
            }
          }
          rv = -5.f;
        }
        __syncthreads();
        // undo log: the first wave to touch a sample in this call keeps its original.  One atomic or per word of the bitmap tells which
        // of the samples are new to the log, one atomic add reserves their entries; then the samples go out.
        unsigned long long firsts[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
        float opo[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        int nfirst = 0;
        if (a.phase == 0) {
          if (a.twochan) {                                // the own channel's power ring is not staged: read before the atomics go out
#pragma unroll
            for (int it = 0; it < 5; it++) { const int jj = 64 * it + lane; if (jj <= sub) opo[it] = a.pwr_own[POS(p_max - sub / 2 + jj)]; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          unsigned long long olda[5], oldb[5];
#pragma unroll
          for (int it = 0; it < 5; it++) {
            olda[it] = 0ull; oldb[it] = 0ull;
            if (64 * it > sub) continue;
            const ClvWordPair wp = clv_words(POS(p_max - sub / 2 + 64 * it), __ballot(64 * it + lane <= sub), wmask);
            if (lane == 0) { if (wp.ma) olda[it] = atomicOr(&a.logged[wp.wa], wp.ma); if (wp.mb) oldb[it] = atomicOr(&a.logged[wp.wb], wp.mb); }
          }
#pragma unroll
          for (int it = 0; it < 5; it++) {
            if (64 * it > sub) continue;
            const int sh = POS(p_max - sub / 2 + 64 * it) & 63;
            const unsigned long long oa = clv_bcast64(olda[it]), ob = clv_bcast64(oldb[it]);
            const bool was = sh + lane < 64 ? (oa >> (sh + lane)) & 1 : (ob >> (sh + lane - 64)) & 1;
            firsts[it] = __ballot(64 * it + lane <= sub && !was);
            nfirst += __popcll(firsts[it]);
          }
          int base = 0;
          if (nfirst) { if (lane == 0) base = atomicAdd(&a.reg_ctl[3], nfirst); base = __builtin_amdgcn_readfirstlane(base); }
#pragma unroll
          for (int it = 0; it < 5; it++) {
            if ((firsts[it] >> lane) & 1) {
              const int jj = 64 * it + lane, q = W - sub / 2 + jj, e = base + __popcll(firsts[it] & ((1ull << lane) - 1));
              a.bk_pos[e] = POS(p_max - sub / 2 + jj); a.bk_pwr[e] = s_old[q];
              if (a.twochan) { a.bk_tf[e] = a.chan ? s_oty[q] : s_otf[q]; a.bk_ty[e] = a.chan ? s_otf[q] : s_oty[q]; a.bk_pwo[e] = opo[it]; }
              else a.bk_tf[e] = s_otf[q];
            }
            base += __popcll(firsts[it]);
          }
        }
        for (int jj = lane; jj <= sub; jj += 64) {
          const int q = W - sub / 2 + jj, pos = POS(p_max - sub / 2 + jj);
          ring_x[pos] = s_tf[q]; a.pwr[pos] = s_pw[q];
          if (a.twochan) { ring_y[pos] = s_ty[q]; const float2 o = a.chan ? s_ty[q] : s_tf[q]; a.pwr_own[pos] = o.x * o.x + o.y * o.y; }
        }
      }
    }
    const unsigned char value = rv < 0.f ? 65 : 66;
    if (rv < 0.f) rejected++; else fitted++;
    // ---- set_flag (blank1.c:615-682): +-pulsewidth, then outwards for as long as the power keeps falling
    __syncthreads();
    auto PW = [&](int oo) -> float { const int d = oo - p_max; return (d >= -W && d <= W) ? s_pw[d + W] : a.pwr[POS(oo)]; };
    auto SETF = [&](int oo) {                             // inside the neighbourhood: in LDS, written out together below
      const int d = oo - p_max, pos = POS(oo);
      if (d >= -W && d <= W) s_fl[d + W] = value;
      else if (lane == 0) { a.flag[pos] = value; if (oo >= 0 && oo <= total) atomicAnd(&a.cand[pos >> 6], ~(1ull << (pos & 63))); }
    };
    SETF(p_max);
    int pa = p_max, pb = p_max;
    for (int i = 0; i < pwid; i++) { pb--; pa++; SETF(pa); SETF(pb); }
    int p0 = pb; pb--;
    if (!(pb < 1)) while (PW(pb) < PW(p0) && pb != 0) { SETF(pb); p0 = pb; pb--; }
    p0 = pa; pa++;
    if (!(pa >= total)) while (PW(pa) < PW(p0) && pa != total) { p0 = pa; SETF(pa); pa++; }
    ext_lo = min(ext_lo, pb); ext_hi = max(ext_hi, pa);
    __syncthreads();
    // the flags just set are the run pb+1 .. pa-1
    for (int oo = max(pb + 1, p_max - W) + lane; oo <= min(pa - 1, p_max + W); oo += 64) a.flag[POS(oo)] = value;
    // ---- candidate bits of the samples the subtraction rewrote
    for (int c0 = 0; c0 <= 2 * W; c0 += 64) {
      const int i = c0 + lane, oo = p_max - W + i;
      const bool in = i <= 2 * W && oo >= 0 && oo <= total;
      const bool hot = in && s_pw[min(i, 2 * W)] > nfl && s_fl[min(i, 2 * W)] <= 64;
      const int pos0 = POS(p_max - W + c0);
      const ClvWordPair off = clv_words(pos0, __ballot(in && !hot), wmask), on = clv_words(pos0, __ballot(hot), wmask);
      if (lane == 0) {
        if (off.ma) atomicAnd(&a.cand[off.wa], ~off.ma);
        if (off.mb) atomicAnd(&a.cand[off.wb], ~off.mb);
        if (on.ma) atomicOr(&a.cand[on.wa], on.ma);
        if (on.mb) atomicOr(&a.cand[on.wb], on.mb);
      }
    }
  }
  if (lane == 0) {
    if (a.phase == 1) { s->clever_out[0] = POS(pf); s->clever_out[1] = fitted; s->clever_out[2] = rejected; s->clever_serial_calls++; }
    else {
      a.reg_ext[2 * reg] = ext_lo; a.reg_ext[2 * reg + 1] = ext_hi;
      if (a.reg_dbg) { a.reg_dbg[2 * reg] = fitted + rejected; a.reg_dbg[2 * reg + 1] = (int)((long long)wall_clock64() - t_begin); }
      if (fitted) atomicAdd(&s->clever_out[1], fitted);
      if (rejected) atomicAdd(&s->clever_out[2], rejected);
      if (reg == nreg - 1) a.reg_ctl[2] = pf;           // where the walk of the whole span stops (blank1.c:1458)
    }
  }
  __syncthreads();
  }
}

// The search in two halves: `front` = candidate bits, flag clear and the list of regions -- reads the power ring, touches no sample, so it
// may run as soon as the span's samples exist and the limit is final; `back` = the replay, the check of the extents and the (normally idle)
// one-wave replay, which the host issues when the check reports a collision (clever_out[3]).  parts: 1 front, 2 back, 4 replay.
hipError_t launch_clever(const CleverArgs &a0, hipStream_t st, int parts)
{
  CleverArgs a = a0;
  const int nwords = ((a.pbeg & 63) + a.total + 64) >> 6;
  const dim3 gp((nwords + 3) / 4 < 2048 ? (nwords + 3) / 4 : 2048);
  const int rblocks = std::max(1, std::min(CLV_RBLOCKS_MAX, (nwords + 1023) / 1024));      // >= 4 words per thread
  a.phase = 0;
  if (parts & 1) {
    hipLaunchKernelGGL(k_clever_prep, gp, dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_clever_count, dim3(rblocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_clever_regions, dim3(rblocks), dim3(256), 0, st, a);
  }
  if (parts & 2) {
    hipLaunchKernelGGL(k_clever, dim3(a.max_regions < 16384 ? a.max_regions : 16384), dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_clever_check, dim3(1), dim3(1024), 0, st, a);
  }
  if (parts & 4) {                                         // the extents collided (or the test switch asks for it): samples back, one wave over the span
    a.phase = 1;
    hipLaunchKernelGGL(k_clever_restore, dim3(256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_clever_prep, gp, dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_clever, dim3(1), dim3(64), 0, st, a);
  }
  return hipGetLastError();
}

hipError_t launch_blanker(const BlankArgs &a0, int ring_words, hipStream_t st)
{
  BlankArgs a = a0;
  if (a.phase == 2) {                                    // coupled call, second half: statistics / threshold update only
    a.npartials = 0; a.nremoved = 0;
    hipLaunchKernelGGL(k_blank_update, dim3(1), dim3(256), 0, st, a);
    return hipGetLastError();
  }
  const int ntiles = (a.total + LRH_BLN_TILE) / LRH_BLN_TILE;     // sequence positions 0 .. total
  const int first_pos = (a.pbeg + 1 - a.clr1 - 32) & a.mask;
  const int nwords = (a.total + a.clr1 + a.clr2 + 64 + 31) / 32 + 1;
  a.ncounts = 0;
  if (a.mode != 0) {
    a.npartials = ntiles; a.nremoved = (nwords + 255) / 256; a.ncounts = ntiles;
    hipLaunchKernelGGL(k_blank_scan<LRH_BLN_BACK>, dim3(ntiles), dim3(256), 0, st, a);
    if (!(a.clr1 == 0 && a.clr2 == 1 && a.tiles)) {      // calibrated blanker only
      hipLaunchKernelGGL(k_blank_scan<LRH_BLN_BACK2>, dim3(ntiles), dim3(256), 0, st, a);      // returns at once unless a lane gave up
      const char *e_ = getenv("LRH_BLN_SERIAL"); const int one_lane = e_ ? atoi(e_) : 0;     // (read per call: the comparison test flips it)
      if (one_lane == 1) hipLaunchKernelGGL(k_blank_serial, dim3(1), dim3(1), 0, st, a);
      else if (one_lane == 2 || !a.wbusy) hipLaunchKernelGGL(k_blank_serial_wave, dim3(1), dim3(64), 0, st, a);   // (LRH_BLN_SERIAL=2: the one-wave walk, for comparison)
      else {                                              // tile-parallel walk: three launches that return at once unless both scans gave up
        a.nwt = (a.total + LRH_BLN_WTILE - 1) / LRH_BLN_WTILE;
        hipLaunchKernelGGL(k_blank_walk_spec, dim3(a.nwt), dim3(64), 0, st, a);
        hipLaunchKernelGGL(k_blank_walk_chain, dim3(1), dim3(64), 0, st, a);
        hipLaunchKernelGGL(k_blank_walk_final, dim3(a.nwt), dim3(64), 0, st, a);
      }
    }
    else {         // long-run replay: two launches that return at once unless a lane gave up
      hipLaunchKernelGGL(k_blank_runs_pre, dim3(ntiles), dim3(256), 0, st, a);
      hipLaunchKernelGGL(k_blank_runs, dim3(ntiles), dim3(256), 0, st, a);   // k_blank_update takes the flag down
    }
    hipLaunchKernelGGL(k_blank_apply, dim3(a.nremoved), dim3(256), 0, st, a, first_pos >> 5, nwords, ring_words - 1);
    if (a.own || a.post_stats) {                          // per-channel statistic from the own ring after clearing / statistic span shorter than the scan
      a.nremoved = 0;
      a.npartials = a.nstat < 4096 ? 1 : (a.nstat / 4096 < ntiles ? a.nstat / 4096 : ntiles);
      hipLaunchKernelGGL(k_blank_stats, dim3(a.npartials), dim3(256), 0, st, a);
    }
  } else {
    a.nremoved = 0;
    hipLaunchKernelGGL(k_blank_stats, dim3(a.npartials), dim3(256), 0, st, a);
  }
  hipLaunchKernelGGL(k_blank_update, dim3(1), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace lrh

// =====================================================================================================
// selective limiter on the device-resident spectra (control plane, once per averaging period)
// =====================================================================================================
// fft1_update_liminfo is a serial, data-dependent scan over the fft1 bins.  One workgroup: the power block and liminfo
// sit in LDS, everything that is independent per bin (products with the slow average, group minima, hold-off counters,
// the routing words) runs on all threads, and thread 0 walks the two scans exactly like the reference does -- but jumps over
// the stretches where nothing can happen (bins below the limit / below the noise floor) with per-chunk "first hot bin"
// tables the other threads have built.  Arithmetic as the reference (float, sqrt / pow in double).
namespace lrh {
#define LRH_SL_BIG 300000000000000000000000000000000000000.F
// One workgroup, and every pass over the bins is a latency-bound stream through global memory or LDS: 1024 threads keep 16 loads in
// flight per pass where 256 kept 64 trips of one (k_sellim 500-650 us -> see DESIGN.md 4.8).
#define LRH_SL_THREADS 1024
__device__ __forceinline__ float sl_three_smallest(const float *v, int ia, int ib)
{
  float t1 = LRH_SL_BIG, t2 = LRH_SL_BIG, t3 = LRH_SL_BIG;
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

// selfreq_liminfo (sellim.c:38-157, float path): the selected passband keeps its own routing; then liminfo_amplitude_factor,
// what the strong bins take away from a pulse's amplitude (:108-155).  Thread 0; B = the table in LDS.
__device__ __forceinline__ void sl_selfreq(const SellimArgs &a, float *B, int tid)
{
  const int N = a.n;
  if (tid == 0 && a.selfreq >= 0) {
    int ia = (int)(a.selfreq * a.points_per_hz);
    int k = (int)(a.bw_fftxpts * .7);
    if (a.par6 == 0) k += 3;
    if (a.second_fft) { int ratio = a.n2 / N; if (ratio < 1) ratio = 1; ia /= ratio; k /= ratio; if (k < 3) k = 3; }
    int ib = ia + k; ia -= k;
    if (ia < 0) ia = 0;
    if (ib >= N) ib = N - 1;
    a.st->sel_ia = ia; a.st->sel_ib = ib;
    if (a.ston_scale) { for (int i = ia; i <= ib; i++) B[i] = -1; }
    else {
      float t1 = 0, t2 = 2;
      for (int i = ia; i <= ib; i++) { if (B[i] < 0) t1 = 1; if (B[i] > 0 && t2 > B[i]) t2 = B[i]; }
      bool skip = false;
      if (t2 > 1) { if (t1 == 0) skip = true; t2 = 1; }
      if (!skip) {
        t1 = 1 / t2;
        t1 *= (float)sqrt((double)(float)(a.n2 / N));
        if (a.par5 == 2) t2 = -1;
        if (a.par5 == 1) { if (t1 < 0x7fff / a.maxlevel) t2 = 0; }
        if (a.par5 == 0) { if (t1 < 0x7ffff / a.maxlevel) t2 = 0; }
        for (int i = ia; i <= ib; i++) B[i] = t2;
      }
    }
  }
  __shared__ int s_strong;
  if (tid == 0) s_strong = 0;
  __syncthreads();                                         // also: the passband above is in place for everybody
  if (a.second_fft && !a.desired) {                        // uncalibrated: the share of strong bins in the passband (a count: any order)
    int k = 0;
    for (int i = a.first_inband + tid; i <= a.last_inband; i += LRH_SL_THREADS) if (B[i] != 0) k++;
    if (k) atomicAdd(&s_strong, k);
  }
  __syncthreads();
  if (tid == 0) {
    float f = 1.f;
    if (a.second_fft) {
      if (a.desired) {                                     // calibrated: a float sum, in the reference's order
        float t1 = 0;
        for (int i = a.first_point; i <= a.last_point; i++) if (B[i] != 0) t1 += a.desired[i] * a.desired[i];
        f = a.desired_totsum / (a.desired_totsum - t1);
      } else {
        const int n = a.last_inband - a.first_inband + 1;
        f = (float)(n) / (n - s_strong);
      }
    }
    if (f > 2) f = 0;
    a.bst->amp_factor = f;
  }
}

// the routing words k_timf2 consumes (bit s of word i: bin i + s N/R0 is weak, timf2.c:50) and the weak-bin count, from the finished
// table in LDS: the limiter kernels end with it (a separate one-workgroup launch cost 20-90 us on the path to the next make_timf2)
__device__ __forceinline__ void sl_pack(const SellimArgs &a, const float *B, int tid)
{
  __shared__ int s_low;
  if (tid == 0) s_low = 0;
  __syncthreads();
  int low = 0;
  if (a.r0 == 0) {                                        // four-step timf2: dense bits, bit (k & 31) of word k >> 5 (lrh_set_liminfo)
    for (int w = tid; w < a.n / 32; w += LRH_SL_THREADS) {
      unsigned int m = 0;
      for (int s = 0; s < 32; s++) if (B[32 * w + s] == 0) { m |= 1u << s; low++; }
      a.pack[w] = m;
    }
  } else {
    const int nb = a.n / a.r0;
    for (int i = tid; i < nb; i += LRH_SL_THREADS) {
      unsigned int m = 0;
      for (int s = 0; s < a.r0; s++) if (B[i + s * nb] == 0) { m |= 1u << s; low++; }
      a.pack[i] = m;
    }
  }
  for (int off = 32; off > 0; off >>= 1) low += __shfl_xor(low, off);
  if ((tid & 63) == 0 && low) atomicAdd(&s_low, low);
  __syncthreads();
  if (tid == 0) a.st->low = s_low;
}

// the same by one wave: every lane keeps the three smallest of its share, then the triples are merged across the lanes (the three
// smallest values of a set do not depend on the order they are met in)
__device__ __forceinline__ float sl_three_smallest_wave(const float *v, int ia, int ib, int lane)
{
  float t1 = LRH_SL_BIG, t2 = LRH_SL_BIG, t3 = LRH_SL_BIG;
  // insertion into the sorted triple without branches (the nested ifs diverge in every lane: ~100 ns per value, 19 us per update)
  auto put = [&](float x) {
    const float m1 = fmaxf(t1, x); t1 = fminf(t1, x);
    const float m2 = fmaxf(t2, m1); t2 = fminf(t2, m1);
    t3 = fminf(t3, m2);
  };
  for (int i = ia + lane; i < ib; i += 64) put(v[i]);
  for (int off = 32; off > 0; off >>= 1) {
    const float o1 = __shfl_xor(t1, off), o2 = __shfl_xor(t2, off), o3 = __shfl_xor(t3, off);
    put(o1); put(o2); put(o3);
  }
  return (float)(0.3333333 * (t1 + t2 + t3));
}

// BIG (fft1_size 32768): the power block alone fills the LDS, so the table and the group minima live in global memory (two scratch
// arrays of the context; one CU, every access an L1 / L2 hit) -- same code, thread 0's walks pay a cache latency per dependent read.
template <bool BIG>
__global__ __launch_bounds__(LRH_SL_THREADS) void k_sellim(SellimArgs a)
{
  extern __shared__ float sm[];
  const int N = a.n, tid = threadIdx.x;
  float *A = sm + 8;                       // power block, later fftt_tmp (the scans look two bins beyond their range)
  float *B = BIG ? a.big_b : A + N + 16;   // liminfo
  float *G = BIG ? a.big_g : B + N + 8;    // liminfo_group_min (at most N/16 groups; sized N/4 + 8)
  unsigned int *hotw = (unsigned int *)(BIG ? A + N + 16 : G + N / 4 + 8);    // [N/32 + 2] one bit per bin: above the limit / above the noise floor
  unsigned int *touched = hotw + (N + 31) / 32 + 4;        // [N/32 + 2] pass 1: bins thread 0 has decided
  unsigned int *l2set = touched + (N + 31) / 32 + 4;       // second level over hotw: bit w set = word w has a set bit / ...
  unsigned int *l2clr = l2set + (N + 31) / 1024 + 2;       // ... = word w has a clear bit (thread 0 crosses an empty band in N/1024 reads)
  __shared__ int s_pass2; __shared__ float s_limit, s_nf; __shared__ int s_k, s_ia;
  const int NW = (N + 31) / 32;
  long long ts[10]; int nts = 0;
  auto stamp = [&]() { if (a.debug && tid == 0 && nts < 10) ts[nts++] = wall_clock64(); };
  stamp();
  for (int i = tid; i < N; i += LRH_SL_THREADS) { A[i] = a.sumsq[i]; B[i] = a.liminfo[i]; }
  for (int i = tid; i < 8; i += LRH_SL_THREADS) { A[-8 + i] = 0.f; A[N + i] = 0.f; A[N + 8 + i] = 0.f; }
  if (tid == 0) {
    int tot = a.st->sumsq_tot + a.avg1;
    if (tot > a.spek_avgnum) tot = a.spek_avgnum;
    a.st->sumsq_tot = tot;
    s_pass2 = tot >= a.spek_avgnum;
    float t1 = (float)a.maxlevel, limit = t1 * t1 * a.avg1 * 1;
    limit *= N; limit /= a.n2;
    s_limit = limit;
  }
  __syncthreads();
  const float limit = s_limit;
  const int sel_ia = a.st->sel_ia, sel_ib = a.st->sel_ib, par7 = a.par7;
  const int ix = a.first_point, iy = a.last_point - 1;
  auto build_bits = [&](float thr) {       // all threads: bit i = A[i] > thr; a wave reads 64 consecutive bins (no bank conflicts) and votes
    const int lane = tid & 63, wave = tid >> 6, nwaves = LRH_SL_THREADS / 64;
    for (int r = wave; 64 * r < 32 * (NW + 2); r += nwaves) {
      const int i = 64 * r + lane;
      const unsigned long long m = __ballot(i < N && A[i] > thr);
      if (lane == 0) { hotw[2 * r] = (unsigned int)m; if (2 * r + 1 < NW + 2) hotw[2 * r + 1] = (unsigned int)(m >> 32); }
    }
  };
  // second level of the bit words (all threads, after build_bits and a barrier): one bit per word
  auto build_l2 = [&]() {
    const int nw2 = (NW + 31) / 32;
    for (int v = tid; v < nw2; v += LRH_SL_THREADS) {
      unsigned int ms = 0, mc = 0;
      for (int b = 0; b < 32 && 32 * v + b < NW; b++) { const unsigned int h = hotw[32 * v + b]; if (h) ms |= 1u << b; if (~h) mc |= 1u << b; }
      l2set[v] = ms; l2clr[v] = mc;
    }
  };
  // thread 0's jumps: a dependent LDS read and its branch cost ~100 ns, so the walk finds the next interesting word through the
  // second-level bits (1024 bins per read) instead of reading the words of an empty stretch one after the other
  auto next_word = [&](const unsigned int *l2, int w) -> int {   // first word >= w whose second-level bit is set (NW if none)
    while (w < NW) {
      const unsigned int m2 = l2[w >> 5] >> (w & 31);
      if (m2) return w + __ffs(m2) - 1;
      w = (w | 31) + 1;
    }
    return NW;
  };
  auto next_set = [&](int i) -> int {      // first bin >= i whose bit is set (N if none)
    if (i >= N) return N;
    { const unsigned int m = hotw[i >> 5] >> (i & 31); if (m) { const int r = i + __ffs(m) - 1; return r < N ? r : N; } }
    const int w = next_word(l2set, (i >> 5) + 1);
    if (w >= NW) return N;
    const int r = 32 * w + __ffs(hotw[w]) - 1;
    return r < N ? r : N;
  };
  auto next_clear = [&](int i) -> int {    // first bin >= i whose bit is clear (N if none)
    if (i >= N) return N;
    { const unsigned int m = ~hotw[i >> 5] >> (i & 31); const int room = 32 - (i & 31);
      if (m & (room == 32 ? 0xffffffffu : ((1u << room) - 1))) { const int r = i + __ffs(m) - 1; return r < N ? r : N; } }
    const int w = next_word(l2clr, (i >> 5) + 1);
    if (w >= NW) return N;
    const int r = 32 * w + __ffs(~hotw[w]) - 1;
    return r < N ? r : N;
  };
  // ---- pass 1 (sellim.c:789-865): bins at or below the limit become weak; a run above the limit gets one attenuation over
  // its whole width and tapered skirts.  The serial scan zeroes a bin when it passes it and reads, ahead of itself, the previous
  // update's values.  Here B keeps the previous values while thread 0 visits the runs only: a bin the scan has passed (below
  // `done`) counts as zero when it is at or below the limit and thread 0 has not decided it (`touched`); afterwards all threads
  // zero those bins for good.  No global reads inside the walk.
  stamp();
  build_bits(limit);
  for (int w = tid; w < NW + 2; w += LRH_SL_THREADS) touched[w] = 0u;
  __syncthreads();
  build_l2();
  __syncthreads();
  auto zeroable = [&](int j) -> bool { return !(A[j] > limit) && (j > sel_ib || j < sel_ia || par7 == 0); };
  if (tid == 0) {
    auto touch = [&](int j) { touched[j >> 5] |= 1u << (j & 31); };
    auto is_touched = [&](int j) -> bool { return (touched[j >> 5] >> (j & 31)) & 1u; };
    int ia = ix;
    for (;;) {
      int nh = next_set(ia);
      if (nh >= iy) break;                                   // the serial loop handles bins ia < iy
      ia = nh;
      const int done = ia;                                   // bins below `done` hold this update's values, the rest the previous one's
      auto cur = [&](int j) -> float { return (j < done && j >= ix && !is_touched(j) && zeroable(j)) ? 0.f : B[j]; };
      float maxval = A[ia];
      int ib = next_clear(ia + 1); if (ib > iy + 1) ib = iy + 1;       // while(sumsq[ib] > limit && ib <= iy) ib++
      for (int j = ia + 1; j < ib; j++) if (A[j] > maxval) maxval = A[j];
      while (ia > ix && A[ia - 1] / A[ia] < 0.3) ia--;
      while (ib < iy && A[ib + 1] / A[ib] < 0.3) ib++;
      int ja = ia, jb = ib;
      float t1 = cur(ja), t2;
      for (int j = ja + 1; j <= jb; j++) { const float v = cur(j); if (v > 0 && v < t1) t1 = v; }
      t2 = (float)sqrt((double)(limit / maxval));
      if (t1 / t2 > 0.1 && t1 / t2 < 10) t2 = (float)(0.8 * t1 + 0.2 * t2);
      if (ja > sel_ib || jb < sel_ia || par7 == 0) {
        for (int j = ja; j <= jb; j++) {
          if (j > sel_ib || j < sel_ia || par7 == 0) B[j] = t2;
          else if (j < done) B[j] = cur(j);                  // inside the selected passband: what the scan left there
          touch(j);
        }
      } else for (int j = ja; j <= jb; j++) { if (j < done) B[j] = cur(j); touch(j); }    // left alone by the serial scan
      t1 = t2;
      int j = 1 + (ib - ia) / 4;
      while (ia > ix && j > 0) {
        j--; ia--; ja = ia;
        t1 = (float)pow((double)t1, 0.9);
        const float v = cur(ja);
        if (v <= 0 || v > t1) { if (ja > sel_ib || ja < sel_ia || par7 == 0) { B[ja] = t1; touch(ja); } }
        else break;
      }
      j = 1 + (ib - ia) / 4;
      while (ib < iy && j > 0) {
        j--; ib++; jb = ib;
        t2 = (float)pow((double)t2, 0.9);
        const float v = B[jb];                               // not reached yet by the serial scan: the previous update's value
        touch(jb);
        if (v <= 0 || v > t1) B[jb] = t2;
        else break;                                          // the scan resumes behind this bin: it keeps its previous value
      }
      ia = ib + 1;
      if (ia >= iy) break;
    }
  }
  __syncthreads();
  for (int i = ix + tid; i < iy; i += LRH_SL_THREADS) if (zeroable(i) && !((touched[i >> 5] >> (i & 31)) & 1u)) B[i] = 0;
  __syncthreads();
  stamp();
  if (s_pass2) {
    // ---- pass 2 (sellim.c:866-1147): noise floor of the slow average, everything above it joins the strong signals
    const int gp = a.group_points;
    for (int i = tid; i < N; i += LRH_SL_THREADS) A[i] = a.tmp[i];
    __syncthreads();
    int ja = a.first_inband / gp, jb;
    if (a.par2 == 0) {
      jb = 1 + a.last_inband / gp;
      if ((jb - ja) * gp > N) jb--;
      for (int i = ja * gp + tid; i < jb * gp; i += LRH_SL_THREADS) A[i] = a.yfac[i] * a.slowsum[i];
      __syncthreads();
      for (int j = ja + (tid >> 6); j < jb; j += LRH_SL_THREADS / 64) { const float m = sl_three_smallest_wave(A, j * gp, j * gp + gp, tid & 63); if ((tid & 63) == 0) G[j] = m; }
    } else {
      // running boundaries: group 0 ends at (ja+1) gp, the last one is cut at last_inband + 1
      jb = ja + 1;
      int ib_end = jb * gp;
      do { ib_end += gp; if (ib_end > a.last_inband) ib_end = a.last_inband + 1; jb++; } while (ib_end < a.last_inband);
      // (the group loop fills the spectrum up to the end of its last group, the tail loop behind it up to last_point - 1: sellim.c:925-983)
      for (int i = tid; i < (ib_end > a.last_point ? ib_end : a.last_point); i += LRH_SL_THREADS) A[i] = a.yfac[i] * a.slowsum[i];
      __syncthreads();
      for (int j = ja + (tid >> 6); j < jb; j += LRH_SL_THREADS / 64) {
        const int lo = j == ja ? a.first_inband : j * gp;
        int hi = (j + 1) * gp; if (j > ja && hi > a.last_inband) hi = a.last_inband + 1;
        const float m = sl_three_smallest_wave(A, lo, hi, tid & 63);
        if ((tid & 63) == 0) G[j] = m;
      }
    }
    __syncthreads();
    if (tid == 0) {
      float t1 = 0, t2;
      for (int j = ja; j < jb; j++) t1 += G[j];
      t1 /= jb - ja;
      int k = 0;
      float nf = 0;
      t1 *= (float)(2 * (1 + 2. / a.spek_avgnum));
      for (int j = ja; j < jb; j++) if (G[j] < t1) { nf += G[j]; k++; }
      if (a.par3 == 1) {
        t2 = (float)(0.05 * nf / k);
        int fg = ja, lg = jb;
        while (G[fg] < t2) fg++;
        while (G[lg - 1] < t2) lg--;
        if (fg != ja || lg != jb) { k = 0; nf = 0; for (int j = fg; j < lg; j++) if (G[j] < t1) { nf += G[j]; k++; } }
      }
      if (nf < 0.0001) nf = 0.0001f;
      if (k != 0) { nf *= (float)((1 + 2. / a.spek_avgnum) / k); nf *= a.ston; }
      s_nf = nf; s_k = k;
    }
    __syncthreads();
    stamp();
    const float nf = s_nf;
    build_bits(nf);
    __syncthreads();
    build_l2();
    // Every bin above the noise floor from bin 2 up to last_point - 1 ends up marked by the serial scan (as a member of a
    // run, of a skirt, or as the start of the next run): all threads mark them now; thread 0 then walks the runs only, for
    // the skirts below and above each run and the end of the band.
    if (s_k != 0) for (int i = 2 + tid; i < a.last_point; i += LRH_SL_THREADS) if (A[i] > nf && B[i] == 0) B[i] = -1;
    __syncthreads();
    if (tid == 0) {
      auto mark = [&](int i) { if (B[i] == 0) B[i] = -1; };
      int ia = N;                                          // k == 0 cannot happen (one group is always below twice the mean); the reference
      if (s_k != 0) {                                      // would then carry on with the group loop's leftover index, beyond the band
        ia = 0;
        while (ia < a.first_point || ia < 2) { mark(ia); ia++; }
        { const int e = next_clear(ia); for (int i = max(ia, a.last_point); i < e; i++) mark(i); ia = e; }   // while(tmp[ia] > nf && ia < N)
        const float t1 = a.par4 == 0 ? 4.F : 3.F;
        while (t1 * A[ia + 1] < A[ia] && ia < N) { ia++; mark(ia); }
        for (;;) {
          { int nh = next_set(ia); if (nh > a.last_point) nh = a.last_point; if (nh > ia) ia = nh; }   // while(tmp[ia] <= nf && ia < last) ia++
          if (ia >= a.last_point) break;
          int ib = ia;
          mark(ia);
          while ((2.f * A[ib - 1] < A[ib] || 4.f * A[ib - 2] < A[ib]) && ib > a.first_point) { ib--; mark(ib); }
          { int e = next_clear(ia + 1) - 1; if (e > a.last_point) e = a.last_point; if (e > ia) ia = e; mark(ia); }   // to the end of the run
          if (ia != a.last_point) {
            while ((2.f * A[ia + 1] < A[ia] || 4.f * A[ia + 2] < A[ia]) && ia < a.last_point) { ia++; mark(ia); }
            ia++;
          }
          if (ia >= a.last_point) break;
        }
      }
      if (ia > N - 2) ia = N - 2;
      s_ia = ia;
    }
    __syncthreads();
    stamp();
    const int ia_end = s_ia;
    for (int i = ia_end + tid; i < N; i += LRH_SL_THREADS) { if (a.par8 == 0) B[i] = -1; else if (B[i] == 0) B[i] = -1; }
    __syncthreads();
    // hold-off and slow release (sellim.c:1121-1147), per bin
    int k = (int)(1 + 1 / (a.avg1 * a.blocktime));
    const unsigned int wait_n = k < 255 ? (unsigned)k : 255u;
    // (all global loads first: the byte stores to a.wait may alias anything as far as the compiler knows, so a load-use-store loop
    // pays one memory round trip per iteration -- 24 us for 16 bins per thread against 3)
    constexpr int PT = (BIG ? 32768 : 16384) / LRH_SL_THREADS;   // N <= 16384, BIG: 32768 (launch_sellim)
    unsigned char w_[PT]; float o_[PT];
#pragma unroll
    for (int q = 0; q < PT; q++) { const int i = tid + q * LRH_SL_THREADS; if (i < N) { w_[q] = a.wait[i]; o_[q] = a.old_liminfo[i]; } }
#pragma unroll
    for (int q = 0; q < PT; q++) {
      const int i = tid + q * LRH_SL_THREADS;
      if (i >= N) continue;
      unsigned char w = w_[q];
      float l = B[i];
      if (l != 0) w = (unsigned char)wait_n;
      else { if (w > 0) w--; if (w > 0) l = -1; }
      const float o = o_[q];
      if (o > 0) { const float t1 = (float)(o * 1.15); if (t1 < 1) { if (l > 0 && l > t1) l = t1; } }
      a.wait[i] = w; B[i] = l;
      a.tmp[i] = A[i];
    }
    __syncthreads();
  }
  stamp();
  sl_selfreq(a, B, tid);
  __syncthreads();
  stamp();
  for (int i = tid; i < N; i += LRH_SL_THREADS) {
    a.old_liminfo[i] = B[i];
    const float v = (i < 2 || i >= N - 2) ? 0.f : B[i];      // sellim.c:1152-1155
    a.liminfo[i] = v; B[i] = v;
  }
  sl_pack(a, B, tid);
  stamp();
  if (a.debug && tid == 0) {
    printf("k_sellim ticks (10 ns):");
    for (int i = 1; i < nts; i++) printf(" %lld", ts[i] - ts[i - 1]);
    printf("  [load, pass1, groups+floor, pass2 scan, hold-off, selfreq, store]\n");
  }
}

// fft2_update_liminfo, sellim.c:159-736, hg.sellim_par1 = 2 (:535-731).  Per fft1 bin the mean of the summed fft2 power over the
// bin's width (A), group statistics -> global noise floor (thread 0 adds the groups in order: float sums), the neighbour fix-up
// next to strong bins (serial: it reads what it has just lowered), thinning of an overgrown table, marking of everything within two
// bins of power above 0.5 * ston * floor.  A persists between calls like the reference's fftf_tmp (zero outside the passband).
template <bool BIG>
__global__ __launch_bounds__(LRH_SL_THREADS) void k_sellim2(SellimArgs a)
{
  extern __shared__ float sm[];
  const int N = a.n, tid = threadIdx.x, nn = a.n2 / a.n, gp = a.group_points, groups = N / gp;
  float *A = sm + 8;
  float *B = BIG ? a.big_b : A + N + 16;
  float *reg_min = BIG ? A + N + 16 : B + N + 8, *reg_ston = reg_min + groups + 1, *reg_noise = reg_ston + groups + 1;   // 3 N/16 + 3 <= N/4 + 8
  __shared__ float s_t1; __shared__ int s_go, s_k;
  for (int i = tid; i < N; i += LRH_SL_THREADS) { A[i] = a.tmp[i]; B[i] = a.liminfo[i]; }
  for (int i = tid; i < 8; i += LRH_SL_THREADS) { A[-8 + i] = 0.f; A[N + i] = 0.f; A[N + 8 + i] = 0.f; }
  __syncthreads();
  for (int i = a.first_point + tid; i < a.last_point; i += LRH_SL_THREADS) {
    float t1 = 0;
    for (int j = nn * i; j < nn * i + nn; j++) t1 += a.powersum2[j];
    A[i] = t1 * a.yfac[i] / nn;
  }
  __syncthreads();
  for (int g = tid; g < groups; g += LRH_SL_THREADS) {
    const int ia = g * gp, ib = ia + gp;
    float t1 = 0;
    for (int j = ia; j < ib; j++) t1 += A[j];
    t1 /= gp;
    int k = 0; float t3 = LRH_SL_BIG, t2 = 0; t1 *= 0.001F;
    for (int j = ia; j < ib; j++) if (A[j] > t1) { k++; if (A[j] < t3) t3 = A[j]; if (A[j] > t2) t2 = A[j]; }
    reg_min[g] = k < 2 ? -1.f : t3;
    reg_ston[g] = t2 / t3;
  }
  __syncthreads();
  if (tid == 0) {
    s_go = 0;                                            // 0: return without selfreq (sellim.c:604); 1: fft2updx; 2: all the way
    float t1 = 0; int k = 0;
    for (int g = 0; g < groups; g++) if (reg_ston[g] < 2000.F) { t1 += reg_min[g]; k++; }
    if (k != 0) {
      s_go = 1;
      t1 /= k;
      float gnf = 0; k = 0;
      for (int g = 0; g < groups; g++) if (reg_min[g] > 0.03F * t1 && reg_min[g] < 30.F * t1) { gnf += reg_min[g]; k++; }
      if (k >= 3) { gnf /= k; s_t1 = 5 * gnf; s_go = 2; }
    }
  }
  __syncthreads();
  if (s_go == 2) {
    const float lim5 = s_t1;
    for (int g = tid; g < groups; g += LRH_SL_THREADS) {
      const int ia = g * gp, ib = ia + gp;
      float t2 = 0; int k = 0;
      for (int j = ia; j < ib; j++) if (A[j] < lim5) { k++; t2 += A[j]; }
      reg_noise[g] = k > 2 ? t2 / k : -1.f;
    }
    __syncthreads();
    if (tid == 0) {
      s_go = 1;
      float t1 = 0; int k = 0;
      for (int g = 0; g < groups; g++) if (reg_noise[g] > 0) { k++; t1 += reg_noise[g]; }
      if (k >= 3) {
        t1 /= k;
        float gnf = 0; k = 0;
        for (int g = 0; g < groups; g++) if (reg_noise[g] > 0.1F * t1 && reg_noise[g] < 10.F * t1) { gnf += reg_noise[g]; k++; }
        if (k >= 3) { gnf /= k; s_t1 = (float)(0.5 * a.ston2 * gnf); s_go = 2; }
      }
    }
    __syncthreads();
  }
  if (s_go == 2) {
    const float t1 = s_t1;
    int ia = a.first_point; if (ia < 2) ia = 2;
    int ib = a.last_point; if (ib < N - 2) ib = N - 2;   // as written (sellim.c:668-669)
    if (tid == 0) {
      int k = 0;
      for (int i = ia; i < ib; i++) {
        if (B[i] != 0) {
          if (B[i - 1] == 0 && A[i - 1] > A[i - 2]) A[i - 1] = A[i - 2];
          if (B[i + 1] == 0 && A[i + 1] > A[i + 2]) A[i + 1] = A[i + 2];
          k++;
        }
      }
      s_k = k;
    }
    __syncthreads();
    if (s_k > (ib - ia) / 4) {
      __syncthreads();
      if (tid == 0) s_k = 0;
      __syncthreads();
      int k = 0;
      for (int i = ia + tid; i < ib; i += LRH_SL_THREADS) { if (B[i] < 0 && A[i] < t1) { B[i] = 0; a.wait[i] = 0; } if (B[i] != 0) k++; }
      if (k) atomicAdd(&s_k, k);
      __syncthreads();
      if (s_k > (ib - ia) / 4) {
        const float t2 = 10.F * t1;
        for (int i = ia + tid; i < ib; i += LRH_SL_THREADS) if (B[i] < 0 && A[i] < t2) { B[i] = 0; a.wait[i] = 0; }
      }
      __syncthreads();
    }
    const unsigned wn = (unsigned)(1 + (1 + (a.blocktime2 * a.wf_avgnum)) / (a.avg1 * a.blocktime));
    const unsigned char wait_n = (unsigned char)(wn > 255u ? 255u : wn);
    for (int i = ia + tid; i < ib; i += LRH_SL_THREADS)               // the fifth term repeats i-2 in the reference (sellim.c:723)
      if (2. * A[i - 2] > t1 || A[i - 1] > t1 || A[i] > t1 || A[i + 1] > t1 || 2. * A[i - 2] > t1) {
        if (B[i] == 0) B[i] = -1;
        a.wait[i] = wait_n;
      }
    __syncthreads();
  }
  for (int i = tid; i < N; i += LRH_SL_THREADS) a.tmp[i] = A[i];
  if (s_go == 0) return;
  sl_selfreq(a, B, tid);
  __syncthreads();
  for (int i = tid; i < N; i += LRH_SL_THREADS) { a.old_liminfo[i] = B[i]; a.liminfo[i] = B[i]; }
  sl_pack(a, B, tid);
}

// fft2_update_liminfo with hg.sellim_par1 = 0 (sellim.c:170-281): the noise floor is the median of all fft2 bin powers (times the fft1
// bin's display factor).  The reference sorts the lower half of the spectrum by selection to read it off; here a radix select over the
// float bit patterns finds the same order statistic in four histogram passes of one workgroup.  Then the band edges (first / last fft2
// bin not below 2 % of the median), and every fft1 bin holding an fft2 bin above ston * median joins the strong signals.
template <bool BIG>
__global__ __launch_bounds__(LRH_SL_THREADS) void k_sellim2_median(SellimArgs a)
{
  extern __shared__ float sm[];
  const int N = a.n, N2 = a.n2, tid = threadIdx.x, nn = N2 / N;
  float *B = BIG ? a.big_b : sm;
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_rank; __shared__ int s_lo, s_hi;
  for (int i = tid; i < N; i += LRH_SL_THREADS) B[i] = a.liminfo[i];
  if (tid == 0) { s_prefix = 0; s_rank = (unsigned)(N2 / 2 - 1); s_lo = N2 - 1; s_hi = 0; }
  auto power = [&](int j) { return a.powersum2[j] * a.yfac[j / nn]; };
  auto key = [&](int j) { const unsigned u = __float_as_uint(power(j)); return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u); };   // unsigned order = float order
  for (int pass = 0; pass < 4; pass++) {
    const int sh = 24 - 8 * pass;
    for (int i = tid; i < 256; i += LRH_SL_THREADS) hist[i] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix, himask = pass == 0 ? 0u : 0xffffffffu << (sh + 8);
    for (int j = tid; j < N2; j += LRH_SL_THREADS) { const unsigned k = key(j); if ((k & himask) == prefix) atomicAdd(&hist[(k >> sh) & 255u], 1u); }
    __syncthreads();
    if (tid == 0) {
      unsigned r = s_rank, d = 0;
      while (d < 255 && r >= hist[d]) { r -= hist[d]; d++; }
      s_rank = r; s_prefix = prefix | (d << sh);
    }
    __syncthreads();
  }
  const unsigned mk = s_prefix;
  const float median = __uint_as_float(mk ^ ((mk >> 31) ? 0x80000000u : 0xffffffffu));
  const float edge = median * 0.02F;
  int ib0 = (nn + 1) * a.last_point; if (ib0 > N2) ib0 = N2;          // as written (sellim.c:239)
  {
    int lo = N2 - 1, hi = 0;                                          // hi: last bin (+1) below ib0 that is not under the edge limit; 1 at least
    for (int j = tid; j < N2; j += LRH_SL_THREADS) {
      const bool up = !(power(j) < edge);
      if (up && j >= nn * a.first_point && j < lo) lo = j;
      if (up && j < ib0 && j + 1 > hi) hi = j + 1;
    }
    atomicMin(&s_lo, lo); atomicMax(&s_hi, hi);
  }
  __syncthreads();
  int first = (s_lo + nn / 2) / nn, last = ((s_hi < 1 ? 1 : s_hi) + nn / 2) / nn;
  if (first < 5) first = 5;
  if (last > N - 6) last = N - 6;
  const unsigned wn = (unsigned)(1 + (1 + (a.blocktime2 * a.wf_avgnum)) / (a.avg1 * a.blocktime));
  const unsigned char wait_n = (unsigned char)(wn > 255u ? 255u : wn);
  const float limit = a.ston2 * median;
  for (int i = tid; i < N; i += LRH_SL_THREADS) {
    bool strong = i < first || i >= last;
    if (!strong) for (int j = nn * i; j < nn * i + nn; j++) strong |= power(j) > limit;
    if (strong) { B[i] = -1; a.wait[i] = wait_n; }
  }
  __syncthreads();
  sl_selfreq(a, B, tid);
  __syncthreads();
  for (int i = tid; i < N; i += LRH_SL_THREADS) { a.old_liminfo[i] = B[i]; a.liminfo[i] = B[i]; }
  sl_pack(a, B, tid);
}

// fft2_update_liminfo with hg.sellim_par1 = 1 (sellim.c:283-533).  The attenuated carriers (liminfo > 0) cut the band into weak-signal
// regions; thread 0 walks them through a bit map of the carriers (64 bins a step), and per region of six bins or more the workgroup
// forms the fft2 power of every bin, thread 0 adds the bins under the limit in index order (a float sum: the order is the result), the
// workgroup marks what stands ston above that floor, thread 0 keeps the region list.  After the walk: regions, then single bins, above
// ston times the length-weighted mean floor.  The list (noise, first point, length) persists like the reference's arrays, and the
// clean-up of a nearly full list keeps the reference's indexing (sellim.c:424, 447-460).
template <bool BIG>
__global__ __launch_bounds__(LRH_SL_THREADS) void k_sellim2_regions(SellimArgs a)
{
  extern __shared__ float sm[];
  const int N = a.n, tid = threadIdx.x, nn = a.n2 / a.n, G = N / a.group_points, last = a.last_point;
  float *A = sm + 8;
  float *B = BIG ? a.big_b : A + N + 16;
  float *reg_noise = BIG ? A + N + 16 : B + N + 8;
  int *reg_first = (int *)(reg_noise + G + 8), *reg_len = reg_first + G + 8;
  unsigned long long *carrier = (unsigned long long *)(reg_len + G + 8 + ((G & 1) ? 1 : 0));      // bit i: liminfo[i] > 0
  __shared__ int s_ia, s_ib, s_state, s_regs, s_first, s_last; __shared__ unsigned s_low; __shared__ float s_over;
  const int nwords = (N + 63) / 64;
  long long tph[6] = { 0, 0, 0, 0, 0, 0 }, tlast = 0; int nreg_dbg = 0;             // LRH_SELLIM_DEBUG: 100 MHz ticks per phase
  auto phase = [&](int k) { if (a.debug && tid == 0) { const long long t = wall_clock64(); if (k >= 0) tph[k] += t - tlast; tlast = t; } };
  phase(-1);
  for (int i = tid; i < N; i += LRH_SL_THREADS) { A[i] = a.tmp[i]; B[i] = a.liminfo[i]; }
  for (int i = tid; i < 8; i += LRH_SL_THREADS) { A[-8 + i] = 0.f; A[N + i] = 0.f; A[N + 8 + i] = 0.f; }
  for (int i = tid; i < G + 8; i += LRH_SL_THREADS) { reg_noise[i] = a.reg_noise[i]; reg_first[i] = a.reg_first[i]; reg_len[i] = a.reg_len[i]; }
  __syncthreads();
  for (int i0 = (tid / 64) * 64; i0 < nwords * 64; i0 += LRH_SL_THREADS) {
    const int i = i0 + (tid & 63);
    const unsigned long long w = __ballot(i < N && B[i] > 0);
    if ((tid & 63) == 0) carrier[i0 / 64] = w;
  }
  const unsigned wn = (unsigned)(1 + (1 + (a.blocktime2 * a.wf_avgnum)) / (a.avg1 * a.blocktime));
  const unsigned char wait_n = (unsigned char)(wn > 255u ? 255u : wn);
  const float ston = a.ston2;
  // first bin >= i (and < end) whose carrier bit equals `want`; end if none
  auto next_bit = [&](int i, int end, bool want) {
    while (i < end) {
      unsigned long long w = carrier[i >> 6]; if (!want) w = ~w;
      w &= ~0ull << (i & 63);
      if (w) { const int j = (i & ~63) + __ffsll((long long)w) - 1; return j < end ? j : end; }
      i = (i & ~63) + 64;
    }
    return end;
  };
  auto mark0 = [&](int b) { B[b] = -1; a.wait[b] = wait_n; carrier[b >> 6] &= ~(1ull << (b & 63)); };      // thread 0, any bin
  auto drop = [&](int k, int n) { for (int j = k + 1; j < n; j++) { reg_noise[j - 1] = reg_noise[j]; reg_first[j - 1] = reg_first[j]; reg_len[j - 1] = reg_len[j]; } };
  auto mean_noise = [&](int n, bool skip_negative) {
    int k = 0; float t = 0;
    for (int i = 0; i < n; i++) { if (skip_negative && reg_noise[i] < 0) continue; k += reg_len[i]; t += reg_noise[i] * reg_len[i]; }
    return t / k;
  };
  if (tid == 0) { s_ia = a.first_point; s_regs = 0; }
  __syncthreads();
  phase(0);
  for (;;) {
    if (tid == 0) {                                       // the next region of six bins or more, or the end of the band
      int ia = s_ia, ib = 0, state = 0;
      for (;;) {
        ia = next_bit(ia, last, false);                   // while (liminfo[ia] > 0 && ia < last) ia++
        if (ia >= last) break;
        ib = next_bit(ia, last, true);                    // while (liminfo[ib] <= 0 && ib < last) ib++
        if (ib - ia >= 6) { state = 1; break; }
        ia = ib;
      }
      s_state = state; s_ia = ia + 1; s_ib = ib - 1; s_low = __float_as_uint(LRH_SL_BIG); s_first = N; s_last = -1;
    }
    __syncthreads();
    phase(1);
    if (!s_state) break;
    nreg_dbg++;
    const int ia = s_ia, ib = s_ib;
    {
      float lowest = LRH_SL_BIG;
      for (int i = ia + tid; i < ib; i += LRH_SL_THREADS) {
        float t = 0;
        for (int j = nn * i; j < nn * i + nn; j++) t += a.powersum2[j];
        t *= a.yfac[i];
        A[i] = t;
        if (t < lowest && i >= a.first_inband && i <= a.last_inband) lowest = t;
      }
      if (lowest < LRH_SL_BIG) atomicMin(&s_low, __float_as_uint(lowest < 0 ? 0.f : lowest));     // powers: the bit patterns order like the values
    }
    __syncthreads();
    phase(2);
    if (tid < 64) {
      // the bins under the limit, added in index order: a float sum that decides a threshold, so the order is part of the result.  One
      // wave: the lanes load and test 64 bins at once, the sum then takes them lane by lane (x + 0 = x leaves it as it was)
      if (tid == 0) { A[ia - 1] = A[ia]; A[ib] = A[ib - 1]; }
      float limit = __uint_as_float(s_low);
      limit *= 2 * (1 + 2. / a.wf_avgnum);
      const int ja = ia < a.first_inband ? a.first_inband : ia, jb = ib > a.last_inband ? a.last_inband + 1 : ib;
      float sum = 0; int cnt = 0;                          // not cleared when the limit is widened (sellim.c:356-371)
      for (;;) {
        for (int base = ja; base < jb; base += 64) {
          const int i = base + tid;
          const float v = i < jb ? A[i] : LRH_SL_BIG;
          const bool in = i < jb && v < limit;
          const unsigned long long m = __ballot(in);
          if (!m) continue;
          cnt += __popcll(m);
          const int mv = __float_as_int(in ? v : 0.f);
          if (__popcll(m) < 20) {                            // few: only those
            for (unsigned long long r = m; r; r &= r - 1) sum += __int_as_float(__builtin_amdgcn_readlane(mv, __ffsll((long long)r) - 1));
          } else {
#pragma unroll
            for (int l = 0; l < 64; l++) sum += __int_as_float(__builtin_amdgcn_readlane(mv, l));
          }
        }
        if (cnt == 0 || cnt >= (jb - ja) / 4) break;
        limit *= 3;
      }
      if (tid == 0) {
        s_state = cnt ? 1 : 2;                             // 2: nothing under the limit, the region is passed over
        if (cnt) { const float fl = sum / cnt; reg_noise[s_regs] = fl; reg_first[s_regs] = ia - 1; s_over = fl * ston; }
      }
    }
    __syncthreads();
    phase(3);
    if (s_state == 1) {
      const float over = s_over;
      int f0 = N, f1 = -1;
      for (int i = ia + tid; i < ib; i += LRH_SL_THREADS) if (A[i] > over) { if (i < f0) f0 = i; if (i > f1) f1 = i; B[i] = -1; a.wait[i] = wait_n; }
      if (f1 >= 0) { atomicMin(&s_first, f0); atomicMax(&s_last, f1); }
      __syncthreads();
      if (tid == 0) {
        int n = s_regs;
        const float fl = reg_noise[n];
        if (s_last < 0) reg_len[n++] = ib - ia + 1;
        else {                                             // the quiet parts below the first and above the last bin taken out
          reg_len[n++] = s_first - ia + 1;
          if (ib - s_last > 4) { reg_noise[n] = fl; reg_first[n] = s_last + 1; reg_len[n] = ib - s_last; n++; }
        }
        if (n >= G - 2) {                                  // the list is nearly full (sellim.c:406-469)
          float t1 = mean_noise(n, false) * ston;
          for (int k = 0; k < n; k++)
            if (reg_noise[k] > t1) {
              for (int i = 0; i < G + 8 && i < reg_len[i]; i++) { const int b = i + reg_first[k]; if (b >= 0 && b < N) mark0(b); }   // entry i's own length ends it
              drop(k, n);
              n--;                                         // k moves on past the entry that slid into place
            }
          if (n >= 3 * G / 4) {
            t1 /= ston;
            while (n > 0) { if (reg_noise[0] < t1) drop(0, n); n--; }      // the count drops every pass (sellim.c:447-460)
          }
        }
        s_regs = n;
      }
    }
    if (tid == 0) s_ia = ib + 1;
    __syncthreads();
    phase(4);
  }
  const int regs0 = s_regs;
  if (regs0 > 0) {
    if (tid == 0) {
      const float t1 = mean_noise(regs0, false) * ston;
      int dropped = 0;
      for (int k = 0; k < regs0; k++) if (reg_noise[k] > t1) { dropped = 1; reg_noise[k] = -1; }
      s_state = dropped; s_over = t1;
    }
    __syncthreads();
    if (s_state) {
      for (int k = 0; k < regs0; k++)                      // a whole region above the common floor
        if (reg_noise[k] < 0) for (int i = tid; i < reg_len[k]; i += LRH_SL_THREADS) { B[i + reg_first[k]] = -1; a.wait[i + reg_first[k]] = wait_n; }
      __syncthreads();
      if (tid == 0) {
        float t1 = mean_noise(regs0, true);
        int n = regs0;
        for (int i = 0; i < n; i++) if (reg_noise[i] < 0) { drop(i, n); i--; n--; }
        s_over = t1 * ston; s_regs = n;
      }
      __syncthreads();
    }
    const float t1 = s_over; const int n = s_regs;
    for (int k = 0; k < n; k++)
      for (int i = tid; i < reg_len[k]; i += LRH_SL_THREADS) { const int b = i + reg_first[k]; if (A[b] > t1) { B[b] = -1; a.wait[b] = wait_n; } }
    __syncthreads();
  }
  for (int i = tid; i < N; i += LRH_SL_THREADS) a.tmp[i] = A[i];
  for (int i = tid; i < G + 8; i += LRH_SL_THREADS) { a.reg_noise[i] = reg_noise[i]; a.reg_first[i] = reg_first[i]; a.reg_len[i] = reg_len[i]; }
  sl_selfreq(a, B, tid);
  __syncthreads();
  for (int i = tid; i < N; i += LRH_SL_THREADS) { a.old_liminfo[i] = B[i]; a.liminfo[i] = B[i]; }
  sl_pack(a, B, tid);
  phase(5);
  if (a.debug && tid == 0)
    printf("k_sellim2_regions ticks (10 ns): %lld %lld %lld %lld %lld %lld  [load, walk, bin power, ordered sum, mark + list, tail]; %d regions, %d listed\n",
           tph[0], tph[1], tph[2], tph[3], tph[4], tph[5], nreg_dbg, regs0);
}

hipError_t launch_sellim(const SellimArgs &a, hipStream_t st)
{
  const bool big = a.n > 16384;
  const size_t words = 2 * sizeof(int) * ((a.n + 31) / 32 + 4) + 2 * sizeof(int) * ((a.n + 31) / 1024 + 2);
  const size_t lds = big ? sizeof(float) * (size_t)(8 + a.n + 16) + words : sizeof(float) * (size_t)(8 + a.n + 16 + a.n + 8 + a.n / 4 + 8) + words;
  static bool once = false;
  if (!once) { hipFuncSetAttribute((const void *)k_sellim<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
               hipFuncSetAttribute((const void *)k_sellim<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); once = true; }
  if (lds > 160 * 1024 - 64 || a.n > 32768 || (big && (!a.big_b || !a.big_g))) return hipErrorInvalidValue;
  if (a.r0 < (big ? 0 : 1)) return hipErrorInvalidValue;
  if (big) hipLaunchKernelGGL(k_sellim<true>, dim3(1), dim3(LRH_SL_THREADS), lds, st, a);
  else hipLaunchKernelGGL(k_sellim<false>, dim3(1), dim3(LRH_SL_THREADS), lds, st, a);
  return hipGetLastError();
}
hipError_t launch_sellim2(const SellimArgs &a, hipStream_t st)
{
  const bool big = a.n > 16384;
  const size_t lds = big ? sizeof(float) * (size_t)(8 + a.n + 16 + 3 * (a.n / (a.group_points > 0 ? a.group_points : 1) + 1) + 8)
                         : sizeof(float) * (size_t)(8 + a.n + 16 + a.n + 8 + a.n / 4 + 8) + sizeof(int) * ((a.n + 31) / 32 + 4);
  static bool once = false;
  if (!once) { hipFuncSetAttribute((const void *)k_sellim2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
               hipFuncSetAttribute((const void *)k_sellim2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); once = true; }
  static bool once2 = false;
  if (!once2) {
    hipFuncSetAttribute((const void *)k_sellim2_median<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);   // 1 KiB of histogram is static
    hipFuncSetAttribute((const void *)k_sellim2_median<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    hipFuncSetAttribute((const void *)k_sellim2_regions<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    hipFuncSetAttribute((const void *)k_sellim2_regions<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    if (hipGetLastError() != hipSuccess) return hipErrorInvalidValue;
    once2 = true;
  }
  if (a.group_points < 16 || a.n > 32768 || (big && !a.big_b)) return hipErrorInvalidValue;
  if (a.r0 < (big ? 0 : 1)) return hipErrorInvalidValue;
  const int groups = a.n / a.group_points;
  // the packing step's words sit behind the table in every variant (sl_pack)
  if (a.par1 == 0) {
    const size_t l0 = (big ? 64 : sizeof(float) * (size_t)(a.n + 8)) + sizeof(int) * ((a.n + 31) / 32 + 4);
    if (a.n2 < a.n || a.n2 % a.n || l0 > 158 * 1024) return hipErrorInvalidValue;
    if (big) hipLaunchKernelGGL(k_sellim2_median<true>, dim3(1), dim3(LRH_SL_THREADS), l0, st, a);
    else hipLaunchKernelGGL(k_sellim2_median<false>, dim3(1), dim3(LRH_SL_THREADS), l0, st, a);
    return hipGetLastError();
  }
  if (a.par1 == 1) {
    const size_t l1 = sizeof(float) * (size_t)(8 + a.n + 16 + (big ? 0 : a.n + 8) + 3 * (groups + 8) + 2) + 8 * (size_t)((a.n + 63) / 64 + 1)
                      + sizeof(int) * ((a.n + 31) / 32 + 4);
    if (l1 > 160 * 1024 - 256 || !a.reg_noise || !a.reg_first || !a.reg_len) return hipErrorInvalidValue;
    if (big) hipLaunchKernelGGL(k_sellim2_regions<true>, dim3(1), dim3(LRH_SL_THREADS), l1, st, a);
    else hipLaunchKernelGGL(k_sellim2_regions<false>, dim3(1), dim3(LRH_SL_THREADS), l1, st, a);
    return hipGetLastError();
  }
  if (lds > 160 * 1024 - 64) return hipErrorInvalidValue;
  if (big) hipLaunchKernelGGL(k_sellim2<true>, dim3(1), dim3(LRH_SL_THREADS), lds, st, a);
  else hipLaunchKernelGGL(k_sellim2<false>, dim3(1), dim3(LRH_SL_THREADS), lds, st, a);
  return hipGetLastError();
}
}  // namespace lrh

// =====================================================================================================
// spur subtraction (eliminate_spurs, spur.c:36-494; one RF channel, float spectra)
// =====================================================================================================
// A tracked spur is a phase-locked loop over the newest n = spur_speknum transforms of its seven bins: per transform the new
// bins are projected on the line shape of the predicted frequency, the loop re-estimates phase / frequency / drift / amplitude
// from the whole history, and amplitude x line shape at the predicted phase leaves the new transform.  Only the order of the
// transforms is serial (the state after transform t feeds t + 1).  One WAVE per spur: lane m works on the history entry of age
// m (ring slot na - m; n > 64: lanes take m, m + 64, ...).  What the reference does with running rotations and running sums over
// the history is closed form per lane here (the oscillator's angle at age m is a quadratic in m, the smoothing a centred window,
// the phase track a wave scan), and every sum over the history is a wave reduction; threshold decisions see the same quantities
// to float32 rounding (goldens: loop state after every transform, spectra to 1e-5).
namespace lrh {
struct DevSpur { int location, flag; float freq, d0pha, d1pha, d2pha, ampl, noise, avgd2; };   // = lrh_spur
#define LRH_PI 3.1415926535897932

// Cross-lane steps as DPP operands of the VALU (a few cycles each) instead of __shfl's ds_bpermute (an LDS round trip, ~100 cycles):
// one look at the history is a chain of ~60 such steps (8.5 -> 6.5 us per transform and spur at spur_speknum 16; what remains is the chain
// itself: ~12 LDS hand-overs between the lanes, eight single-precision sincos / atan2 and the reductions, twice per transform).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false)); }   // 0 where the source lane does not exist
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }   // l wave-uniform
__device__ __forceinline__ float up1(float v) { return dpp_f<0x138>(v); }            // wave_shr:1 -- lane l gets lane l - 1's value (lane 0: 0)
__device__ __forceinline__ float wsum(float v)
{
  v += dpp_f<0xb1>(v);                                     // quad_perm [1,0,3,2]
  v += dpp_f<0x4e>(v);                                     // quad_perm [2,3,0,1]
  v += dpp_f<0x124>(v);                                    // row_ror:4
  v += dpp_f<0x128>(v);                                    // row_ror:8: every lane holds the sum of its row of sixteen
  return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}
__device__ __forceinline__ float2 wsum2(float2 v) { return make_float2(wsum(v.x), wsum(v.y)); }
__device__ __forceinline__ int wmax(int v)
{
  v = max(v, dpp_i<0xb1>(v)); v = max(v, dpp_i<0x4e>(v)); v = max(v, dpp_i<0x124>(v)); v = max(v, dpp_i<0x128>(v));     // values >= 0: the 0 of a missing lane is harmless
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// inclusive prefix sums over the lanes: within the rows of sixteen by row_shr 1, 2, 4, 8, then the totals of the rows before
__device__ __forceinline__ float wscan(float v, int lane)
{
  v += dpp_f<0x111>(v); v += dpp_f<0x112>(v); v += dpp_f<0x114>(v); v += dpp_f<0x118>(v);
  const float t0 = lane_f(v, 15), t1 = lane_f(v, 31), t2 = lane_f(v, 47);
  const int row = lane >> 4;
  return v + (row >= 1 ? t0 : 0.f) + (row >= 2 ? t1 : 0.f) + (row >= 3 ? t2 : 0.f);
}
__device__ __forceinline__ int wscan(int v, int lane)
{
  v += dpp_i<0x111>(v); v += dpp_i<0x112>(v); v += dpp_i<0x114>(v); v += dpp_i<0x118>(v);
  const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
  const int row = lane >> 4;
  return v + (row >= 1 ? t0 : 0) + (row >= 2 ? t1 : 0) + (row >= 3 ? t2 : 0);
}
// The reference calls the C library in double on float data and rounds the results to float.  Here the angle is reduced to (-pi, pi] in
// double (three operations) and everything else runs in single precision: results agree to float rounding, and a lone wave -- which is
// what a spur's loop is -- would spend a good part of its time in the double versions (~150 dependent instructions each)
__device__ __forceinline__ void sincos_red(double ang, float &sn, float &cs)
{
  double t = ang * (0.5 / LRH_PI);
  t -= rint(t);
  sincosf((float)(t * (2 * LRH_PI)), &sn, &cs);
}
__device__ __forceinline__ float2 rot(float2 z, double ang)          // z e^{j ang}
{
  float s_, c; sincos_red(ang, s_, c);
  return make_float2(c * z.x - s_ * z.y, c * z.y + s_ * z.x);
}
__device__ __forceinline__ float2 unit_of(float2 z, float floor_)    // z / |z|, zero below the floor
{
  const float r = sqrtf(z.x * z.x + z.y * z.y);
  return r > floor_ ? make_float2(z.x / r, z.y / r) : make_float2(0.f, 0.f);
}
__device__ __forceinline__ float2 mulc(float2 a_, float2 b) { return make_float2(a_.x * b.x + a_.y * b.y, a_.y * b.x - a_.x * b.y); }   // a conj(b)

// working set of one spur in LDS: h = history de-rotated by the loop's oscillator (index 0 oldest .. n-1 newest), d / g = step between
// neighbours and its smoothed form (n - 1 values), t = phase track (n values), then scratch for the fits
struct SpurWork { float2 *h, *d, *g; float *t; float *row; };   // row: the newest transform's seven bins (the lanes that fetched them are not the lane that projects them)

struct SpurWave {
  const SpurArgs &a; DevSpur &q; float *tab, *zsig; int *uind; SpurWork w; const int lane, n, maxn, mask; const float *spec;   // tab / zsig / uind / spec: the tables in global memory (acquisition) or their copies in LDS (tracking), ring of maxn = mask + 1 slots

  __device__ int slot(int na, int age) const { return (na - age) & mask; }
  // line shape for a frequency: offset into the table of shapes and the half-bin position j (0 / 1); -1 when the carrier has left the
  // window of seven bins (then j tells which way, raw)
  __device__ static int shape(float freq, int loc, int &j)
  {
    j = (int)(freq) + 2 - loc - 4;
    if (j < 0 || j > 1) return -1;
    j = 1 - j;
    int ind = (int)(256 * (freq - (int)(freq)));
    if (ind == 256) ind = 255;
    return ind * 8 + j;
  }
  // the bin frequency the loop's phase slope stands for: the fractional part comes from the slope, the integer part stays nearest to `near`
  __device__ float freq_of(float slope, float near) const
  {
    float r = (float)(-0.5 * slope / LRH_PI);
    const int i = (int)(near * a.freq_factor - r + 0.5);
    r += i;
    return r / a.freq_factor;
  }
  // seven bins x line shape, sign by the parity of window position and sub-position
  __device__ float2 project(const float *bins, int ind, int j) const
  {
    float2 p = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 7; i++) { const float sh = spec[ind + i]; p.x += bins[2 * i] * sh; p.y += bins[2 * i + 1] * sh; }
    if ((j ^ (q.location & 1)) == 1) { p.x = -p.x; p.y = -p.y; }
    return p;
  }
  // the window of seven bins moves by one bin: every history row follows (shift_spur_table's job); false: the spur ran off the spectrum
  __device__ bool move_window(int dir, int na)
  {
    q.location += dir;
    if (q.location < 7) { q.flag = 1; q.location = 14; return false; }
    if (q.location > a.n2 - 7) { q.flag = 1; q.location = a.n2 - 14; return false; }
    for (int m = lane; m <= n; m += 64) {                  // ages 0 .. n: the n history rows and the one about to leave it
      float *r = tab + slot(na, m) * 14;
      float v[14];
#pragma unroll
      for (int i = 0; i < 14; i++) v[i] = r[i];
#pragma unroll
      for (int i = 0; i < 7; i++) {
        const int src = i + dir;
        r[2 * i] = (src >= 0 && src < 7) ? v[2 * src] : 0.f; r[2 * i + 1] = (src >= 0 && src < 7) ? v[2 * src + 1] : 0.f;
      }
    }
    if (lane == 0) {                                       // the copy of the newest row
      float v[14];
#pragma unroll
      for (int i = 0; i < 14; i++) v[i] = w.row[i];
#pragma unroll
      for (int i = 0; i < 7; i++) {
        const int src = i + dir;
        w.row[2 * i] = (src >= 0 && src < 7) ? v[2 * src] : 0.f; w.row[2 * i + 1] = (src >= 0 && src < 7) ? v[2 * src + 1] : 0.f;
      }
    }
    __builtin_amdgcn_wave_barrier();
    return true;
  }
  __device__ bool centre(float freq, int na, int &ind, int &j)
  {
    for (;;) {
      ind = shape(freq, q.location, j);
      if (ind >= 0) return true;
      if (j < -1 || j > 2) { q.flag = 1; return false; }
      if (!move_window(j < 0 ? -1 : 1, na)) return false;
    }
  }

  // one look at the history through the loop's oscillator: corrections (c0, c1, c2) to phase, slope and curvature; amplitude and noise
  // (ph0, sl0, cv0): the oscillator's phase at the newest entry `na`, its step and the step's change per transform
  __device__ void estimate(int na, double ph0, double sl0, double cv0, float &c0, float &c1, float &c2)
  {
    const int ns = n - 1, av = a.avgnum;
    // history seen from the oscillator: age m is turned back by phase - m slope + m (m - 1) / 2 curvature
    for (int m = lane; m < n; m += 64) {
      const float2 z = make_float2(zsig[2 * slot(na, m)], zsig[2 * slot(na, m) + 1]);
      w.h[ns - m] = rot(z, -(ph0 - m * sl0 + 0.5 * m * (m - 1) * cv0));
    }
    __builtin_amdgcn_wave_barrier();
    // step from each entry to the next, amplitude divided out
    for (int i = lane; i < ns; i += 64) w.d[i] = unit_of(mulc(w.h[i + 1], w.h[i]), 0.000000001f);
    __builtin_amdgcn_wave_barrier();
    // centred window mean, the ends held (window: av made odd, shrunk by two when that would exceed av and a quarter of the history)
    int wd = av | 1;
    if (wd > av && wd > ns / 4) wd -= 2;
    if (wd >= 1) {
      const int hw = wd / 2; const float inv = (float)(1.0 / wd);
      for (int i = lane; i < ns; i += 64) {
        const int cidx = min(max(i, hw), ns - 1 - hw);
        float2 sacc = make_float2(0.f, 0.f);
        for (int k = cidx - hw; k <= cidx + hw; k++) { sacc.x += w.d[k].x; sacc.y += w.d[k].y; }
        w.g[i] = make_float2(sacc.x * inv, sacc.y * inv);
      }
    }
    __builtin_amdgcn_wave_barrier();
    // curvature: mean direction of the change of the smoothed step
    float2 bend = make_float2(0.f, 0.f);
    for (int i = 1 + av / 2 + lane; i < ns - av / 2; i += 64) { const float2 u = unit_of(mulc(w.g[i], w.g[i - 1]), 0.00001f); bend.x += u.x; bend.y += u.y; }
    bend = wsum2(bend);
    c2 = (float)atan2f(bend.y, bend.x);
    {
      float tot = q.d2pha + c2;
      if (fabsf(tot) > a.max_d2 && fabsf(c2) > a.max_d2) c2 = -q.d2pha / n;    // implausible: pull the drift back instead
      else {
        tot = a.weiold * q.avgd2 + a.weinew * tot;
        float trust = 1.f;                                // strong spurs follow the averaged drift, weak ones the new estimate
        if (q.noise > 0.000001 && fabsf(q.ampl) > 0.000001) { trust = 0.1f * fabsf(q.ampl) / q.noise; trust = 1 / (1 + trust); }
        c2 = trust * (tot - q.d2pha) + (1 - trust) * c2;
      }
    }
    // phase track relative to the newest entry: unwrapped step angles summed from the new end, the curvature just found taken out
    for (int base = 0, carry = 0; base < ns; base += 64) {              // unwrap: running count of 2 pi jumps between neighbours
      const int i = base + lane;
      const float cur = i < ns ? (float)atan2f(w.g[i].y, w.g[i].x) : 0.f;
      float prev = up1(cur);
      if (lane == 0) prev = base ? w.t[ns] : cur;                        // w.t[ns]: last raw angle of the previous chunk (parked below)
      int jump = 0;
      if (i < ns && i > 0) { if (cur - prev > LRH_PI) jump = -1; else if (cur - prev < -LRH_PI) jump = 1; }
      const int incl = wscan(jump, lane);
      if (i < ns) w.t[i] = cur + (float)(2 * LRH_PI) * (float)(carry + incl);
      carry += __builtin_amdgcn_readlane(incl, 63);
      const float last_raw = lane_f(cur, 63);
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) w.t[ns] = last_raw;
      __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_wave_barrier();
    {                                                                     // suffix sums from the new end: track[k] = -sum_{m >= k} step[m]
      float carry = 0.f;
      const int nchunk = (ns + 63) / 64;
      for (int ch = 0; ch < nchunk; ch++) {
        const int i = ns - 1 - (ch * 64 + lane);                          // lane 0 takes the newest
        const float v = i >= 0 ? w.t[i] : 0.f;
        const float incl = wscan(v, lane);
        const float tot = lane_f(incl, 63);
        __builtin_amdgcn_wave_barrier();
        if (i >= 0) { const int age = ns - i; w.t[i] = -(carry + incl) - (age >= 2 ? age * (age - 1) * (float)(c2 * 0.5) : 0.f); }
        carry += tot;
      }
      __builtin_amdgcn_wave_barrier();
    }
    // slope: difference of the means of the two halves of the newer part of the track
    int len = n - av;
    if (len < 10) len = n - av / 2;
    if (len < 3) len = n;
    {
      const int k = len / 2, first = n - len;                             // track index n - 1 is the newest entry (value 0)
      float lo = 0.f, hi = 0.f;
      for (int i = lane; i < k; i += 64) { lo += (first + i < ns ? w.t[first + i] : 0.f); hi += (first + k + i < ns ? w.t[first + k + i] : 0.f); }
      c1 = (wsum(hi) - wsum(lo)) / (k * k);
    }
    // what is left after slope and curvature is the phase offset: mean direction of the history turned by them
    float2 dir = make_float2(0.f, 0.f);
    for (int m = lane; m < n; m += 64) {
      const int i = ns - m;
      const float2 r = rot(w.h[i], -((1.0 - m) * (double)c1 + 0.5 * m * (m - 1) * (double)c2));
      w.g[i] = r;                                                          // (the smoothed steps are done with; n slots)
      const float2 u = unit_of(r, 0.f);
      dir.x += u.x; dir.y += u.y;
    }
    dir = wsum2(dir);
    c0 = (float)atan2f(dir.y, dir.x);
    { const float nrm = sqrtf(dir.x * dir.x + dir.y * dir.y); dir.x /= nrm; dir.y /= nrm; }
    __builtin_amdgcn_wave_barrier();
    // straight-line fit of the residual phase across the history: a last correction of the slope
    float fit = 0.f;
    for (int i = lane; i < n; i += 64) {
      const float2 r = mulc(w.g[i], dir);
      w.g[i] = r;
      const float x = (float)(-0.5 * ns) + i;
      if (r.x > 0 && fabsf(r.y) < fabsf(r.x)) fit += x * r.y / fabsf(r.x);
      else fit += x * atan2f(r.y, r.x);
    }
    const float tilt = wsum(fit) / a.linefit;
    c1 += tilt;
    __builtin_amdgcn_wave_barrier();
    // amplitude = mean in-phase part with the tilt removed, noise = rms of what remains
    float inph = 0.f;
    for (int i = lane; i < n; i += 64) { const float2 r = rot(w.g[i], ((double)i - 0.5 * ns) * (double)tilt); w.g[i] = r; inph += r.x; }
    const float ampl = wsum(inph) / n;
    __builtin_amdgcn_wave_barrier();
    float res = 0.f;
    for (int i = lane; i < n; i += 64) { const float2 r = w.g[i]; res += (r.x - ampl) * (r.x - ampl) + r.y * r.y; }
    q.ampl = ampl;
    q.noise = sqrtf(wsum(res) / n);
    __builtin_amdgcn_wave_barrier();
  }
  // the loop takes the corrections: they were found against the oscillator one step ahead
  __device__ void refine(int na)
  {
    float c0, c1, c2;
    float slope = q.d1pha + q.d2pha, phase = q.d0pha + slope, curv = q.d2pha;
    estimate(na, (double)phase, (double)slope, (double)curv, c0, c1, c2);
    phase += c0; slope += c1; curv += c2;
    phase -= slope; slope -= curv;
    q.d0pha = phase; q.d1pha = slope; q.d2pha = curv;
  }


  // ---- acquisition (store_new_spur + spur_phase_lock with verify_spur_pll, spursub.c:619-751, 1247-1843): the carrier in the seven
  // bins from `pnt` is taken into the history, its frequency read off the summed power of the newest n transforms, and the loop is
  // closed on the history in up to five rounds; accepted when the corrections have died down and what the subtraction would leave
  // behind is spectrally flat.  `na` = ring slot behind the newest transform.  Same one-wave form as the tracking.
  __device__ bool acquire(int na, int pnt)
  {
    const int newest = (na - 1) & mask;
    float pwr7[7];
#pragma unroll
    for (int i = 0; i < 7; i++) pwr7[i] = 0.f;
    for (int m = lane; m < n; m += 64) {                   // rows of the history and the power of each bin
      const float2 *z = a.fft2 + (size_t)slot(newest, m) * a.n2 + pnt;
      float *r = tab + slot(newest, m) * 14;
#pragma unroll
      for (int i = 0; i < 7; i++) { const float2 v = z[i]; r[2 * i] = v.x; r[2 * i + 1] = v.y; pwr7[i] += v.x * v.x + v.y * v.y; }
    }
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 7; i++) pwr7[i] = wsum(pwr7[i]);
    { const float base = 0.5f * (pwr7[0] + pwr7[6]);       // the ends of the window stand for the noise floor; unit sum
#pragma unroll
      for (int i = 0; i < 7; i++) { pwr7[i] -= base; if (pwr7[i] < 0) pwr7[i] = 0; tot += pwr7[i]; }
#pragma unroll
      for (int i = 0; i < 7; i++) pwr7[i] /= tot; }
    q.location = pnt; q.flag = 0; q.d0pha = 0; q.d1pha = 0; q.d2pha = 0; q.ampl = 1; q.noise = 0.001f; q.avgd2 = 0;
    int k = 0; float peak = 0.f;
#pragma unroll
    for (int i = 0; i < 7; i++) if (peak < pwr7[i]) { peak = pwr7[i]; k = i; }
    if (k == 0 || k == 6) return false;
    {                                                       // parabola through the peak and its neighbours (amplitudes): the decimals
      float lo = 0.f, hi = 0.f;
#pragma unroll
      for (int i = 1; i < 6; i++) if (i == k) { lo = (float)sqrt((double)pwr7[i - 1]); hi = (float)sqrt((double)pwr7[i + 1]); }
      const float mid = (float)sqrt((double)peak);
      float off; const float d = lo - hi, cur = 2 * (lo + hi - 2 * mid);
      if (cur < 0) { off = d / cur; if (fabs((double)off) > 1) off /= (float)fabs((double)off); } else off = lo > hi ? -1.f : 1.f;
      q.freq = pnt + k + off;
    }
    __builtin_amdgcn_wave_barrier();
    // first look: history weighted with the power spectrum, neighbouring bins with alternating sign
    for (int m = lane; m < n; m += 64) {
      const float *r = tab + slot(newest, m) * 14;
      float2 z = make_float2(0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 6; i += 2) { z.x += pwr7[i] * r[2 * i] - pwr7[i + 1] * r[2 * i + 2]; z.y += pwr7[i] * r[2 * i + 1] - pwr7[i + 1] * r[2 * i + 3]; }
      if (q.location & 1) { z.x = -z.x; z.y = -z.y; }
      zsig[2 * slot(newest, m)] = z.x; zsig[2 * slot(newest, m) + 1] = z.y;
    }
    __builtin_amdgcn_wave_barrier();
    float c0, c1, c2;
    estimate(newest, 0.0, 0.0, 0.0, c0, c1, c2);
    const float need = (float)(3 / sqrt((double)(float)n));
    if (q.ampl < need * q.noise) return false;
    const float turns = q.freq * a.freq_factor;             // whole turns per transform: fixed during the verification
    float e0 = 0.f, e1 = 0.f, e2 = 0.f;
    auto wrap_all = [](float v) { while (v > LRH_PI) v -= (float)(2 * LRH_PI); while (v < -LRH_PI) v += (float)(2 * LRH_PI); return v; };
    auto freq_at = [&](float sl) { float r = (float)(-0.5 * sl / LRH_PI); const int i = (int)(turns - r + 0.5); r += i; return r / a.freq_factor; };
    auto half_pos = [&](float fq) { int j = (int)(fq) + 2 - q.location - 4; if (j < 0) j = 0; j = 1 - j; if (j < 0) j = 0; return j; };
    for (int iter = 1; iter <= 5; iter++) {
      for (int m = lane; m < n; m += 64) {                 // every entry on the line shape of the loop's frequency at its age
        const float fq = freq_at(q.d1pha - m * q.d2pha);
        const int j = half_pos(fq);
        int id = (int)(256 * (fq - (int)(fq)));
        if (id == 256) id = 255;
        id = id * 8 + j;
        const int sidx = slot(newest, m);
        uind[sidx] = id;
        const float2 pr = project(tab + sidx * 14, id, j);
        zsig[2 * sidx] = pr.x; zsig[2 * sidx + 1] = pr.y;
      }
      __builtin_amdgcn_wave_barrier();
      estimate(newest, (double)q.d0pha, (double)q.d1pha, (double)q.d2pha, c0, c1, c2);
      if (q.ampl < need * q.noise) return false;
      q.d0pha = wrap_all(q.d0pha + c0); q.d1pha = wrap_all(q.d1pha + c1); q.d2pha = wrap_all(q.d2pha + c2);
      q.freq = freq_at(q.d1pha);
      if (iter > 1 && fabs((double)c0) < 0.1 && fabs((double)c1) < 0.01 && fabs((double)c2) < 0.001 &&
          fabs((double)e0) < 0.3 && fabs((double)e1) < 0.03 && fabs((double)e2) < 0.003) {
        // residual after the subtraction over the window and one bin either side, mean square per bin
        float res[9];
#pragma unroll
        for (int i = 0; i < 9; i++) res[i] = 0.f;
        const float turns2 = q.freq * a.freq_factor;
        for (int m = lane; m < n; m += 64) {
          float r = (float)(-0.5 * (q.d1pha - m * q.d2pha) / LRH_PI); const int it = (int)(turns2 - r + 0.5); r += it;
          const int j = half_pos(r / a.freq_factor);
          const int sidx = slot(newest, m), id = uind[sidx];
          float2 carrier = rot(make_float2(q.ampl, 0.f), (double)q.d0pha - m * (double)q.d1pha + 0.5 * m * (m - 1) * (double)q.d2pha);
          if ((j ^ (q.location & 1)) == 1) { carrier.x = -carrier.x; carrier.y = -carrier.y; }
          const float2 *z = a.fft2 + (size_t)sidx * a.n2 + q.location;
#pragma unroll
          for (int i = 0; i < 7; i++) {
            const float sh = spec[id + i]; const float2 v = z[i];
            res[i + 1] += (float)((double)(v.x - sh * carrier.x) * (double)(v.x - sh * carrier.x) + (double)(v.y - sh * carrier.y) * (double)(v.y - sh * carrier.y));
          }
          res[0] += z[-1].x * z[-1].x + z[-1].y * z[-1].y;
          res[8] += z[8].x * z[8].x + z[8].y * z[8].y;
        }
        float mean = 0.f;
#pragma unroll
        for (int i = 0; i < 9; i++) { res[i] = wsum(res[i]) / n; mean += res[i]; res[i] = (float)sqrt((double)res[i]); }
        mean = (float)sqrt((double)(mean / 9));
        float spread = 0.f;
#pragma unroll
        for (int i = 0; i < 9; i++) spread += (res[i] - mean) * (res[i] - mean);
        spread = (float)sqrt((double)(spread / 9));
        if (spread > 0.5 * mean / sqrt((double)n) + 0.02 * q.ampl) return false;
        q.avgd2 = q.d2pha;
        // initial_remove_spur (spursub.c:346-470, called right behind the lock, :309): the carrier also leaves the transforms the loop was closed on
        for (int m = lane; m < n; m += 64) {
          float r = (float)(-0.5 * (q.d1pha - m * q.d2pha) / LRH_PI); const int it = (int)(turns2 - r + 0.5); r += it;
          const int j = half_pos(r / a.freq_factor);
          const int sidx = slot(newest, m), id = uind[sidx];
          float2 carrier = rot(make_float2(q.ampl, 0.f), (double)q.d0pha - m * (double)q.d1pha + 0.5 * m * (m - 1) * (double)q.d2pha);
          if ((j ^ (q.location & 1)) == 1) { carrier.x = -carrier.x; carrier.y = -carrier.y; }
          float2 *z = a.fft2 + (size_t)sidx * a.n2 + q.location;
#pragma unroll
          for (int i = 0; i < 7; i++) { const float sh = spec[id + i]; float2 v = z[i]; v.x -= sh * carrier.x; v.y -= sh * carrier.y; z[i] = v; }
        }
        return true;
      }
      e0 = c0; e1 = c1; e2 = c2;
    }
    return false;
  }

  // one transform of a locked spur
  // `pre`: lane l < 7 brings bin pre_loc + l of this transform, fetched while the previous transform was being worked on
  __device__ void track(int na, float2 pre, int pre_loc)
  {
    float2 *z = a.fft2 + (size_t)na * a.n2;
    float *row = tab + slot(na, 0) * 14;
    int ind, j;
    auto fetch = [&]() -> float2 {                         // the transform's seven bins at the window's place
      if (pre_loc == q.location) return pre;
      return lane < 7 ? z[q.location + lane] : make_float2(0.f, 0.f);
    };
    if (q.flag == 1) {                                   // unlocked a moment ago: keep the window on the last known frequency
      j = (int)(q.freq) + 2 - q.location - 4;
      if (j < 0 || j > 1) move_window(j < 0 ? -1 : 1, na);
    }
    if (q.flag != 0) {                                   // not locked: the history goes on, re-locking is the control plane's business
      { const float2 v = fetch(); if (lane < 7) { row[2 * lane] = v.x; row[2 * lane + 1] = v.y; } }
      q.flag++;
      if (q.flag > 1000000) q.flag -= 2 * 3 * 5 * 7 * n;
      __builtin_amdgcn_wave_barrier();
      return;
    }
    q.freq = freq_of(q.d1pha + q.d2pha, q.freq);
    if (lane < 14) w.row[lane] = 0.f;                      // (a window move below shifts the copy along with the rows)
    __builtin_amdgcn_wave_barrier();
    if (!centre(q.freq, na, ind, j)) return;
    const float2 bins = fetch();
    const int bins_loc = q.location;
    if (lane < 7) { row[2 * lane] = bins.x; row[2 * lane + 1] = bins.y; w.row[2 * lane] = bins.x; w.row[2 * lane + 1] = bins.y; }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { const int s0 = slot(na, 0); uind[s0] = ind; const float2 pr = project(w.row, ind, j); zsig[2 * s0] = pr.x; zsig[2 * s0 + 1] = pr.y; }
    __builtin_amdgcn_wave_barrier();
    int moved = 0;
    for (int iter = 1;; iter++) {
      refine(na);
      // with the refined loop each history entry may belong to another line shape: frequency per age from the slope at that age
      // (integer part carried along from entry to entry), projection redone where the shape changed
      moved = 0;
      bool left = false;
      float fq_near = q.freq;
      for (int base = 0; base <= n; base += 64) {          // ages 0 .. n: one entry more than the estimate reads, like the reference's walk
        const int m = base + lane;
        const float sl = (q.d1pha + q.d2pha) - m * q.d2pha;
        // fractional part from this age's slope; integer part: nearest to the previous age's frequency, which changes by far less than a bin
        float fq = freq_of(sl, fq_near);
        const float prev = up1(fq);
        if (lane > 0) fq = freq_of(sl, prev);
        int jj; const int id = m <= n ? shape(fq, q.location, jj) : 0;
        if (__any(m <= n && id < 0)) { left = true; break; }
        if (m <= n) {
          const int sidx = slot(na, m);
          int k = (uind[sidx] - id + 2048) & 2047;
          if (k > 1024) k = 2048 - k;
          moved = max(moved, k);
          uind[sidx] = id;
          if (k != 0) { const float2 pr = project(m == 0 ? w.row : tab + sidx * 14, id, jj); zsig[2 * sidx] = pr.x; zsig[2 * sidx + 1] = pr.y; }
        }
        fq_near = lane_f(fq, 63);
      }
      moved = wmax(moved);
      __builtin_amdgcn_wave_barrier();
      if (left || !(moved > 20 && iter < 5)) break;
    }
    if (moved != 0) refine(na);
    if (fabsf(q.ampl) < a.minston * q.noise) { q.flag = 1; return; }      // lost in the noise
    // settled: the loop moves on by one transform and the carrier leaves the new bins
    const float curv = q.d2pha, ampl = q.ampl;
    const float slope = q.d1pha + curv, phase = q.d0pha + slope;
    auto wrap = [](float v) { if (v > LRH_PI) v -= (float)(2 * LRH_PI); if (v < -LRH_PI) v += (float)(2 * LRH_PI); return v; };
    q.d0pha = wrap(phase); q.d1pha = wrap(slope); q.d2pha = wrap(q.d2pha);
    q.avgd2 = a.weiold * q.avgd2 + a.weinew * curv;
    q.freq = freq_of(slope, q.freq);
    if (!centre(q.freq, na, ind, j)) return;
    float cr, ci; { float sn_, cs_; sincos_red((double)phase, sn_, cs_); cr = cs_ * ampl; ci = sn_ * ampl; }
    if ((j ^ (q.location & 1)) == 1) { cr = -cr; ci = -ci; }
    if (lane < 7) {                                        // (the bins are still in registers unless the window has moved since)
      float2 v = bins_loc == q.location ? bins : z[q.location + lane];
      const float sh = spec[ind + lane]; v.x -= sh * cr; v.y -= sh * ci; z[q.location + lane] = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
};

// LDS of k_spur: the loop's working set (SpurWork), the line-shape table, and the newest `slots` entries (a power of two >= speknum + 2)
// of the spur's history -- rows of seven bins, projected signal, shape index -- which live in global memory between launches.  Nothing
// on the path from one transform to the next waits for global memory: the history is read once, the transform's seven bins are
// fetched one transform ahead, stores are not waited for.
constexpr int SPUR_SPECTRA = 256 * 8;      // SPUR_SPECTRA (include/linrad_hip.h): 256 line shapes of 8 floats
__host__ __device__ inline int spur_lds_slots(int speknum) { int l = 4; while (l < speknum + 2) l <<= 1; return l; }
__host__ __device__ inline size_t spur_lds_bytes(int speknum, bool mirror)
{
  size_t b = (size_t)speknum * (3 * sizeof(float2) + sizeof(float)) + 16 * sizeof(float) + 16;
  if (mirror) b += sizeof(float) * (SPUR_SPECTRA + (size_t)spur_lds_slots(speknum) * 17);
  return b;
}
__global__ __launch_bounds__(64) void k_spur(SpurArgs a)
{
  extern __shared__ float spur_lds[];
  const int s = blockIdx.x, lane = threadIdx.x, n = a.speknum;
  const int gmax = a.na_mask + 1, L = spur_lds_slots(n);
  DevSpur q = reinterpret_cast<DevSpur *>(a.spurs)[s];
  SpurWork w;
  w.h = reinterpret_cast<float2 *>(spur_lds); w.d = w.h + n; w.g = w.d + n; w.t = reinterpret_cast<float *>(w.g + n); w.row = w.t + n + 2;
  float *spec = w.row + 16 + 4, *tab_l = spec + SPUR_SPECTRA, *zsig_l = tab_l + (size_t)L * 14;
  int *uind_l = reinterpret_cast<int *>(zsig_l + 2 * L);
  float *const g_tab = a.table + (size_t)s * gmax * 14, *const g_sig = a.signal + (size_t)s * gmax * 2;
  int *const g_ind = a.ind + (size_t)s * gmax;
  for (int i = lane; i < SPUR_SPECTRA; i += 64) spec[i] = a.spectra[i];
  for (int m = lane; m <= n; m += 64) {                    // the history behind the first transform of this launch
    const int gs = (a.first_na - 1 - m) & a.na_mask, ls = gs & (L - 1);
#pragma unroll
    for (int i = 0; i < 14; i++) tab_l[ls * 14 + i] = g_tab[gs * 14 + i];
    zsig_l[2 * ls] = g_sig[2 * gs]; zsig_l[2 * ls + 1] = g_sig[2 * gs + 1]; uind_l[ls] = g_ind[gs];
  }
  __builtin_amdgcn_wave_barrier();
  SpurWave T{a, q, tab_l, zsig_l, uind_l, w, lane, n, L, L - 1, spec};
  int lo = q.location, hi = q.location;
  float2 pre = make_float2(0.f, 0.f); int pre_loc = q.location;
  if (lane < 7) pre = a.fft2[(size_t)(a.first_na & a.na_mask) * a.n2 + pre_loc + lane];
  for (int b = 0; b < a.batch; b++) {
    const int na = (a.first_na + b) & a.na_mask;
    float2 nxt = make_float2(0.f, 0.f); const int nxt_loc = q.location;
    if (b + 1 < a.batch && lane < 7) nxt = a.fft2[(size_t)((a.first_na + b + 1) & a.na_mask) * a.n2 + nxt_loc + lane];
    T.track(na, pre, pre_loc);
    pre = nxt; pre_loc = nxt_loc;
    lo = min(lo, q.location); hi = max(hi, q.location);
  }
  __builtin_amdgcn_wave_barrier();
  for (int m = lane; m <= n; m += 64) {                    // the history as the next launch will want it
    const int gs = (a.first_na + a.batch - 1 - m) & a.na_mask, ls = gs & (L - 1);
#pragma unroll
    for (int i = 0; i < 14; i++) g_tab[gs * 14 + i] = tab_l[ls * 14 + i];
    g_sig[2 * gs] = zsig_l[2 * ls]; g_sig[2 * gs + 1] = zsig_l[2 * ls + 1]; g_ind[gs] = uind_l[ls];
  }
  if (lane == 0) {
    reinterpret_cast<DevSpur *>(a.spurs)[s] = q;
    if (a.touched) { a.touched[2 * s] = lo; a.touched[2 * s + 1] = hi + 7; }   // bins whose power sums k_spur_patch redoes
  }
}
hipError_t launch_spur(const SpurArgs &a, hipStream_t st)
{
  const size_t lds = spur_lds_bytes(a.speknum, true);
  if (lds > 150 * 1024) return hipErrorInvalidValue;       // speknum <= 1022 (lrh_spur_config)
  if (lds > 48 * 1024) {
    static bool raised = false;
    if (!raised) { const hipError_t e = hipFuncSetAttribute((const void *)k_spur, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); if (e != hipSuccess) return e; raised = true; }
  }
  hipLaunchKernelGGL(k_spur, dim3(a.nspurs), dim3(64), lds, st, a);
  return hipGetLastError();
}

// store_new_spur + spur_phase_lock for spur number `a.nspurs` (the next free one): result[0] = 1 when the loop locked
__global__ __launch_bounds__(64) void k_spur_acquire(SpurArgs a, int pnt, int *result)
{
  extern __shared__ float spur_lds[];
  const int s = a.nspurs, lane = threadIdx.x, n = a.speknum;
  const int maxn = a.na_mask + 1;
  DevSpur q;
  SpurWork w;
  w.h = reinterpret_cast<float2 *>(spur_lds); w.d = w.h + n; w.g = w.d + n; w.t = reinterpret_cast<float *>(w.g + n); w.row = w.t + n + 2;
  SpurWave L{a, q, a.table + (size_t)s * maxn * 14, a.signal + (size_t)s * maxn * 2, a.ind + (size_t)s * maxn, w, lane, n, maxn, a.na_mask, a.spectra};
  const bool ok = L.acquire(a.first_na & a.na_mask, pnt);
  if (lane == 0) { reinterpret_cast<DevSpur *>(a.spurs)[s] = q; result[0] = ok ? 1 : 0; }
}
hipError_t launch_spur_acquire(const SpurArgs &a, int pnt, int *result, hipStream_t st)
{
  const size_t lds = (size_t)a.speknum * (3 * sizeof(float2) + sizeof(float)) + 16 * sizeof(float) + 16;
  if (lds > 60 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_spur_acquire, dim3(1), dim3(64), lds, st, a, pnt, result);
  return hipGetLastError();
}

// The transform kernels have already summed |X|^2 over the waterfall groups (k_fft2<.., fused>, k_fft2_rows) when k_spur takes the
// carriers out: the sums of the few bins it touched are redone here from the cleaned spectra, same group arithmetic and the same
// additions in the same order as the transform kernels (k_powersum2's), so that the fused form stays in use with spurs tracked.
__global__ __launch_bounds__(64) void k_spur_patch(SpurPatchArgs a)
{
  const int s = blockIdx.x, g = blockIdx.y;
  const int lo = max(a.touched[2 * s], 0), hi = min(a.touched[2 * s + 1], a.n);
  const int start = g == 0 ? 0 : g * a.avgnum - a.counter;
  int count = a.avgnum - (g == 0 ? a.counter : 0);
  const bool complete = count <= a.count - start;
  if (!complete) count = a.count - start;
  const bool accumulate = g == 0 && a.counter > 0;
  for (int i = lo + threadIdx.x; i < hi; i += 64) {
    float acc = accumulate ? a.powersum_in[i] : 0.f;
    for (int b = 0; b < count; b++) {
      const float2 v = a.fft2[(size_t)((a.first_na + start + b) & a.na_mask) * a.n + i];
      const float pw = v.x * v.x + v.y * v.y;
      acc = (b == 0 && !accumulate) ? pw : acc + pw;
    }
    if (complete) a.wf_scratch[(size_t)g * a.n + i] = acc;
    if (g == (int)gridDim.y - 1) a.powersum_out[i] = acc;
  }
}
hipError_t launch_spur_patch(const SpurPatchArgs &a, int nspurs, int ngroups, hipStream_t st)
{
  hipLaunchKernelGGL(k_spur_patch, dim3(nspurs, ngroups), dim3(64), 0, st, a);
  return hipGetLastError();
}
}  // namespace lrh
