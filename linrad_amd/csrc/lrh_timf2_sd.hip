// lrh_timf2_sd.hip -- the strong stream of make_timf2 by direct summation (gfx950), see the kernel's comment.
#include <cstdlib>
#include "lrh_fft.hip.h"
#include "lrh_kernels.hip.h"
#pragma clang diagnostic ignored "-Winline-asm"   // the "m0" clobber of the index-register block below is deliberate (M0 is a reserved register: the note is all the warning says)

namespace lrh {
// =====================================================================================================
// strong stream of make_timf2 by direct summation (k_timf2_sd)
// =====================================================================================================
// Behind k_fft1v / k_fft1w the only thing left of make_timf2 is the back transform of the bins routed strong -- normally a few
// dozen of 16384 (the carriers the selective limiter takes out of the blanker's way, timf2.c:48-66).  A full transform per block for
// them (k_timf2<.., STRONG_ONLY>) costs what any transform costs: three exchanges through the LDS behind barriers, 131-151 us per 4096
// blocks to write 256 MiB.  Here the K listed bins are summed directly, split the way the output is stored:
//     n = n1 + T n2  (T = N/32 threads, thread n1),   out[n] = sum_k2 W32^(n2 k2) A[k2],   A[k2] = sum_{k = k2 mod 32} C[k] W_N^(n1 k)
// A[] costs one complex multiply-add and one twiddle per listed bin and thread, the 32-point transform over k2 stays in registers and
// only its first 16 outputs are formed (sin^2 overlap: one transform of the combined spectrum C = S_t + (-1)^k S_(t-1), of which the
// first half is kept, see k_timf2) -- two 16-point transforms and 16 products.  No exchange, one barrier per block (the K combined
// values go through LDS), every store a whole 512-byte line per wave.
// The strong bins come as runs of neighbours (a carrier and its window skirts): the list is cut into runs of at most 8 consecutive
// bins (a run never crosses a multiple of 8), the twiddle of a run's first bin comes from two table cells in LDS (W^(128 h) W^l),
// the following ones by one multiplication with the thread's own W^(n1) each -- a table look-up per bin made the kernel LDS-bound
// (two 64-lane gathers per bin and wave: 146 us).  The accumulator A[k mod 32] is picked with the VGPR index register (the bin
// number is the same in every lane): one v_add per component.  The list is made once per workgroup from the routing words, in
// ascending order of the bin number: the order of summation is fixed.  More than LRH_SD_KMAX bins routed strong: the kernel returns
// and k_timf2<.., STRONG_ONLY>, launched behind it, does the launch (both count the same bits).
typedef float v32f __attribute__((ext_vector_type(32)));

template <int LOG2N>
__global__ __launch_bounds__((1 << LOG2N) / 32) void k_timf2_sd(Timf2Args a)
{
  constexpr int N = 1 << LOG2N, T = N / 32, NW = T / 64, NBW = N / 16, SH = LOG2N - 4, HI = N / 128, KM = LRH_SD_KMAX;
  static_assert(LOG2N >= 12 && LOG2N <= 14, "fft1_size 4096 / 8192 / 16384");
  __shared__ unsigned int s_pack[2][NBW];                 // routing words: [0] the batch's table, [1] the table of the transform before it
  __shared__ int s_wsum[2][8], s_tot[2], s_heads[2], s_hw[2][2];
  __shared__ unsigned short s_k[2][KM];                   // listed bins, ascending: set 0 strong under the batch's table, set 1 under either (block 0)
  __shared__ unsigned char s_fl[KM];                      // set 1: bit 0 strong in S_t's table, bit 1 strong in S_(t-1)'s
  __shared__ unsigned int s_run[2][KM + 1];               // runs: first bin | entries << 16 | first entry << 20
  __shared__ float2 s_twA[HI], s_twB[128];
  __shared__ float2 s_C[2][KM + 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < NBW; i += T) { s_pack[0][i] = a.pack_cur[i]; s_pack[1][i] = a.pack_prev[i]; }
  if (tid < 128) s_twB[tid] = a.tw[tid];
  if (tid < HI) s_twA[tid] = a.tw[tid * 128];
  for (int i = tid; i < 2 * (KM + 8); i += T) s_C[0][i] = make_float2(0.f, 0.f);
  for (int i = tid; i < 2 * (KM + 1); i += T) s_run[0][i] = 0u;
  const lrh_v2f g1 = to_v(a.tw[tid]);                      // W_N^(n1): from one bin of a run to the next
  __syncthreads();
  // bit s of word i: bin i + s N/16 is weak (lrh_set_liminfo, sl_pack).  Thread j looks at the bins 32 j .. 32 j + 31.
  unsigned int sw[2] = {0u, 0u};
#pragma unroll 8
  for (int k2 = 0; k2 < 32; k2++) {
    const int bin = 32 * tid + k2, i = bin & (NBW - 1), sbit = bin >> SH;
    const unsigned int wc = (s_pack[0][i] >> sbit) & 1u, wp = (s_pack[1][i] >> sbit) & 1u;
    sw[0] |= (wc ^ 1u) << k2; sw[1] |= ((wc & wp) ^ 1u) << k2;
  }
  int off[2];
#pragma unroll
  for (int set = 0; set < 2; set++) {                      // rank of the thread's first strong bin: prefix of the popcounts over the workgroup
    const int c = __popc(sw[set]);
    int x = c;
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if (lane >= d) x += y; }
    if (lane == 63) s_wsum[set][wave] = x;
    off[set] = x - c;
  }
  __syncthreads();
#pragma unroll
  for (int set = 0; set < 2; set++) {
    int b = 0, tot = 0;
    for (int w = 0; w < NW; w++) { const int v = s_wsum[set][w]; if (w < wave) b += v; tot += v; }
    off[set] += b;
    if (tid == 0) s_tot[set] = tot;
  }
  __syncthreads();
  if (s_tot[1] > a.sd_kmax || s_tot[1] > KM) return;       // too many for this kernel: the transform kernel behind it takes the launch
#pragma unroll
  for (int set = 0; set < 2; set++) {
    unsigned int m = sw[set];
    int r = off[set];
    while (m) {
      const int l = __builtin_ctz(m); m &= m - 1;
      const int k = 32 * tid + l;
      s_k[set][r] = (unsigned short)k;
      if (set == 1) s_fl[r] = (unsigned char)(((sw[0] >> l) & 1u) | 2u * ((((s_pack[1][k & (NBW - 1)] >> (k >> SH)) & 1u) ^ 1u)));
      r++;
    }
  }
  __syncthreads();
  // runs: entry i starts one when it is the first, does not follow its predecessor, or its bin is a multiple of 8
#pragma unroll
  for (int set = 0; set < 2; set++) {
    const int K = s_tot[set];
    bool head = false; int k = 0;
    if (tid < K) { k = s_k[set][tid]; head = tid == 0 || (int)s_k[set][tid - 1] + 1 != k || (k & 7) == 0; }
    const unsigned long long hm = __ballot(head);
    if (tid < 128 && lane == 0) s_hw[set][wave] = __popcll(hm);
    __syncthreads();
    if (head) {
      const int r = __popcll(hm & ((1ull << lane) - 1ull)) + (wave ? s_hw[set][0] : 0);
      int len = 1;
      while (len < 8 && tid + len < K && (int)s_k[set][tid + len] == k + len && ((k + len) & 7) != 0) len++;
      s_run[set][r] = (unsigned int)k | ((unsigned int)len << 16) | ((unsigned int)tid << 20);
    }
    if (tid == 0) s_heads[set] = s_hw[set][0] + (KM > 64 ? s_hw[set][1] : 0);
  }
  __syncthreads();
  auto order = [&](int i) { return a.xcd ? xcd_order(i, a.batch) : i; };
  // the K combined values of a block: thread i < K fetches bin k_i of S_t and of S_(t-1) (whichever its table routes strong)
  auto fetch = [&](int b, float2 &cu, float2 &pv) {
    const int set = b == 0 ? 1 : 0, K = s_tot[set];
    cu = make_float2(0.f, 0.f); pv = make_float2(0.f, 0.f);
    if (tid < K) {
      const int k = s_k[set][tid], fl = set ? s_fl[tid] : 3;
      if (fl & 1) cu = a.spec[(size_t)((a.first_nb + b) & a.nb_mask) * N + k];
      if (fl & 2) pv = a.spec[(size_t)((a.first_nb + b - 1) & a.nb_mask) * N + k];
    }
  };
  auto put = [&](int b, int buf, float2 cu, float2 pv) {
    const int set = b == 0 ? 1 : 0, K = s_tot[set];
    if (tid < K) {
      const float sg = (s_k[set][tid] & 1) ? -a.ampfac : a.ampfac;       // S_t + (-1)^k S_(t-1), scaled like timf2_store does
      s_C[buf][tid] = make_float2(a.ampfac * cu.x + sg * pv.x, a.ampfac * cu.y + sg * pv.y);
    }
  };
  auto cells = [&](unsigned int k, lrh_v2f &wa, lrh_v2f &wb) {
    const unsigned int idx = __umul24((unsigned int)tid, k) & (unsigned int)(N - 1);
    wa = to_v(s_twA[idx >> 7]); wb = to_v(s_twB[idx & 127]);
  };
  int bi = blockIdx.x, buf = 0;
  float2 cu, pv;
  if (bi < a.batch) { fetch(order(bi), cu, pv); put(order(bi), 0, cu, pv); }
  __syncthreads();
#pragma unroll 1
  for (; bi < a.batch; bi += gridDim.x) {
    const int b = order(bi), set = b == 0 ? 1 : 0;
    const int bn = bi + (int)gridDim.x < a.batch ? order(bi + gridDim.x) : b;
    fetch(bn, cu, pv);                                     // the next block's values are on their way during this one's sums
    v32f Ar = 0.f, Ai = 0.f;
    const int H = __builtin_amdgcn_readfirstlane(s_heads[set]);
    unsigned int rw = s_run[set][0];
    lrh_v2f wa, wb;
    cells(rw & 0xffffu, wa, wb);
#pragma unroll 1
    for (int h = 0; h < H; h++) {
      const unsigned int run = (unsigned int)__builtin_amdgcn_readfirstlane((int)rw);
      const int k0 = run & 0xffff, len = (run >> 16) & 15, i0 = run >> 20;
      lrh_v2f c[8];
#pragma unroll
      for (int j = 0; j < 8; j++) c[j] = to_v(s_C[buf][i0 + j]);       // (the array is 8 longer than the list)
      rw = s_run[set][h + 1];                              // the next run's first twiddle: its cells are read during this run's sums
      lrh_v2f wa_n, wb_n;
      cells(rw & 0xffffu, wa_n, wb_n);
      lrh_v2f w = cmul_v(wa, wb);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if (j < len) {
          const lrh_v2f v = cmul_v(c[j], w);
          const int k2 = (k0 + j) & 31;
          // A[k2] += v with the index register on source 0 and destination: the accumulators live in v[64:127]
          asm volatile("s_set_gpr_idx_on %2, gpr_idx(SRC0,DST)\n\tv_add_f32 v64, v64, %3\n\tv_add_f32 v96, v96, %4\n\ts_set_gpr_idx_off"
                       : "+{v[64:95]}"(Ar), "+{v[96:127]}"(Ai) : "s"(k2), "v"(v.x), "v"(v.y) : "m0");      // s_set_gpr_idx_on writes M0: declared, so the compiler keeps nothing live in it across the block
          if (j + 1 < len) w = cmul_v(w, g1);
        }
      }
      wa = wa_n; wb = wb_n;
    }
    // 32-point transform over k2, outputs n2 < 16: two 16-point transforms of the even and of the odd k2, out = E + W32^n2 O
    float2 E[16], O[16];
#pragma unroll
    for (int m = 0; m < 16; m++) { E[m] = make_float2(Ar[2 * m], Ai[2 * m]); O[m] = make_float2(Ar[2 * m + 1], Ai[2 * m + 1]); }
    Dft<-1, 16>::run(E); Dft<-1, 16>::run(O);
    const int pa = a.pa_first + b * a.step;
    float2 *const dst = a.timf2s + (size_t)(pa & a.mask) + tid;
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) {
      const lrh_v2f o = to_v(E[n2]) + cmulc_v(to_v(O[n2]), lrh_cos32(n2), -lrh_sin32(n2));
      __builtin_nontemporal_store(o, reinterpret_cast<lrh_v2f *>(dst + n2 * T));
    }
    put(bn, buf ^ 1, cu, pv);
    __syncthreads();
    buf ^= 1;
  }
}

hipError_t launch_timf2_sd(int log2n, const Timf2Args &a, hipStream_t st)
{
  const int T = (1 << log2n) / 32, per_cu = 1024 / T > 2 ? 1024 / T : 2;
  const int g = a.batch < 256 * per_cu ? a.batch : 256 * per_cu;
  switch (log2n) {
    case 14: hipLaunchKernelGGL((k_timf2_sd<14>), dim3(g), dim3(T), 0, st, a); break;
    case 13: hipLaunchKernelGGL((k_timf2_sd<13>), dim3(g), dim3(T), 0, st, a); break;
    case 12: hipLaunchKernelGGL((k_timf2_sd<12>), dim3(g), dim3(T), 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
}  // namespace lrh
