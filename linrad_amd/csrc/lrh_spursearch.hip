// lrh_spursearch.hip -- the search spectrum for new spurs on the device-resident fft2 power rows (gfx950).
#include "lrh_kernels.hip.h"
#include "lrh_phase.h"

namespace lrh {
// make_fft2's bookkeeping of the search spectrum for one new power row (fft2.c:673-699): element-wise over the search range
__global__ __launch_bounds__(256) void k_spur_search_row(SpurSearchArgs a)
{
  const int i = a.first + blockIdx.x * 256 + threadIdx.x;
  if (i > a.last) return;
  const float2 v = a.z[i];
  const float p = v.x * v.x + v.y * v.y;                   // pwra[i], fft2.c:655-668
  if (a.mode == 0) a.sum[i] = p;
  else if (a.mode == 1) a.sum[i] += p;
  else a.spec[i] = a.sum[i] + p;
}

__device__ __forceinline__ void ss_parabolic_fit(float *amp, float *pos, float y1, float y2, float y3)      // parabolic_fit, llsq.c:113-153
{
  float t4 = y1 - y3;
  const float t3 = 2 * (y1 + y3 - 2 * y2);
  if (t3 < 0) { *amp = y2 - 0.5F * t4 * t4 / t3; t4 = t4 / t3; if (fabsf(t4) > 1) t4 /= fabsf(t4); *pos = t4; }
  else if (y1 > y3) { *amp = y1; *pos = -1; }
  else { *amp = y3; *pos = 1; }
}

// spursearch_spectrum_cleanup (spursub.c:40-175), once per 3 spur_speknum + 2 transforms, one workgroup on a stream of its own:
// the minima of the groups of 32 bins and the floor subtraction are spread over the threads; the two averages of the minima and the walk
// over the peaks -- each may widen the stretch it wipes into what the walk has not reached yet -- are the reference's serial loops,
// run by one lane in the reference's order of float operations (a few hundred microseconds at fft2_size 65536, off the main stream).
__global__ __launch_bounds__(1024) void k_spur_search_cleanup(SpurSearchArgs a)
{
  __shared__ float s_noise, s_thr;
  float *sp = a.spec;
  const int first = a.first, last = a.last, tid = threadIdx.x;
  const int k = (last - first + 31) / 32;
  for (int g = tid; g < k; g += 1024) {
    float m = 1e30f;
    for (int j = 0; j < 32; j++) { const float v = sp[first + 32 * g + j]; if (v < m) m = v; }
    a.mins[g] = m;
  }
  __syncthreads();
  if (k == 0) return;
  if (tid == 0) {
    float t1 = 0;
    for (int i = 0; i < k; i++) t1 += a.mins[i];
    t1 /= k;
    float noise = 0; int j = 0;
    for (int i = 0; i < k; i++) if (a.mins[i] < t1) { noise += a.mins[i]; j++; }
    noise /= j;
    noise = (float)((double)noise * a.noise_factor);
    s_noise = noise; s_thr = (float)((double)noise * a.thr_factor);
    a.out[0] = s_thr; a.out[1] = noise;
  }
  __syncthreads();
  const float noise = s_noise, thr = s_thr;
  for (int i = first + tid; i < last; i += 1024) { float v = sp[i] - noise; if (v < 0) v = 0; sp[i] = v; }
  __threadfence();
  __syncthreads();
  if (tid != 0) return;
  int ia = first;
  for (;;) {
    while (ia < last && sp[ia] < thr) ia++;
    if (ia == last) break;
    int ib = ia + 1;
    while (ib < last && sp[ib] > thr) ib++;
    if (ib == last && ib - ia < 8) break;
    float maxpow = 0; int kk = ia;
    for (int i = ia; i < ib; i++) if (sp[i] > maxpow) { maxpow = sp[i]; kk = i; }
    float amp, pos;
    ss_parabolic_fit(&amp, &pos, sp[kk - 1], sp[kk], sp[kk + 1]);
    int nn = kk - 8 / 2 + 1;
    if (pos < 0) { pos += 1; nn--; }
    int si = (int)(pos * 256); if (si >= 256) si = 255;
    const float *spk = a.spectra + si * 8;
    float refamp = fabsf(spk[4]);
    if (refamp < fabsf(spk[3])) refamp = fabsf(spk[3]);
    const float maxamp = (float)sqrt((double)maxpow), t2 = maxamp / refamp;
    float tot = 0, rem = 0, edge = 0; bool bad = false;
    for (int i = 0; i < 8; i++) {
      const float v = sp[nn + i];
      if (v < 0) { bad = true; break; }
      tot += v;
      const double dd = sqrt((double)v) - (double)t2 * fabs((double)spk[i]);
      const float r1 = (float)(dd * dd);
      rem += r1;
      if (edge < r1 && (i < 2 || i >= 6)) edge = r1;
    }
    if (bad || (edge / noise > 5 && tot / edge < 1000) || (edge / noise > 2 && tot / edge < 300) || rem / tot > 0.1) {
      while ((sp[ia] > sp[ia - 1] || sp[ia] > sp[ia - 2] || sp[ia] > sp[ia - 3]) && ia > first) ia--;
      while ((sp[ib] > sp[ib + 1] || sp[ib] > sp[ib + 2] || sp[ib] > sp[ib + 3]) && ib < last) ib++;
      for (int i = ia; i < ib; i++) sp[i] = -0.00000001f;
    }
    ia = ib;
  }
}

hipError_t launch_spur_search_row(const SpurSearchArgs &a, hipStream_t st)
{
  hipLaunchKernelGGL(k_spur_search_row, dim3((a.last - a.first + 256) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_spur_search_cleanup(const SpurSearchArgs &a, hipStream_t st)
{
  hipLaunchKernelGGL(k_spur_search_cleanup, dim3(1), dim3(1024), 0, st, a);
  return hipGetLastError();
}

// mix1's phase tables (lrh_host.hip mix1_run): the host writes, per transform, the phase increments (t2, r2), the two running phases at the transform's
// first sample and the bin to cut at into a page-locked table; this kernel reads it from there (the host pays one launch, not a hipMemcpyAsync of
// 75-88 us) and writes the table k_mix1_back / k_mix1_out read: increments, the phases at every LRH_PH_CHUNK-th sample -- derived with the same
// closed-form advance the host walks from transform to transform with, bit for bit the float recursion of do_mix1 (mix1.c:141-195) -- and the bins
__global__ __launch_bounds__(256) void k_phase_expand(const float2 *h_inc, const float2 *h_st, const int *h_point, float2 *d_inc, float2 *d_start, int *d_point,
                                                      int batch, int nchunks, int chunk)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= batch * nchunks) return;
  const int b = i / nchunks, cidx = i - b * nchunks;
  const float2 inc = h_inc[b], st = h_st[b];
  d_start[i] = make_float2(lrh_phase_advance(st.x, inc.x, cidx * chunk), lrh_phase_advance(st.y, inc.y, cidx * chunk));
  if (cidx == 0) { d_inc[b] = inc; if (d_point) d_point[b] = h_point[b]; }
}
hipError_t launch_phase_expand(const float2 *h_inc, const float2 *h_st, const int *h_point, float2 *d_inc, float2 *d_start, int *d_point, int batch, int nchunks, int chunk, hipStream_t st)
{
  if (batch < 1) return hipSuccess;
  hipLaunchKernelGGL(k_phase_expand, dim3((batch * nchunks + 255) / 256), dim3(256), 0, st, h_inc, h_st, h_point, d_inc, d_start, d_point, batch, nchunks, chunk);
  return hipGetLastError();
}

// a few KiB from a page-locked host table to the device by a kernel: the host pays one launch (mix1's phase tables, lrh_host.hip mix1_run)
__global__ __launch_bounds__(256) void k_copy_words(const unsigned int *src, unsigned int *dst, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
hipError_t launch_copy_words(const unsigned int *src, unsigned int *dst, size_t n, hipStream_t st)
{
  if (n == 0) return hipSuccess;
  int g = (int)((n + 255) / 256); if (g > 64) g = 64;
  hipLaunchKernelGGL(k_copy_words, dim3(g), dim3(256), 0, st, src, dst, n);
  return hipGetLastError();
}
}  // namespace lrh
