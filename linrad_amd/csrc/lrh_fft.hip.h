// lrh_fft.hip.h -- workgroup-cooperative complex FFT for gfx950 (CDNA4, wave64).
//
// One workgroup transforms one length-N sequence.  Every thread keeps P points in
// registers for the whole transform; a pass is a radix-16/8/4/2 Stockham butterfly done
// entirely in registers, and between passes the points are redistributed through LDS
// (float2 cells, ds_write_b64 / ds_read_b64, one pad cell per 16 to spread the strided
// scatter of the first exchange over all banks).  N = 16384 uses 1024 threads x 16 points
// and 136 KiB of the CU's 160 KiB LDS; the first pass reads straight from global memory
// and the last pass leaves natural-order results in registers, so a length-16384
// transform costs three LDS round trips instead of the reference's fourteen radix-2
// sweeps over a 128 KiB array (fft0.c:161-195).
//
// Index scheme (natural order in, natural order out, no bit reversal):
//   pass with radix R, completed length p, butterfly i in [0, N/R):
//     k = i mod p,  inputs  x[i + s*N/R] * w^(s),  w = exp(-+2 pi j k/(p R)),
//     outputs y[(i-k)*R + k + q*p] = sum_s (...) exp(-+2 pi j s q / R)
//   thread tid owns butterflies i = tid + m*T (T = N/P threads, m < P/R).
#pragma once
#include <hip/hip_runtime.h>

namespace lrh {

// Complex values travel as float2 (HIP's struct) between kernels and butterflies; inside the butterflies they are
// native 2-vectors so that every complex add/sub is ONE packed instruction and nothing is left to the SLP
// vectoriser, which pairs unrelated halves and then spends a v_mov per butterfly output re-pairing them.
typedef float lrh_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lrh_v2f to_v(float2 a) { return lrh_v2f{a.x, a.y}; }
__device__ __forceinline__ float2 to_f2(lrh_v2f a) { return make_float2(a.x, a.y); }

// complex multiply by a compile-time constant c + js in two packed ops: (u.x,u.x)*(c,s) + (u.y,u.y)*(-s,c)
__device__ __forceinline__ lrh_v2f cmulc_v(lrh_v2f u, float c, float s)
{
  const lrh_v2f t = lrh_v2f{u.x, u.x} * lrh_v2f{c, s};
  return __builtin_elementwise_fma(lrh_v2f{u.y, u.y}, lrh_v2f{-s, c}, t);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return to_f2(cmulc_v(to_v(a), b.x, b.y)); }
// complex multiply of two register values in two packed instructions:
//   t = (a.x b.x, a.x b.y);  r = (t.x - a.y b.y, t.y + a.y b.x)
// hipcc's own lowering of the scalar expression takes three packed ops plus a v_mov to re-pair the halves, and the
// per-lane negation is not folded from vector code either (a v_xor appears), hence the asm.
__device__ __forceinline__ lrh_v2f cmul_v(lrh_v2f av, lrh_v2f bv)
{
  lrh_v2f t, r;
#ifdef LRH_CMUL_ONE_ASM
  // one statement: the compiler puts a hazard nop between two asm statements whose second reads the first's result (it cannot see
  // what they are); the pair needs none.  Costs an early-clobber temporary: only where registers are to spare (k_fft1v).
  asm("v_pk_mul_f32 %1, %2, %3 op_sel_hi:[0,1]\n\tv_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r), "=&v"(t) : "v"(av), "v"(bv));
#else
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(av), "v"(bv));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(av), "v"(bv), "v"(t));
#endif
  return r;
}
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return to_f2(cmul_v(to_v(a), to_v(b))); }
// a * conj(b), same two instructions with the negation moved:  t = (a.x b.x, -a.x b.y);  r = (t.x + a.y b.y, t.y + a.y b.x)
__device__ __forceinline__ lrh_v2f cmul_conj_v(lrh_v2f av, lrh_v2f bv)
{
  lrh_v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(av), "v"(bv));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(av), "v"(bv), "v"(t));
  return r;
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b) { return to_f2(cmul_conj_v(to_v(a), to_v(b))); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -j (DIR = -1, forward) or +j (DIR = +1)
template <int DIR> __device__ __forceinline__ float2 mulj(float2 a) { return DIR < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x); }
template <int DIR> __device__ __forceinline__ float2 tw_dir(float2 w) { return DIR < 0 ? w : make_float2(w.x, -w.y); }
// s + mulj<DIR>(t) and s - mulj<DIR>(t) in one packed add each: swapped halves of t, one lane negated
template <int DIR> __device__ __forceinline__ lrh_v2f add_mulj(lrh_v2f s, lrh_v2f t)
{
  lrh_v2f r;
  if (DIR < 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(s), "v"(t));   // (s.x + t.y, s.y - t.x)
  else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(s), "v"(t));           // (s.x - t.y, s.y + t.x)
  return r;
}
template <int DIR> __device__ __forceinline__ lrh_v2f sub_mulj(lrh_v2f s, lrh_v2f t) { return add_mulj<-DIR>(s, t); }

template <int DIR> __device__ __forceinline__ void bfly2(float2 &a, float2 &b) { float2 t = csub(a, b); a = cadd(a, b); b = t; }

// 4-point DFT, outputs in natural order: a,b,c,d <- X0..X3 of inputs x0..x3 = a,b,c,d
template <int DIR> __device__ __forceinline__ void bfly4_v(lrh_v2f &a, lrh_v2f &b, lrh_v2f &c, lrh_v2f &d)
{
  const lrh_v2f s0 = a + c, s1 = a - c, s2 = b + d, t = b - d;
  a = s0 + s2; c = s0 - s2; b = add_mulj<DIR>(s1, t); d = sub_mulj<DIR>(s1, t);
}
template <int DIR> __device__ __forceinline__ void bfly4(float2 &a, float2 &b, float2 &c, float2 &d)
{
  lrh_v2f av = to_v(a), bv = to_v(b), cv = to_v(c), dv = to_v(d);
  bfly4_v<DIR>(av, bv, cv, dv);
  a = to_f2(av); b = to_f2(bv); c = to_f2(cv); d = to_f2(dv);
}

#define LRH_C8 0.70710678118654752440f
#define LRH_C16A 0.92387953251128675613f /* cos(pi/8) */
#define LRH_S16A 0.38268343236508977173f /* sin(pi/8) */

// exp(DIR * 2 pi j * e / 16), e in 0..9 (only the values the 4x4 split needs)
template <int DIR, int E> __device__ __forceinline__ float2 w16()
{
  constexpr float c[10] = {1.f, LRH_C16A, LRH_C8, LRH_S16A, 0.f, -LRH_S16A, -LRH_C8, -LRH_C16A, -1.f, -LRH_C16A};
  constexpr float s[10] = {0.f, LRH_S16A, LRH_C8, LRH_C16A, 1.f, LRH_C16A, LRH_C8, LRH_S16A, 0.f, -LRH_S16A};
  return make_float2(c[E], DIR < 0 ? -s[E] : s[E]);
}

// exp(DIR * 2 pi j * e / 32), any e (the last-pass roots of BlockFftL: order = points per thread)
__host__ __device__ constexpr float lrh_cos32(int e)
{
  constexpr float q[9] = {1.f, 0.98078528040323044913f, LRH_C16A, 0.83146961230254523708f, LRH_C8, 0.55557023301960222474f, LRH_S16A, 0.19509032201612826785f, 0.f};
  e &= 31;
  return e <= 8 ? q[e] : (e <= 16 ? -q[16 - e] : (e <= 24 ? -q[e - 16] : q[32 - e]));
}
__host__ __device__ constexpr float lrh_sin32(int e) { return lrh_cos32(e - 8); }
// exp(2 pi j e / 64), 0 <= e < 32 (k_fft1v<REAL>: the split's twiddle)
__host__ __device__ constexpr float lrh_cos64(int e)
{
  constexpr float q[17] = {1.f, 0.99518472667219688624f, 0.98078528040323044913f, 0.95694033573220886494f, 0.92387953251128675613f, 0.88192126434835502971f,
                           0.83146961230254523708f, 0.77301045336273696081f, 0.70710678118654752440f, 0.63439328416364549822f, 0.55557023301960222474f,
                           0.47139673682599764856f, 0.38268343236508977173f, 0.29028467725446236764f, 0.19509032201612826785f, 0.09801714032956060199f, 0.f};
  return e <= 16 ? q[e] : -q[32 - e];
}
__host__ __device__ constexpr float lrh_sin64(int e) { return e <= 16 ? lrh_cos64(16 - e) : lrh_cos64(e - 16); }

// R-point DFT of u[0..R) in place, natural order
template <int DIR, int R> struct Dft;
template <int DIR> struct Dft<DIR, 2> { __device__ __forceinline__ static void run(float2 *u) { bfly2<DIR>(u[0], u[1]); } };
template <int DIR> struct Dft<DIR, 4> { __device__ __forceinline__ static void run(float2 *u) { bfly4<DIR>(u[0], u[1], u[2], u[3]); } };
template <int DIR> struct Dft<DIR, 8> {
  __device__ __forceinline__ static void run(float2 *u)
  {
    // s = 4a+b: radix-2 over a, twiddle w8^(b c), radix-4 over b; output index c + 2d
    float2 y[8];
#pragma unroll
    for (int b = 0; b < 4; b++) { float2 p = u[b], q = u[b + 4]; y[b] = cadd(p, q); y[4 + b] = csub(p, q); }   // y[c*4+b]
    y[4 + 1] = cmulc(y[4 + 1], make_float2(LRH_C8, DIR < 0 ? -LRH_C8 : LRH_C8));
    y[4 + 2] = mulj<DIR>(y[4 + 2]);
    y[4 + 3] = cmulc(y[4 + 3], make_float2(-LRH_C8, DIR < 0 ? -LRH_C8 : LRH_C8));
#pragma unroll
    for (int c = 0; c < 2; c++) {
      bfly4<DIR>(y[c * 4 + 0], y[c * 4 + 1], y[c * 4 + 2], y[c * 4 + 3]);
#pragma unroll
      for (int d = 0; d < 4; d++) u[c + 2 * d] = y[c * 4 + d];
    }
  }
};
template <int DIR> struct Dft<DIR, 16> {
  __device__ __forceinline__ static void run(float2 *uf)
  {
    // s = 4a+b: radix-4 over a (b fixed), twiddle w16^(b c), radix-4 over b; output index c + 4d
    lrh_v2f u[16];
#pragma unroll
    for (int q = 0; q < 16; q++) u[q] = to_v(uf[q]);
#pragma unroll
    for (int b = 0; b < 4; b++) bfly4_v<DIR>(u[b], u[b + 4], u[b + 8], u[b + 12]);   // u[4c+b] = y[b][c]
    constexpr float sg = DIR < 0 ? -1.f : 1.f;
    u[4 * 1 + 1] = cmulc_v(u[4 * 1 + 1], LRH_C16A, sg * LRH_S16A);                   // w16^1
    u[4 * 1 + 2] = cmulc_v(u[4 * 1 + 2], LRH_C8, sg * LRH_C8);                       // w16^2
    u[4 * 1 + 3] = cmulc_v(u[4 * 1 + 3], LRH_S16A, sg * LRH_C16A);                   // w16^3
    u[4 * 2 + 1] = cmulc_v(u[4 * 2 + 1], LRH_C8, sg * LRH_C8);                       // w16^2
    u[4 * 2 + 2] = cmulc_v(u[4 * 2 + 2], 0.f, sg);                                   // w16^4 = +-j
    u[4 * 2 + 3] = cmulc_v(u[4 * 2 + 3], -LRH_C8, sg * LRH_C8);                      // w16^6
    u[4 * 3 + 1] = cmulc_v(u[4 * 3 + 1], LRH_S16A, sg * LRH_C16A);                   // w16^3
    u[4 * 3 + 2] = cmulc_v(u[4 * 3 + 2], -LRH_C8, sg * LRH_C8);                      // w16^6
    u[4 * 3 + 3] = cmulc_v(u[4 * 3 + 3], -LRH_C16A, -sg * LRH_S16A);                 // w16^9
#pragma unroll
    for (int c = 0; c < 4; c++) {
      bfly4_v<DIR>(u[4 * c + 0], u[4 * c + 1], u[4 * c + 2], u[4 * c + 3]);          // over b -> d
#pragma unroll
      for (int d = 0; d < 4; d++) uf[c + 4 * d] = to_f2(u[4 * c + d]);
    }
  }
};

// 8-point DFT on native 2-vectors: y[0..8) natural in, out[c + 2 d] (natural order) -- s = 4a + b: radix 2 over a, w8^(b c), radix 4 over b
template <int DIR> __device__ __forceinline__ void dft8_v(lrh_v2f (&y)[8])
{
  constexpr float sg = DIR < 0 ? -1.f : 1.f;
  lrh_v2f t[8];
#pragma unroll
  for (int b = 0; b < 4; b++) { t[b] = y[b] + y[b + 4]; t[4 + b] = y[b] - y[b + 4]; }   // t[4c + b]
  t[4 + 1] = cmulc_v(t[4 + 1], LRH_C8, sg * LRH_C8);
  t[4 + 2] = cmulc_v(t[4 + 2], 0.f, sg);
  t[4 + 3] = cmulc_v(t[4 + 3], -LRH_C8, sg * LRH_C8);
#pragma unroll
  for (int c = 0; c < 2; c++) {
    bfly4_v<DIR>(t[4 * c + 0], t[4 * c + 1], t[4 * c + 2], t[4 * c + 3]);
#pragma unroll
    for (int d = 0; d < 4; d++) y[c + 2 * d] = t[4 * c + d];
  }
}
// 32-point DFT in registers: s = 8a + b (a < 4, b < 8): radix 4 over a (b fixed), twiddle w32^(b c), radix 8 over b; output index c + 4 d
template <int DIR> struct Dft<DIR, 32> {
  __device__ __forceinline__ static void run(float2 *uf)
  {
    constexpr float sg = DIR < 0 ? -1.f : 1.f;
    lrh_v2f u[32];
#pragma unroll
    for (int q = 0; q < 32; q++) u[q] = to_v(uf[q]);
#pragma unroll
    for (int b = 0; b < 8; b++) bfly4_v<DIR>(u[b], u[b + 8], u[b + 16], u[b + 24]);          // u[8c + b] = y[b][c]
#pragma unroll
    for (int c = 1; c < 4; c++)
#pragma unroll
      for (int b = 1; b < 8; b++) {
        const int e = (b * c) & 31;
        if (e == 8) u[8 * c + b] = cmulc_v(u[8 * c + b], 0.f, sg);
        else if (e == 16) u[8 * c + b] = -u[8 * c + b];
        else u[8 * c + b] = cmulc_v(u[8 * c + b], lrh_cos32(e), sg * lrh_sin32(e));
      }
#pragma unroll
    for (int c = 0; c < 4; c++) {
      lrh_v2f y[8];
#pragma unroll
      for (int b = 0; b < 8; b++) y[b] = u[8 * c + b];
      dft8_v<DIR>(y);
#pragma unroll
      for (int d = 0; d < 8; d++) uf[c + 4 * d] = to_f2(y[d]);
    }
  }
};

// ---- staged DFTs in registers (k_fft1v): R = NC x ND.  a(): first stage in place -- afterwards u[ND c + b] holds the input of the second
// stage; b(u, c, y): second stage of group c, y[d] = output number out(c, d).  in(i): the order in which a() consumes its inputs, so that a
// caller that loads them in that order can start on the first butterfly while the last loads are still on their way; the second stage hands
// out ND finished values at a time, which the caller can store while the next group is being computed.
template <int DIR, int R> struct SDft;
template <int DIR> struct SDft<DIR, 32> {
  static constexpr int NC = 4, ND = 8;
  __host__ __device__ static constexpr int in(int i) { return (i >> 2) + 8 * (i & 3); }
  __host__ __device__ static constexpr int out(int c, int d) { return c + 4 * d; }
  __device__ __forceinline__ static void a(lrh_v2f (&u)[32])
  {
    constexpr float sg = DIR < 0 ? -1.f : 1.f;
#pragma unroll
    for (int b = 0; b < 8; b++) {
      bfly4_v<DIR>(u[b], u[b + 8], u[b + 16], u[b + 24]);
#pragma unroll
      for (int c = 1; c < 4; c++) {
        const int e = (b * c) & 31;
        if (e == 0) continue;
        if (e == 8) u[8 * c + b] = cmulc_v(u[8 * c + b], 0.f, sg);
        else u[8 * c + b] = cmulc_v(u[8 * c + b], lrh_cos32(e), sg * lrh_sin32(e));
      }
    }
  }
  __device__ __forceinline__ static void b(const lrh_v2f (&u)[32], int c, lrh_v2f (&y)[8])
  {
#pragma unroll
    for (int i = 0; i < 8; i++) y[i] = u[8 * c + i];
    dft8_v<DIR>(y);
  }
};
template <int DIR> struct SDft<DIR, 16> {
  static constexpr int NC = 4, ND = 4;
  __host__ __device__ static constexpr int in(int i) { return (i >> 2) + 4 * (i & 3); }
  __host__ __device__ static constexpr int out(int c, int d) { return c + 4 * d; }
  __device__ __forceinline__ static void a(lrh_v2f (&u)[16])
  {
    constexpr float sg = DIR < 0 ? -1.f : 1.f;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      bfly4_v<DIR>(u[b], u[b + 4], u[b + 8], u[b + 12]);
#pragma unroll
      for (int c = 1; c < 4; c++) {
        const int e = 2 * ((b * c) & 15);                  // w16^(b c) in 32nds of a turn
        if (e == 0) continue;
        if (e == 8) u[4 * c + b] = cmulc_v(u[4 * c + b], 0.f, sg);
        else u[4 * c + b] = cmulc_v(u[4 * c + b], lrh_cos32(e), sg * lrh_sin32(e));
      }
    }
  }
  __device__ __forceinline__ static void b(const lrh_v2f (&u)[16], int c, lrh_v2f (&y)[4])
  {
#pragma unroll
    for (int i = 0; i < 4; i++) y[i] = u[4 * c + i];
    bfly4_v<DIR>(y[0], y[1], y[2], y[3]);
  }
};
template <int DIR> struct SDft<DIR, 8> {
  static constexpr int NC = 2, ND = 4;
  __host__ __device__ static constexpr int in(int i) { return (i >> 1) + 4 * (i & 1); }
  __host__ __device__ static constexpr int out(int c, int d) { return c + 2 * d; }
  __device__ __forceinline__ static void a(lrh_v2f (&u)[8])
  {
    constexpr float sg = DIR < 0 ? -1.f : 1.f;
#pragma unroll
    for (int b = 0; b < 4; b++) { const lrh_v2f p = u[b], q = u[b + 4]; u[b] = p + q; u[4 + b] = p - q; }
    u[4 + 1] = cmulc_v(u[4 + 1], LRH_C8, sg * LRH_C8);
    u[4 + 2] = cmulc_v(u[4 + 2], 0.f, sg);
    u[4 + 3] = cmulc_v(u[4 + 3], -LRH_C8, sg * LRH_C8);
  }
  __device__ __forceinline__ static void b(const lrh_v2f (&u)[8], int c, lrh_v2f (&y)[4])
  {
#pragma unroll
    for (int i = 0; i < 4; i++) y[i] = u[4 * c + i];
    bfly4_v<DIR>(y[0], y[1], y[2], y[3]);
  }
};
template <int DIR> struct SDft<DIR, 4> {
  static constexpr int NC = 1, ND = 4;
  __host__ __device__ static constexpr int in(int i) { return i; }
  __host__ __device__ static constexpr int out(int c, int d) { return d; }
  __device__ __forceinline__ static void a(lrh_v2f (&u)[4]) { bfly4_v<DIR>(u[0], u[1], u[2], u[3]); }
  __device__ __forceinline__ static void b(const lrh_v2f (&u)[4], int, lrh_v2f (&y)[4]) { y[0] = u[0]; y[1] = u[1]; y[2] = u[2]; y[3] = u[3]; }
};

// ---- compile-time pass plan -------------------------------------------------------------
// points per thread: 16 from N = 1024 up, 4 below.
// Measured on MI355X (round 1): 32 points/thread + half-round exchanges at N = 16384 (512 threads, 68 KiB LDS, two
// workgroups per CU) needs > 128 VGPRs, spills ~130 registers and runs 1.7-2.2x SLOWER than 16 points x 1024
// threads with one workgroup per CU; the half-round path stays available (fft_halves) but is not selected.
#ifndef LRH_P14
#define LRH_P14 16          /* points per thread at N = 16384: 16 (1024 threads, 128 VGPRs) or 32 (512 threads, 256 VGPRs) */
#endif
__host__ __device__ constexpr int points_per_thread(int log2n) { return log2n >= 10 ? 16 : 4; }
// k_fft1 / k_timf2 (BlockFftL, one persistent workgroup per CU at N = 16384)
__host__ __device__ constexpr int points_fft1(int log2n) { return log2n == 14 ? LRH_P14 : points_per_thread(log2n); }
__host__ __device__ constexpr int fft1_threads(int log2n) { return (1 << log2n) / points_fft1(log2n); }
__host__ __device__ constexpr int fft_halves(int log2n) { return 1; }
__host__ __device__ constexpr int fft_threads(int log2n) { return (1 << log2n) / points_per_thread(log2n); }
// waves per SIMD to ask for in __launch_bounds__ (N = 8192: two 512-thread workgroups per CU)
__host__ __device__ constexpr int fft_min_waves(int log2n) { return log2n == 13 ? 2 : (log2n == 12 ? 3 : 1); }

template <int LOG2N, int P> struct FftPlan {
  static constexpr int N = 1 << LOG2N;
  static constexpr int T = N / P;                       // threads per workgroup
  static constexpr int MAXLOG = (P >= 16) ? 4 : (P == 8 ? 3 : 2);      // log2 of the largest radix (8 points per thread: radix 8, e.g. 256 = 8 x 8 x 4 in three passes)
  static constexpr int FULL = LOG2N / MAXLOG;
  static constexpr int REM = LOG2N % MAXLOG;
  static constexpr int NPASS = FULL + (REM ? 1 : 0);
  static constexpr int HALVES = fft_halves(LOG2N);
  __host__ __device__ static constexpr int radix(int pass) { return pass < FULL ? (1 << MAXLOG) : (1 << REM); }
  __host__ __device__ static constexpr int done(int pass) { int p = 1; for (int i = 0; i < pass; i++) p *= radix(i); return p; }
  static constexpr int R0 = FULL > 0 ? (1 << MAXLOG) : (1 << REM);
  static constexpr int RL = REM ? (1 << REM) : (1 << MAXLOG);
  static constexpr int LDS_CELLS = (N + (N >> 4)) / HALVES;   // float2 cells incl. padding
  static_assert(P % R0 == 0 && P % RL == 0, "P must be a multiple of every radix");
  static_assert(HALVES == 1 || ((P / R0) % 2 == 0 && (P / RL) % 2 == 0), "half-round exchange needs an even butterfly count");
};

__device__ __forceinline__ int lds_pad(int idx) { return idx + (idx >> 4); }

// Register layout contract of BlockFft::run:
//   in : x[m*R0 + s] = input[(tid + m*T) + s*(N/R0)]      m < P/R0, s < R0
//   out: x[m*RL + q] = output[(tid + m*T) + q*(N/RL)]     m < P/RL, q < RL
// tw[m] = exp(-2 pi j m / N), m in [0, N) (forward table; DIR=+1 conjugates it).
// `lds` must hold FftPlan::LDS_CELLS float2. The caller must __syncthreads() before reusing lds.
//
// Half-round exchange (HALVES = 2): output cell p*(a*R+q)+k of butterfly i = a*p+k lies in the lower half of the
// sequence exactly when i < N/(2R), i.e. for the first half of every thread's butterflies, and the next pass reads
// cell i' + s*N/R' from the lower half exactly for s < R'/2.  So the redistribution splits into two independent
// rounds that each move N/2 cells through the same N/2-cell buffer.
// WAVE: the T threads of a transform are the lanes of ONE wave (T <= 64, contiguous lanes): the exchange needs no workgroup barrier --
// a wave's LDS operations execute in order -- only a fence against the compiler's reordering, and the waves of a workgroup that runs
// several transforms side by side stop marching in lock step.
template <int LOG2N, int P, int DIR, bool WAVE = false> struct BlockFft {
  using Plan = FftPlan<LOG2N, P>;
  __device__ __forceinline__ static void sync() { if constexpr (WAVE) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } else __syncthreads(); }
  static constexpr int N = Plan::N, T = Plan::T, HALVES = Plan::HALVES;

  // twiddles of one pass, fetched before the LDS exchange that feeds it so their L2 latency hides behind the exchange
  template <int PASS> struct Tw {
    static constexpr int R = Plan::radix(PASS);
    static constexpr int PER = R == 16 ? 6 : (R == 8 ? 4 : R - 1);
    static constexpr int NB = P / R;
    float2 w[PER * NB > 0 ? PER * NB : 1];
  };
  template <int PASS> __device__ __forceinline__ static void load_tw(Tw<PASS> &t, const float2 *__restrict__ tw, int tid)
  {
    constexpr int R = Plan::radix(PASS);
    constexpr int p = Plan::done(PASS);
    constexpr int NB = P / R, PER = Tw<PASS>::PER;
    if constexpr (p > 1) {
#pragma unroll
      for (int m = 0; m < NB; m++) {
        const int i = tid + m * T;
        const int k = i & (p - 1);
        const int base = k * (N / (p * R));
        if constexpr (R == 16) {
#pragma unroll
          for (int b = 1; b < 4; b++) { t.w[m * PER + b - 1] = tw_dir<DIR>(tw[b * base]); t.w[m * PER + 2 + b] = tw_dir<DIR>(tw[4 * b * base]); }
        } else if constexpr (R == 8) {
#pragma unroll
          for (int b = 1; b < 4; b++) t.w[m * PER + b - 1] = tw_dir<DIR>(tw[b * base]);
          t.w[m * PER + 3] = tw_dir<DIR>(tw[4 * base]);
        } else {
#pragma unroll
          for (int s = 1; s < R; s++) t.w[m * PER + s - 1] = tw_dir<DIR>(tw[s * base]);
        }
      }
    }
  }

  template <int PASS> __device__ __forceinline__ static void pass(float2 (&x)[P], float2 *lds, const float2 *__restrict__ tw, int tid, const Tw<PASS> &t)
  {
    constexpr int R = Plan::radix(PASS);
    constexpr int p = Plan::done(PASS);
    constexpr int NB = P / R, PER = Tw<PASS>::PER;       // butterflies per thread
#pragma unroll
    for (int m = 0; m < NB; m++) {
      float2 *u = &x[m * R];
      if constexpr (p > 1) {
        if constexpr (R == 16) {
          // w^(4a+b) = w^(4a) * w^b: six table gathers instead of fifteen, one extra rounding
#pragma unroll
          for (int s = 1; s < 16; s++) {
            const int a = s >> 2, b = s & 3;
            const float2 w = a == 0 ? t.w[m * PER + b - 1] : (b == 0 ? t.w[m * PER + 2 + a] : cmul(t.w[m * PER + 2 + a], t.w[m * PER + b - 1]));
            u[s] = cmul(u[s], w);
          }
        } else if constexpr (R == 8) {
#pragma unroll
          for (int s = 1; s < 8; s++) {
            const float2 w = s < 4 ? t.w[m * PER + s - 1] : (s == 4 ? t.w[m * PER + 3] : cmul(t.w[m * PER + 3], t.w[m * PER + s - 5]));
            u[s] = cmul(u[s], w);
          }
        } else {
#pragma unroll
          for (int s = 1; s < R; s++) u[s] = cmul(u[s], t.w[m * PER + s - 1]);
        }
      }
      Dft<DIR, R>::run(u);
    }
    if constexpr (PASS + 1 < Plan::NPASS) {
      constexpr int R2 = Plan::radix(PASS + 1);
      constexpr int NB2 = P / R2;
      Tw<PASS + 1> tn;
      load_tw<PASS + 1>(tn, tw, tid);
      float2 xn[P];
#pragma unroll
      for (int h = 0; h < HALVES; h++) {
        if (PASS > 0 || h > 0) sync();                  // reads of the previous round are done
#pragma unroll
        for (int m = h * NB / HALVES; m < (h + 1) * NB / HALVES; m++) {
          const int i = tid + m * T;
          const int k = i & (p - 1);
          const int j = (i - k) * R + k - h * (N / 2);
#pragma unroll
          for (int q = 0; q < R; q++) lds[lds_pad(j + q * p)] = x[m * R + q];
        }
        sync();
#pragma unroll
        for (int m = 0; m < NB2; m++) {
          const int i = tid + m * T - h * (N / 2);
#pragma unroll
          for (int s = h * R2 / HALVES; s < (h + 1) * R2 / HALVES; s++) xn[m * R2 + s] = lds[lds_pad(i + s * (N / R2))];
        }
      }
#pragma unroll
      for (int e = 0; e < P; e++) x[e] = xn[e];
      pass<PASS + 1>(x, lds, tw, tid, tn);
    }
  }

  __device__ __forceinline__ static void run(float2 (&x)[P], float2 *lds, const float2 *__restrict__ tw, int tid) { Tw<0> t0; pass<0>(x, lds, tw, tid, t0); }
};

// ---------------------------------------------------------------------------------------------------------
// BlockFftL: the same transform with every twiddle kept on chip, for the persistent kernels (k_fft1, k_timf2).
//
// Why: vmcnt retires loads, stores and atomics in issue order, so a twiddle gather issued after the output stores
// of the previous transform cannot return before those stores have drained to L2/HBM -- at one workgroup per CU
// that drain is fully exposed.  Here the per-pass tables live in LDS behind the exchange buffer (filled once per
// workgroup).  The last pass of N >= 8192 (k = butterfly index, a full table would not fit) reads one cell per
// thread, w^tid, squares/cubes it and applies compile-time 16th roots of unity for the butterflies tid + m T.
// All LDS addresses are one per-thread base plus constants.  run() may be called again without a barrier in between.
//   table of pass with completed length p, radix R:  cell[e*p + k] = w^(mult(e) k N/(pR)),  k < p
//   mult = {1,2,3,4,8,12} (R=16), {1,2,3,4} (R=8), {1..R-1} otherwise -- the factors pass() combines.
// CONJTAB: the tables in LDS were filled by the transform of the OTHER direction (BlockFftL<LOG2N, P, -DIR>::init) and are
// conjugated as they are applied -- a kernel that runs forward and back transforms (k_fft1w) keeps one set of tables.
template <int LOG2N, int P, int DIR, bool CONJTAB = false> struct BlockFftL {
  using Plan = FftPlan<LOG2N, P>;
  __device__ __forceinline__ static float2 twmul(float2 u, float2 w) { return CONJTAB ? cmul_conj(u, w) : cmul(u, w); }
  static constexpr int N = Plan::N, T = Plan::T, NPASS = Plan::NPASS;
  static_assert(Plan::HALVES == 1, "full-round exchange only");
  __host__ __device__ static constexpr int per(int R) { return R == 16 ? 6 : (R == 8 ? 4 : R - 1); }
  static constexpr bool ROOT_LAST = NPASS > 1 && Plan::done(NPASS - 1) >= 4096;
  static constexpr int NTAB = NPASS - (ROOT_LAST ? 1 : 0);         // passes 1 .. NTAB-1 read a full LDS table
  __host__ __device__ static constexpr int tab_off(int pass) { int o = 0; for (int i = 1; i < pass; i++) o += Plan::done(i) * per(Plan::radix(i)); return o; }
  static constexpr int TW_CELLS = tab_off(NTAB) + (ROOT_LAST ? T : 0) + 1;
  static constexpr int LDS_CELLS = Plan::LDS_CELLS + TW_CELLS;     // exchange buffer, then the tables
  static_assert(!ROOT_LAST || ((P == 16 || P == 32) && Plan::RL <= 4), "last-pass roots: P-th roots of unity, powers up to 3");

  template <int PASS> __device__ __forceinline__ static void init_tab(float2 *twl, const float2 *__restrict__ tw, int tid)
  {
    if constexpr (PASS < NTAB) {
      constexpr int R = Plan::radix(PASS), p = Plan::done(PASS), PER = per(R), stride = N / (p * R);
      for (int idx = tid; idx < p * PER; idx += T) {
        const int e = idx / p, k = idx & (p - 1);
        const int mult = R == 16 ? (e < 3 ? e + 1 : 4 * (e - 2)) : (R == 8 ? (e < 3 ? e + 1 : 4) : e + 1);
        twl[tab_off(PASS) + idx] = tw_dir<DIR>(tw[mult * k * stride]);
      }
      init_tab<PASS + 1>(twl, tw, tid);
    }
  }
  // once per workgroup; the caller must __syncthreads() before the first run()
  __device__ __forceinline__ static void init(float2 *lds, const float2 *__restrict__ tw, int tid)
  {
    init_tab<1>(lds + Plan::LDS_CELLS, tw, tid);
    if constexpr (ROOT_LAST) lds[Plan::LDS_CELLS + tab_off(NTAB) + tid] = tw_dir<DIR>(tw[tid]);
  }

  template <int E> __device__ __forceinline__ static float2 mul_root(float2 v)       // v * exp(DIR 2 pi j E / P), E in units of 1/32 turn
  {
    constexpr int e = E & 31;
    if constexpr (e == 0) return v;
    else if constexpr (e == 8) return mulj<DIR>(v);
    else if constexpr (e == 16) return make_float2(-v.x, -v.y);
    else if constexpr (e == 24) return mulj<-DIR>(v);
    else return cmulc(v, make_float2(lrh_cos32(e), DIR < 0 ? -lrh_sin32(e) : lrh_sin32(e)));
  }
  // butterfly M of the root pass: element S gets w^(S (tid + M T)) = w^(S tid) * (16th root)^(S M)
  template <int M, int R, int NB> __device__ __forceinline__ static void root_pass(float2 *x, float2 w1, float2 w2, float2 w3)
  {
    if constexpr (M < NB) {
      float2 *u = &x[M * R];
      constexpr int U = 32 / P;                         // butterfly M sits M T = M N/P bins further: M/P of a turn per unit of S
      u[1] = cmul(u[1], mul_root<1 * M * U>(w1));
      if constexpr (R > 2) {
        u[2] = cmul(u[2], mul_root<2 * M * U>(w2));
        u[3] = cmul(u[3], mul_root<3 * M * U>(w3));
      }
      Dft<DIR, R>::run(u);
      root_pass<M + 1, R, NB>(x, w1, w2, w3);
    }
  }

  struct NoHook { __device__ __forceinline__ void operator()() const {} };
  // `before_last` runs once, after the LDS reads that feed the last pass and before its butterflies: the place to
  // issue the global loads the caller needs right after the transform (their latency hides behind the last pass)
  template <int PASS, class Hook> __device__ __forceinline__ static void pass(float2 (&x)[P], float2 *lds, int tid, const Hook &before_last)
  {
    constexpr int R = Plan::radix(PASS);
    constexpr int p = Plan::done(PASS);
    constexpr int NB = P / R, PER = per(R);
    if constexpr (PASS == NPASS - 1) before_last();
    if constexpr (ROOT_LAST && PASS == NPASS - 1) {
      int tt_ = tid;
      asm volatile("" : "+v"(tt_));
      float2 w1 = lds[Plan::LDS_CELLS + tab_off(NTAB) + tt_];         // w^tid, then its square and cube
      if constexpr (CONJTAB) w1.y = -w1.y;
      const float2 w2 = cmul(w1, w1);
      const float2 w3 = cmul(w2, w1);
      root_pass<0, R, NB>(x, w1, w2, w3);
    } else {
      int tt_ = tid;
      asm volatile("" : "+v"(tt_));                                   // see the exchange below
#pragma unroll
      for (int m = 0; m < NB; m++) {
        float2 *u = &x[m * R];
        if constexpr (p > 1) {
          const float2 *tab = lds + Plan::LDS_CELLS + tab_off(PASS) + ((tt_ + m * T) & (p - 1));
          float2 w[PER];
#pragma unroll
          for (int e = 0; e < PER; e++) w[e] = tab[e * p];
          if constexpr (R == 16) {
#pragma unroll
            for (int s = 1; s < 16; s++) {
              const int a = s >> 2, b = s & 3;
              u[s] = twmul(u[s], a == 0 ? w[b - 1] : (b == 0 ? w[2 + a] : cmul(w[2 + a], w[b - 1])));
            }
          } else if constexpr (R == 8) {
#pragma unroll
            for (int s = 1; s < 8; s++) u[s] = twmul(u[s], s < 4 ? w[s - 1] : (s == 4 ? w[3] : cmul(w[3], w[s - 5])));
          } else {
#pragma unroll
            for (int s = 1; s < R; s++) u[s] = twmul(u[s], w[s - 1]);
          }
        }
        Dft<DIR, R>::run(u);
      }
    }
    if constexpr (PASS + 1 < NPASS) {
      constexpr int R2 = Plan::radix(PASS + 1);
      constexpr int NB2 = P / R2;
      static_assert(p <= T && (T * R) % 16 == 0 && (N / R2) % 16 == 0 && T % 16 == 0, "linear padded addressing");
      constexpr int QS = p >= 16 ? p + (p >> 4) : p;                 // padded distance between a butterfly's outputs
      if (PASS > 0) __syncthreads();                                  // reads of the previous exchange are done
      // Fresh opaque copies of the thread index per exchange: LDS offsets beyond the 64 KiB immediate range need
      // their own address registers, and common subexpressions shared between passes (or between the two
      // transforms of k_timf2) would otherwise stay live across the butterflies and spill.
      int tw_ = tid, tr_ = tid;
      asm volatile("" : "+v"(tw_));
      {
        const int k = tw_ & (p - 1);
        float2 *wr = lds + lds_pad((tw_ - k) * R + k);
#pragma unroll
        for (int m = 0; m < NB; m++)
#pragma unroll
          for (int q = 0; q < R; q++) wr[m * (T * R + T * R / 16) + q * QS] = x[m * R + q];
      }
      __syncthreads();
      asm volatile("" : "+v"(tr_));
      const float2 *rd = lds + lds_pad(tr_);
#pragma unroll
      for (int m = 0; m < NB2; m++)
#pragma unroll
        for (int s = 0; s < R2; s++) {
          const int c = m * T + s * (N / R2);
          x[m * R2 + s] = rd[c + c / 16];
        }
      // The barrier that protects the exchange buffer for the NEXT transform sits here, right behind the reads of the
      // last exchange, where the waves were aligned a moment ago by the write->read barrier: the caller needs no
      // barrier at the end of its loop, where the waves arrive spread by a whole pass of VALU contention
      // (measured 6-7k of 27k cycles per transform in k_fft1).
      if constexpr (PASS + 2 == NPASS) __syncthreads();
      pass<PASS + 1>(x, lds, tid, before_last);
    }
  }

  __device__ __forceinline__ static void run(float2 (&x)[P], float2 *lds, int tid) { pass<0>(x, lds, tid, NoHook()); }
  template <class Hook> __device__ __forceinline__ static void run(float2 (&x)[P], float2 *lds, int tid, const Hook &before_last) { pass<0>(x, lds, tid, before_last); }
};

// XCD-aware block order (8 XCDs, blocks dealt round-robin): consecutive work items go to blocks b, b+8, b+16 ...
// so neighbours in time (which share half their input) run on the same XCD and meet in its L2.  Speed only.
__device__ __forceinline__ int xcd_order(int b, int nb) { return (nb & 7) ? b : (b & 7) * (nb >> 3) + (b >> 3); }

}  // namespace lrh
