// lrh_phase.h -- the running phase of do_mix1 (mix1.c:141-195: `t1 += t2` once per output sample, in float) advanced by many samples at once,
// bit for bit.  While t stays inside one binade [2^e, 2^(e+1)) it is a multiple of that binade's ulp u, and fl(t + d) = t + rn_u(d): the increment
// rounded to a multiple of u (to nearest; a tie -- d an odd multiple of u/2 -- would depend on t's parity and is left to the single step).  So k steps
// inside the binade are one multiplication; only the steps that cross a binade boundary, start from zero or hit a tie are real float additions.
// lrh_host.hip (mix1_run) needs the phase at every LRH_PH_CHUNK-th sample and at the end of a transform: ~20 jumps instead of 1024 additions per
// transform (0.23 ms of host time per round of 1024 transforms before).  tests/test_phase_advance_cpu.py holds it to the plain loop.
#ifndef LRH_PHASE_H
#define LRH_PHASE_H
#include <cmath>
#ifdef __HIPCC__
#define LRH_PHASE_HD __host__ __device__
#else
#define LRH_PHASE_HD
#endif

// (host and device: the host advances from transform to transform, k_phase_expand derives the chunk starts inside a transform -- the same function,
// exact operations only: scaling by powers of two, round to nearest even, comparisons)
LRH_PHASE_HD static inline float lrh_phase_advance(float t, float d, int n)
{
  while (n > 0) {
    const float at = fabsf(t);
    if (!(at >= 1.17549435e-38f) || d == 0.f || !(at <= 3.4e38f)) {           // zero, denormal, or nothing to add: plain steps
      if (d == 0.f) return t;
      t += d; n--; continue;
    }
    int e; (void)frexpf(at, &e); e -= 1;                                        // at in [2^e, 2^(e+1))
    const double u = ldexp(1.0, e - 23);
    const double q = (double)d * ldexp(1.0, 23 - e), qr = rint(q);
    if (q - floor(q) == 0.5) { t += d; n--; continue; }   // tie: the single step decides
    const double dq = qr * u;                                                   // what every addition inside this binade really adds
    if (dq == 0.0) return t;                                                    // the increment is below half an ulp: t no longer moves (until n runs out)
    const double lo = ldexp(1.0, e), hi = ldexp(1.0, e + 1);
    const double s = t < 0 ? -1.0 : 1.0, m = dq * s;                            // m > 0: |t| grows
    double kmax;
    // (one ulp of margin at either end: the exact sum of a step differs from the rounded one by up to u/2, and below 2^e the grid is finer)
    if (m > 0) kmax = ceil((hi - u - (double)at) / m) - 1;                 // largest k with at + k m < hi - u
    else kmax = floor(((double)at - lo - u) / -m);                         // largest k with at + k m >= lo + u
    if (kmax < 1) { t += d; n--; continue; }                                    // the next step leaves the binade: a real addition
    const int k = kmax < (double)n ? (int)kmax : n;
    t = (float)((double)t + k * dq);                                            // exact: a multiple of u inside the binade
    n -= k;
  }
  return t;
}
#endif
