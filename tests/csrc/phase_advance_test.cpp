#include "lrh_phase.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
int main() {
  std::mt19937_64 rng(987654321);
  std::uniform_real_distribution<double> U(-1, 1);
  long bad = 0, cases = 0;
  for (int it = 0; it < 400000; it++) {
    float t, d; int n;
    const int mode = it % 8;
    if (mode == 0) { t = (float)(3.14159265 * U(rng)); d = (float)(3.14159265 * U(rng)); n = 1 + (int)(rng() % 2048); }
    else if (mode == 1) { t = (float)(1000 * U(rng)); d = (float)(0.01 * U(rng)); n = 1 + (int)(rng() % 4096); }
    else if (mode == 2) { t = 0.f; d = (float)(U(rng)); n = 1 + (int)(rng() % 1024); }
    else if (mode == 3) { t = (float)(U(rng)); d = -t / (float)(1 + rng() % 64); n = 1 + (int)(rng() % 256); }   // runs through zero
    else if (mode == 4) { t = std::ldexp(1.f, (int)(rng() % 20) - 5); d = std::ldexp(1.f, (int)(rng() % 30) - 28) * (float)(1 + rng() % 7) * 0.5f; n = 1 + (int)(rng() % 512); }  // ties
    else if (mode == 5) { t = (float)(1e6 * U(rng)); d = (float)(3 * U(rng)); n = 1 + (int)(rng() % 512); }
    else if (mode == 6) { t = (float)(3 * U(rng)); d = (float)(1e-6 * U(rng)); n = 1 + (int)(rng() % 100000); }
    else { t = (float)(50 * U(rng)); d = (float)(2 * 3.14159265 * U(rng)); n = 64 * (1 + (int)(rng() % 16)); }
    volatile float r = t; for (int i = 0; i < n; i++) r = r + d;
    const float g = lrh_phase_advance(t, d, n);
    float rr = r; cases++;
    if (std::memcmp(&g, &rr, 4)) { if (bad < 10) std::printf("mismatch mode %d t %a d %a n %d: loop %a jump %a\n", mode, t, d, n, rr, g); bad++; }
  }
  std::printf("%ld cases, %ld mismatches\n", cases, bad);
  return bad != 0;
}
