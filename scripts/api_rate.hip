// scripts/api_rate.hip -- diagnostics: what a HIP API call costs when several host threads enqueue on their own streams, with and without
// other threads waiting inside hipEventSynchronize / on a host-visible flag.  hipcc --offload-arch=gfx950 -O2 -o api_rate api_rate.hip -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <pthread.h>
#include <atomic>
#include <chrono>
#include <vector>
__global__ void k_nop(int *p) { if (p && threadIdx.x == 1234567) *p = 1; }
__global__ void k_flag(volatile unsigned *f, unsigned v) { if (threadIdx.x == 0) { *f = v; __threadfence_system(); } }
static std::atomic<bool> stop{false};
static int mode = 0;   // waiter mode: 0 none, 1 hipEventSynchronize loop, 2 host flag spin, 3 hipEventQuery loop
struct W { int id; double us_per_call; long calls; };
static void *enq(void *a)
{
  W *w = (W *)a; hipSetDevice(0);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming);
  const int N = 20000;
  for (int i = 0; i < 200; i++) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, nullptr); hipEventRecord(e, s); }
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; i++) {
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, nullptr); hipEventRecord(e, s);
    if ((i & 63) == 63) hipStreamSynchronize(s);       // keep the queue short
  }
  hipStreamSynchronize(s);
  w->us_per_call = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (2.0 * N); w->calls = 2L * N;
  return nullptr;
}
static void *waiter(void *a)
{
  hipSetDevice(0);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming);
  volatile unsigned *flag; hipHostMalloc((void **)&flag, 64); *flag = 0;
  unsigned seq = 0; long n = 0; double lat = 0;
  while (!stop) {
    seq++;
    auto t0 = std::chrono::steady_clock::now();
    if (mode == 2) { hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, flag, seq); while (*flag != seq && !stop) { } }
    else { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, nullptr); hipEventRecord(e, s);
      if (mode == 1) hipEventSynchronize(e); else while (hipEventQuery(e) != hipSuccess && !stop) { } }
    lat += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); n++;
  }
  *(double *)a = n ? lat / n : 0;
  return nullptr;
}
int main(int argc, char **argv)
{
  const int nenq = argc > 1 ? atoi(argv[1]) : 3, nwait = argc > 2 ? atoi(argv[2]) : 0; mode = argc > 3 ? atoi(argv[3]) : 0;
  hipSetDevice(0); hipFree(0);
  std::vector<pthread_t> te(nenq), tw(nwait); std::vector<W> w(nenq); std::vector<double> wl(nwait);
  for (int i = 0; i < nwait; i++) pthread_create(&tw[i], nullptr, waiter, &wl[i]);
  for (int i = 0; i < nenq; i++) { w[i].id = i; pthread_create(&te[i], nullptr, enq, &w[i]); }
  double tot = 0;
  for (int i = 0; i < nenq; i++) { pthread_join(te[i], nullptr); tot += 1.0 / w[i].us_per_call; }
  stop = true;
  for (int i = 0; i < nwait; i++) pthread_join(tw[i], nullptr);
  printf("enqueuers %d waiters %d mode %d: us/call per thread", nenq, nwait, mode);
  for (int i = 0; i < nenq; i++) printf(" %.2f", w[i].us_per_call);
  printf(" | aggregate %.0f k calls/s | waiter round trip us:", 1e3 * tot);
  for (int i = 0; i < nwait; i++) printf(" %.1f", wl[i]);
  printf("\n");
  return 0;
}
