// Device copy rate by access width and store policy: what the streaming kernels of this repository can hope for.
// build: hipcc --offload-arch=gfx950 -O3 scripts/copy_bw.hip -o gpurun_out/copy_bw ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, bool NT> __global__ __launch_bounds__(256) void k_copy(const T *__restrict__ src, T *__restrict__ dst, size_t n)
{
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    if constexpr (NT) __builtin_nontemporal_store(src[i], &dst[i]); else dst[i] = src[i];
  }
}
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
template <typename T, bool NT> static void run(const char *name, void *a, void *b, size_t bytes)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t n = bytes / sizeof(T);
  for (int grid : { 2048, 8192, 32768 }) {
    hipLaunchKernelGGL((k_copy<T, NT>), dim3(grid), dim3(256), 0, 0, (const T *)a, (T *)b, n);
    hipEventRecord(e0);
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL((k_copy<T, NT>), dim3(grid), dim3(256), 0, 0, (const T *)a, (T *)b, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s grid %6d  %.2f TB/s (read + write)\n", name, grid, 10 * 2.0 * bytes / (ms * 1e-3) / 1e12);
  }
}
int main()
{
  const size_t bytes = (size_t)1 << 30;
  void *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
  run<float, false>("4 B/lane", a, b, bytes); run<v2, false>("8 B/lane", a, b, bytes); run<v4, false>("16 B/lane", a, b, bytes);
  run<v2, true>("8 B/lane nt store", a, b, bytes); run<v4, true>("16 B/lane nt store", a, b, bytes);
  return 0;
}
