// Diagnostic (not product): is a lone wave's store cost per instruction or per byte?  1024 waves (one per SIMD), each
// writes C column-pairs once per launch with 16 FMAs between store groups, over a 268 MB footprint (as the stepper):
//   A  two 512-B stores (global_store_dwordx2, two different columns) per group
//   B  one 1-KB store (global_store_dwordx4, the two columns interleaved per plant: 16 contiguous bytes per lane)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int MODE>
__global__ __launch_bounds__(64) void k(double *dst, size_t N, int C, unsigned long long *ticks) {
  const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
  double a = threadIdx.x, b = 1.0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int c = 0; c < C; c += 2) {
    if (MODE == 0) { dst[(size_t)c * N + p] = a; dst[(size_t)(c + 1) * N + p] = a + 1.0; }
    else { double2 v = make_double2(a, a + 1.0); *(double2 *)(dst + (size_t)c * N + 2 * p) = v; }
#pragma unroll
    for (int u = 0; u < 16; u++) a = __builtin_fma(a, 0.999, b);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (a == 123.0) dst[p] = a;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int MODE> static void run(const char *name, double *dst, size_t N, int C, unsigned long long *ticks) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<MODE>, dim3(N / 64), dim3(64), 0, 0, dst, N, C, ticks);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k<MODE>, dim3(N / 64), dim3(64), 0, 0, dst, N, C, ticks);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[1024]; CK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
  double sum = 0; for (int i = 0; i < 1024; i++) sum += (double)h[i];
  printf("%-60s %7.1f us per launch, %6.1f ticks per column pair (+16 FMAs)\n", name, ms / 10 * 1e3, sum / 1024 / (C / 2));
}
int main() {
  const size_t N = 65536; const int C = 512; double *dst; unsigned long long *ticks;
  CK(hipMalloc(&dst, N * (C + 2) * 8)); CK(hipMalloc(&ticks, 1024 * 8)); CK(hipMemset(dst, 0, N * (C + 2) * 8));
  run<0>("A two 512-B stores per pair of columns", dst, N, C, ticks);
  run<1>("B one 1-KB store per pair of columns", dst, N, C, ticks);
  run<0>("A again", dst, N, C, ticks);
  return 0;
}
