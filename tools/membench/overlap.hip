// Diagnostic (not product): does sprinkling memory instructions through the arithmetic overlap them at one
// wave per SIMD, compared with issuing them in one burst per phase?  Mimics the stepper: 12 phases, each
// "computes" ITER dependent-ish fp64 FMAs, stores 44 columns and stages 44 columns (LDS-DMA) for a later phase.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define COLS 44
#define PHASES 12

__device__ __forceinline__ void work(double& a, double& b, double& c, double& d, int n) {
#pragma unroll 4
  for (int i = 0; i < n; i++) { a = a * 1.0000001 + b; b = b * 0.9999999 + c; c = c * 1.0000002 + d; d = d * 0.9999998 + a; }
}

// MODE 0: compute only. 1: burst (stores + DMA at the phase boundary). 2: sprinkled (11 ticks of 4 stores + 2 DMA).
// 3: memory only (burst, no compute)
template <int MODE>
__global__ __launch_bounds__(64) void k(const double* __restrict__ src, double* __restrict__ dst, size_t N, int iters) {
  __shared__ __attribute__((aligned(16))) double lds[COLS * 64];
  const int lane = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * 64;
  const size_t p = base + lane;
  const double* g = src + (size_t)(lane >> 5) * N + base + (size_t)(lane & 31) * 2;
  double a = lane, b = 1, c = 2, d = 3;
  for (int ph = 0; ph < PHASES; ph++) {
    const size_t col0 = (size_t)ph * COLS;
    if (MODE == 1 || MODE == 3) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < COLS; q++) dst[(col0 + q) * N + p] = a + q;
#pragma unroll
      for (int q = 0; q < COLS; q += 2) __builtin_amdgcn_global_load_lds((gptr_t*)(g + (col0 + q) * N), (lptr_t*)(lds + q * 64), 16, 0, 0);
      if (MODE == 1) work(a, b, c, d, iters);
    } else if (MODE == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int per = iters / 11;
#pragma unroll
      for (int t = 0; t < 11; t++) {
        work(a, b, c, d, per);
#pragma unroll
        for (int q = 4 * t; q < 4 * t + 4; q++) dst[(col0 + q) * N + p] = a + q;
#pragma unroll
        for (int q = 4 * t; q < 4 * t + 4; q += 2) __builtin_amdgcn_global_load_lds((gptr_t*)(g + (col0 + q) * N), (lptr_t*)(lds + q * 64), 16, 0, 0);
      }
    } else {
      work(a, b, c, d, iters);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  dst[(size_t)(PHASES * COLS) * N + p] = a + b + c + d + lds[lane];
}

template <typename F> static float timeit(F launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; i++) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  const int K = 20;
  for (int i = 0; i < K; i++) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / K * 1e3f;
}

int main() {
  const size_t N = 65536; const int C = PHASES * COLS + 1;
  double *src, *dst;
  CK(hipMalloc(&src, N * C * 8)); CK(hipMalloc(&dst, N * C * 8));
  CK(hipMemset(src, 0, N * C * 8)); CK(hipMemset(dst, 0, N * C * 8));
  dim3 grid(N / 64), block(64);
  printf("bytes moved per launch: %.0f MB\n", 2.0 * N * PHASES * COLS * 8 / 1e6);
  for (int iters : {275, 550, 1100, 2200}) {
    float t0 = timeit([&] { hipLaunchKernelGGL(k<0>, grid, block, 0, 0, src, dst, N, iters); });
    float t1 = timeit([&] { hipLaunchKernelGGL(k<1>, grid, block, 0, 0, src, dst, N, iters); });
    float t2 = timeit([&] { hipLaunchKernelGGL(k<2>, grid, block, 0, 0, src, dst, N, iters); });
    float t3 = timeit([&] { hipLaunchKernelGGL(k<3>, grid, block, 0, 0, src, dst, N, iters); });
    printf("iters/phase %5d: compute-only %7.1f us | memory-only %7.1f us | burst %7.1f us | sprinkled %7.1f us\n", iters, t0, t3, t1, t2);
  }
  return 0;
}
