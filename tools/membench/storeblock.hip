// Diagnostic (not product): does a global store occupy the issuing wave, or does it overlap with that wave's
// arithmetic?  One wave per SIMD (1024 blocks of 64).  Loop: [one store] + K dependent FMAs, for several K.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

// MODE 0: arithmetic only; 1: + one 8-B/lane store per iteration; 2: + one 16-B/lane store; 3: + one LDS-DMA x4 load
template <int MODE>
__global__ __launch_bounds__(64) void k(const double* __restrict__ src, double* __restrict__ dst, size_t N, int iters, int K) {
  __shared__ __attribute__((aligned(16))) double lds[64 * 64];
  const int lane = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * 64;
  double a = lane, a2 = lane + 1, a3 = lane + 2, a4 = lane + 3, b = 1.0000001, c = 0.5;
  for (int it = 0; it < iters; it++) {
    // K/16 blocks of 16 straight-line FMAs on four independent chains (no inner branch: a 4-instruction loop's
    // time depends on where the code lands relative to fetch lines, which differs between the variants)
    for (int q = 0; q < K; q += 16) {
#pragma unroll
      for (int u = 0; u < 4; u++) { a = a * b + c; a2 = a2 * b + c; a3 = a3 * b + c; a4 = a4 * b + c; }
    }
    const size_t col = (size_t)(it & 255);
    if (MODE == 1) dst[col * N + base + lane] = a;
    if (MODE == 2) { d2 v; v.x = a; v.y = a; *(d2*)(dst + (col * 2 + (lane >> 5)) * N + base + (lane & 31) * 2) = v; }
    if (MODE == 3) __builtin_amdgcn_global_load_lds((gptr_t*)(src + (col * 2 + (lane >> 5)) * N + base + (lane & 31) * 2), (lptr_t*)(lds + (it & 31) * 128), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  dst[(size_t)600 * N + base + lane] = a + a2 + a3 + a4 + lds[lane];
}

template <typename F> static float timeit(F launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; i++) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  const int R = 10;
  for (int i = 0; i < R; i++) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / R * 1e3f;
}

int main() {
  const size_t N = 65536;
  double *src, *dst;
  CK(hipMalloc(&src, N * 640 * 8)); CK(hipMalloc(&dst, N * 640 * 8));
  CK(hipMemset(src, 0, N * 640 * 8)); CK(hipMemset(dst, 0, N * 640 * 8));
  dim3 grid(N / 64), block(64);
  const int iters = 300;   // ~ stores per wave per step in the stepper
  printf("300 iterations per wave, 1024 waves; per iteration: K fp64 FMAs (4 chains, straight-line blocks of 16) + one memory instruction\n");
  for (int K : {0, 64, 128, 256, 512, 1024}) {
    float t0 = timeit([&] { hipLaunchKernelGGL(k<0>, grid, block, 0, 0, src, dst, N, iters, K); });
    float t1 = timeit([&] { hipLaunchKernelGGL(k<1>, grid, block, 0, 0, src, dst, N, iters, K); });
    float t2 = timeit([&] { hipLaunchKernelGGL(k<2>, grid, block, 0, 0, src, dst, N, iters, K); });
    float t3 = timeit([&] { hipLaunchKernelGGL(k<3>, grid, block, 0, 0, src, dst, N, iters, K); });
    printf("K=%4d: arithmetic %7.1f us | +store 512B %7.1f | +store 1KB %7.1f | +LDS-DMA 1KB %7.1f\n", K, t0, t1, t2, t3);
  }
  return 0;
}
