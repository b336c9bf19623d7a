// Diagnostic (not product): which ingredient of the stepper's instruction mix keeps a store from overlapping with
// the arithmetic that follows it?  One wave per SIMD, 300 iterations of [one 512-B store + a block of work]:
//   mode 0  64 FMAs, inline constants only
//   mode 1  64 FMAs whose operands are 32 distinct fp64 literals (each costs two s_mov_b32: scalar-ALU heavy)
//   mode 2  64 FMAs + 8 LDS reads
//   mode 3  64 FMAs with compares/selects (v_cmp + v_cndmask) in between
//   mode 4  64 FMAs + 4 fp64 divisions
// each with and without the store.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int STORE>
__global__ __launch_bounds__(64) void k(double* __restrict__ dst, size_t N, int iters) {
  __shared__ double lds[64 * 16];
  const int lane = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * 64;
  double a = lane, a2 = lane + 1, a3 = lane + 2, a4 = lane + 3;
  for (int q = 0; q < 16; q++) lds[q * 64 + lane] = q;
  for (int it = 0; it < iters; it++) {
    if (STORE) dst[(size_t)(it & 255) * N + base + lane] = a;
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (MODE == 0) { a = a * 0.5 + 1.0; a2 = a2 * 0.5 + 2.0; a3 = a3 * 0.5 + 4.0; a4 = a4 * 0.5 + 1.0; }
      if (MODE == 1) { a = a * (0.501 + u * 0.0013) + (1.003 + u * 0.11); a2 = a2 * (0.502 + u * 0.0017) + 2.0;
                       a3 = a3 * (0.503 + u * 0.0019) + (0.77 + u * 0.13); a4 = a4 * 0.5 + (1.007 + u * 0.17); }
      if (MODE == 2) { a = a * 0.5 + ((u & 1) ? lds[(u >> 1) * 64 + lane] : 1.0); a2 = a2 * 0.5 + 2.0; a3 = a3 * 0.5 + 4.0; a4 = a4 * 0.5 + 1.0; }
      if (MODE == 3) { a = a * 0.5 + 1.0; a = (a > 3.0) ? 3.0 : a; a2 = a2 * 0.5 + 2.0; a2 = (a2 < 0.1) ? 0.1 : a2;
                       a3 = a3 * 0.5 + 4.0; a4 = a4 * 0.5 + 1.0; }
      if (MODE == 4) { a = a * 0.5 + 1.0; a2 = a2 * 0.5 + 2.0; a3 = a3 * 0.5 + 4.0; a4 = (u & 3) ? a4 * 0.5 + 1.0 : a4 / (a3 + 1.5); }
    }
  }
  dst[(size_t)600 * N + base + lane] = a + a2 + a3 + a4;
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
#define RUN(M) do { float t0 = timeit([&] { hipLaunchKernelGGL((k<M, 0>), grid, block, 0, 0, dst, N, iters); }); \
                    float t1 = timeit([&] { hipLaunchKernelGGL((k<M, 1>), grid, block, 0, 0, dst, N, iters); }); \
                    printf("mode %d: work only %7.1f us, with one 512-B store per iteration %7.1f us (+%.1f)\n", M, t0, t1, t1 - t0); } while (0)
int main() {
  const size_t N = 65536; double* dst;
  CK(hipMalloc(&dst, N * 640 * 8)); CK(hipMemset(dst, 0, N * 640 * 8));
  dim3 grid(N / 64), block(64); const int iters = 600;
  printf("600 iterations per wave, 1024 waves (600 x 512 B x 1024 = 315 MB stored)\n");
  RUN(0); RUN(1); RUN(2); RUN(3); RUN(4);
  return 0;
}
