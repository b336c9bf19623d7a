// Diagnostic (not product): does a 512-B store cost a lone wave more when its address register was written by the
// instruction just before it?  1024 waves (one per SIMD); per iteration 16 stores to 16 columns, with 8 FMAs between
// stores (a wave that has other work).  Address = SGPR base + 32-bit VGPR offset, offset = s_mul + v_add as in the stepper.
//   A  offset computed right before each store      B  the 16 offsets computed first, then the 16 stores
//   C  offsets computed once outside the loop (16 VGPRs held)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((address_space(1))) char gchar_t;
__device__ __forceinline__ uint32_t voff(uint32_t col, uint32_t pitch, uint32_t lane_off) {
  uint32_t v, t;
  asm volatile("s_mul_i32 %1, %2, %3\n\tv_add_u32 %0, %1, %4" : "=v"(v), "=&s"(t) : "s"(col), "s"(pitch), "v"(lane_off));
  return v;
}
template <int MODE>
__global__ __launch_bounds__(64) void k(double *dst, size_t N, int iters, unsigned long long *ticks) {
  gchar_t *base = (gchar_t *)(dst + (size_t)blockIdx.x * 64);
  const uint32_t pitch = (uint32_t)(N * 8), lane8 = threadIdx.x * 8;
  double a = threadIdx.x, b = 1.0;
  uint32_t pre[16];
  if (MODE == 2) { for (int c = 0; c < 16; c++) pre[c] = voff(c, pitch, lane8); }
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    uint32_t off[16];
    if (MODE == 1) {
#pragma unroll
      for (int c = 0; c < 16; c++) off[c] = voff(c, pitch, lane8);
    }
#pragma unroll
    for (int c = 0; c < 16; c++) {
      uint32_t o = MODE == 0 ? voff(c, pitch, lane8) : (MODE == 1 ? off[c] : pre[c]);
      *(__attribute__((address_space(1))) double *)(base + o) = a;
#pragma unroll
      for (int u = 0; u < 8; u++) { a = __builtin_fma(a, 0.999, b); }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  dst[(size_t)20 * N + (size_t)blockIdx.x * 64 + threadIdx.x] = a;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int MODE> static void run(const char *name, double *dst, size_t N, unsigned long long *ticks) {
  const int iters = 40;
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<MODE>, dim3(N / 64), dim3(64), 0, 0, dst, N, iters, ticks);
  CK(hipDeviceSynchronize());
  unsigned long long h[1024]; CK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
  double sum = 0; for (int i = 0; i < 1024; i++) sum += (double)h[i];
  printf("%-58s %7.1f ticks per store (+ 8 FMAs)\n", name, sum / 1024 / (iters * 16.0));
}
int main() {
  const size_t N = 65536; double *dst; unsigned long long *ticks;
  CK(hipMalloc(&dst, N * 24 * 8)); CK(hipMalloc(&ticks, 1024 * 8));
  run<0>("A offset computed right before each store", dst, N, ticks);
  run<1>("B 16 offsets first, then 16 stores", dst, N, ticks);
  run<2>("C offsets held in registers", dst, N, ticks);
  return 0;
}
