// Diagnostic (not product): does a lone wave per SIMD pay for instruction fetch when its code is one long straight line
// (the stepper: ~226 KB of code executed once per launch, no hot loop)?  The same N dependent-free FMAs (four chains)
// as  A  a loop over a 64-instruction body (stays in the instruction cache)  and  B  fully unrolled straight-line code
// of 128 KB, with 1024 waves (full chip, 8 waves share an instruction cache) and with 64 waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define F4 "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
#define R4(s) s s s s
#define R16(s) R4(R4(s))
#define R256(s) R16(R16(s))
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *ticks, int iters) {
  double a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3; const double m = 0.999, n = 1e-3;
  unsigned long long t0 = __builtin_readcyclecounter();
  if (MODE == 0) {
    for (int it = 0; it < iters * 256; it++) asm volatile(R16(F4) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(n));
  } else {
    for (int it = 0; it < iters; it++) asm volatile(R256(R16(F4)) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(n));  // 16384 FMAs = 128 KB
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int MODE> static void run(const char *name, int waves, double *out, unsigned long long *ticks) {
  const int iters = 2;
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, out, ticks, iters);
  CK(hipDeviceSynchronize());
  unsigned long long h[1024]; CK(hipMemcpy(h, ticks, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
  double sum = 0; for (int i = 0; i < waves; i++) sum += (double)h[i];
  printf("%-44s %4d waves: %5.2f ticks per FMA\n", name, waves, sum / waves / (iters * 16384.0));
}
int main() {
  double *out; unsigned long long *ticks;
  CK(hipMalloc(&out, 65536 * 8)); CK(hipMalloc(&ticks, 1024 * 8));
  run<0>("A loop over 64 instructions", 1024, out, ticks); run<1>("B 128 KB of straight-line code", 1024, out, ticks);
  run<0>("A loop over 64 instructions", 64, out, ticks); run<1>("B 128 KB of straight-line code", 64, out, ticks);
  return 0;
}
