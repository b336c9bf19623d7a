// Diagnostic (not product): issue cost of fp64 vector instructions for a LONE wave per SIMD, as a function of how
// many independent dependency chains the stream offers.  1024 waves (one per SIMD), each runs ITER x 64 instructions
// of one kind spread over C chains (C = 1: every instruction depends on the one before it).
//   fma  x = x * a + b      mul  x = x * a      add  x = x + a      min  x = fmin(x, a) (compare/select class)
//   rcp  x = 1 / x (v_rcp_f64, quarter rate)    sqrt x = sqrt(x) (full sequence)    mix  fma with a distinct literal each
// Output: cycles per instruction at the shader clock (s_memtime delta / instructions).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum { FMA, MUL, ADD, MIN, RCP, CVT };
template <int OP> __device__ __forceinline__ double op(double x, double a, double b) {
  if (OP == FMA) return __builtin_fma(x, a, b);
  if (OP == MUL) return x * a;
  if (OP == ADD) return x + a;
  if (OP == MIN) return x < a ? x : b;
  if (OP == RCP) return __builtin_amdgcn_rcp(x);
  if (OP == CVT) return (double)(float)x;
  return x;
}
template <int OP, int C>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *ticks, int iters, double a, double b) {
  double x[C];
#pragma unroll
  for (int c = 0; c < C; c++) x[c] = threadIdx.x * 0.001 + c + 1.0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 64 / C; u++)
#pragma unroll
      for (int c = 0; c < C; c++) x[c] = op<OP>(x[c], a, b);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int c = 0; c < C; c++) s += x[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int OP, int C> static void run(const char *name, double *out, unsigned long long *ticks, double a, double b) {
  const int iters = 200, waves = 1024;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<OP, C>), dim3(waves), dim3(64), 0, 0, out, ticks, iters, a, b);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<OP, C>), dim3(waves), dim3(64), 0, 0, out, ticks, iters, a, b);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[1024]; CK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
  double sum = 0; for (int i = 0; i < waves; i++) sum += (double)h[i];
  const double n = (double)iters * 64;
  printf("%-5s chains %d: %7.1f us, %6.2f s_memtime ticks per instruction (%.2f ns per instruction)\n", name, C, ms * 1e3,
         sum / waves / n, ms * 1e6 / n);
}
#define ALLC(OP, name) run<OP, 1>(name, out, ticks, a, b); run<OP, 2>(name, out, ticks, a, b); run<OP, 4>(name, out, ticks, a, b); run<OP, 8>(name, out, ticks, a, b)
int main() {
  double *out; unsigned long long *ticks;
  CK(hipMalloc(&out, 65536 * 8)); CK(hipMalloc(&ticks, 1024 * 8));
  const double a = 0.999999, b = 1e-6;
  ALLC(FMA, "fma"); ALLC(MUL, "mul"); ALLC(ADD, "add"); ALLC(MIN, "min"); ALLC(RCP, "rcp"); ALLC(CVT, "cvt");
  return 0;
}
