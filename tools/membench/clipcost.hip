// Diagnostic (not product): cost of np.clip formulations inside a dependent fp64 chain, lone wave per SIMD.
// Each repetition: x = fma(x, a, b); x = clip(x, lo, hi), C independent chains.
//   0 no clip   1 compare-select twice (NaN propagates)   2 fmin(fmax())   3 fmin(fmax()) + NaN fix-up select
//   4 fmin(fmax()) + (x - x)   5 fma(x, 0, fmin(fmax()))   6 fmin(fmax()) + wave-uniform branch to the fix-up if any lane holds NaN
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int M> __device__ __forceinline__ double clipf(double x, double lo, double hi) {
  if (M == 0) return x;
  if (M == 1) { double t = (x < lo) ? lo : x; return (t > hi) ? hi : t; }
  if (M == 2) return __builtin_fmin(__builtin_fmax(x, lo), hi);
  if (M == 3) { double r = __builtin_fmin(__builtin_fmax(x, lo), hi); return (x != x) ? x : r; }
  if (M == 4) { double r = __builtin_fmin(__builtin_fmax(x, lo), hi); return r + (x - x); }
  if (M == 5) return __builtin_fma(x, 0.0, __builtin_fmin(__builtin_fmax(x, lo), hi));
  if (M == 6) { double r = __builtin_fmin(__builtin_fmax(x, lo), hi); if (__builtin_amdgcn_ballot_w64(x != x) != 0) r = (x != x) ? x : r; return r; }
  if (M == 7) { double r = __builtin_fmin(__builtin_fmax(x, lo), hi);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(x != x) != 0, 0)) { asm volatile("; cold path"); r = (x != x) ? x : r; } return r; }
  return x;
}
template <int M, int C>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *ticks, int iters, double a, double b, double lo, double hi) {
  double x[C];
#pragma unroll
  for (int c = 0; c < C; c++) x[c] = threadIdx.x * 0.001 + c + 1.0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 32 / C; u++)
#pragma unroll
      for (int c = 0; c < C; c++) x[c] = clipf<M>(__builtin_fma(x[c], a, b), lo, hi);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int c = 0; c < C; c++) s += x[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int M, int C> static void run(const char *name, double *out, unsigned long long *ticks) {
  const int iters = 200, waves = 1024;
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL((k<M, C>), dim3(waves), dim3(64), 0, 0, out, ticks, iters, 1.0000001, 1e-7, 0.5, 1e6);
  CK(hipDeviceSynchronize());
  unsigned long long h[1024]; CK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
  double sum = 0; for (int i = 0; i < waves; i++) sum += (double)h[i];
  printf("%-44s chains %d: %6.2f ticks per fma+clip\n", name, C, sum / waves / (iters * 32.0));
}
#define BOTH(M, name) run<M, 1>(name, out, ticks); run<M, 4>(name, out, ticks)
int main() {
  double *out; unsigned long long *ticks;
  CK(hipMalloc(&out, 65536 * 8)); CK(hipMalloc(&ticks, 1024 * 8));
  BOTH(0, "fma only"); BOTH(1, "compare-select x2 (exact)"); BOTH(2, "fmin(fmax())"); BOTH(3, "fmin(fmax()) + NaN fix-up select (exact)");
  BOTH(4, "fmin(fmax()) + (x - x)");
  BOTH(5, "fma(x, 0, fmin(fmax()))"); BOTH(6, "fmin(fmax()) + ballot branch to fix-up (exact)"); BOTH(7, "the same, branch kept by a volatile asm (exact)");
  return 0;
}
