// Diagnostic (not product): latency of a vector-compare result reaching the scalar unit (branch on VCC / an SGPR pair)
// for a LONE wave per SIMD, and how much of it overlaps when several compares are issued before the first branch.
//   A  v_cmp_ne_u64 vcc ; s_cbranch_vccnz (never taken)                       one at a time (what "if (ballot(...))" compiles to)
//   B  4 x v_cmp_ne_u64 into s[10:17] ; then 4 x (s_cmp_lg_u64 ; s_cbranch_scc1)   batched by 4
//   C  8 compares, then 8 branches                                             batched by 8
//   D  v_cmp_ne_u64 vcc ; 6 x v_fma_f64 ; s_cbranch_vccnz                      arithmetic between compare and branch
//   E  v_readfirstlane_b32 ; s_cmp_lg_u32 ; s_cbranch_scc1
//   F  s_cmp_lg_u32 ; s_cbranch_scc1 (scalar only, never taken)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP4(s) s s s s
#define REP16(s) REP4(REP4(s))
template <int M>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *ticks, int iters) {
  double x = threadIdx.x * 0.5 + 1.0, y = x, w = 3.0;
  int zero = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (M == 0) asm volatile(REP16("v_cmp_ne_u64 vcc, %0, %1\n s_cbranch_vccnz 1f\n") "1:\n" : : "v"(x), "v"(y) : "vcc");
    if (M == 1) asm volatile(REP4("v_cmp_ne_u64 s[10:11], %0, %1\n v_cmp_ne_u64 s[12:13], %0, %1\n v_cmp_ne_u64 s[14:15], %0, %1\n v_cmp_ne_u64 s[16:17], %0, %1\n"
                                  "s_cmp_lg_u64 s[10:11], 0\n s_cbranch_scc1 1f\n s_cmp_lg_u64 s[12:13], 0\n s_cbranch_scc1 1f\n"
                                  "s_cmp_lg_u64 s[14:15], 0\n s_cbranch_scc1 1f\n s_cmp_lg_u64 s[16:17], 0\n s_cbranch_scc1 1f\n") "1:\n"
                             : : "v"(x), "v"(y) : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "scc");
    if (M == 2) asm volatile("v_cmp_ne_u64 s[10:11], %0, %1\n v_cmp_ne_u64 s[12:13], %0, %1\n v_cmp_ne_u64 s[14:15], %0, %1\n v_cmp_ne_u64 s[16:17], %0, %1\n"
                             "v_cmp_ne_u64 s[18:19], %0, %1\n v_cmp_ne_u64 s[20:21], %0, %1\n v_cmp_ne_u64 s[22:23], %0, %1\n v_cmp_ne_u64 s[24:25], %0, %1\n"
                             "s_cmp_lg_u64 s[10:11], 0\n s_cbranch_scc1 1f\n s_cmp_lg_u64 s[12:13], 0\n s_cbranch_scc1 1f\n"
                             "s_cmp_lg_u64 s[14:15], 0\n s_cbranch_scc1 1f\n s_cmp_lg_u64 s[16:17], 0\n s_cbranch_scc1 1f\n"
                             "s_cmp_lg_u64 s[18:19], 0\n s_cbranch_scc1 1f\n s_cmp_lg_u64 s[20:21], 0\n s_cbranch_scc1 1f\n"
                             "s_cmp_lg_u64 s[22:23], 0\n s_cbranch_scc1 1f\n s_cmp_lg_u64 s[24:25], 0\n s_cbranch_scc1 1f\n"
                             "v_cmp_ne_u64 s[10:11], %0, %1\n v_cmp_ne_u64 s[12:13], %0, %1\n v_cmp_ne_u64 s[14:15], %0, %1\n v_cmp_ne_u64 s[16:17], %0, %1\n"
                             "v_cmp_ne_u64 s[18:19], %0, %1\n v_cmp_ne_u64 s[20:21], %0, %1\n v_cmp_ne_u64 s[22:23], %0, %1\n v_cmp_ne_u64 s[24:25], %0, %1\n"
                             "s_cmp_lg_u64 s[10:11], 0\n s_cbranch_scc1 1f\n s_cmp_lg_u64 s[12:13], 0\n s_cbranch_scc1 1f\n"
                             "s_cmp_lg_u64 s[14:15], 0\n s_cbranch_scc1 1f\n s_cmp_lg_u64 s[16:17], 0\n s_cbranch_scc1 1f\n"
                             "s_cmp_lg_u64 s[18:19], 0\n s_cbranch_scc1 1f\n s_cmp_lg_u64 s[20:21], 0\n s_cbranch_scc1 1f\n"
                             "s_cmp_lg_u64 s[22:23], 0\n s_cbranch_scc1 1f\n s_cmp_lg_u64 s[24:25], 0\n s_cbranch_scc1 1f\n1:\n"
                             : : "v"(x), "v"(y) : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "scc");
    if (M == 3) asm volatile(REP16("v_cmp_ne_u64 vcc, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                                   "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n s_cbranch_vccnz 1f\n") "1:\n"
                             : "+v"(w) : "v"(x), "v"(y) : "vcc");
    if (M == 4) asm volatile(REP16("v_readfirstlane_b32 s10, %0\n s_cmp_lg_u32 s10, 0\n s_cbranch_scc1 1f\n") "1:\n" : : "v"(zero) : "s10", "scc");
    if (M == 5) asm volatile(REP16("s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1f\n") "1:\n" : : "s"(zero) : "scc");
    if (M == 6) asm volatile(REP16("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                                   "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n") : "+v"(w) : "v"(x), "v"(y));
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = x + y + w;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int M> static void run(const char *name, double *out, unsigned long long *ticks) {
  const int iters = 100, waves = 1024;
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL((k<M>), dim3(waves), dim3(64), 0, 0, out, ticks, iters);
  CK(hipDeviceSynchronize());
  unsigned long long h[1024]; CK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
  double sum = 0; for (int i = 0; i < waves; i++) sum += (double)h[i];
  printf("%-72s %6.2f ticks per compare+branch\n", name, sum / waves / (iters * 16.0));
}
int main() {
  double *out; unsigned long long *ticks;
  CK(hipMalloc(&out, 65536 * 8)); CK(hipMalloc(&ticks, 1024 * 8));
  run<0>("A v_cmp_ne_u64 vcc; s_cbranch_vccnz", out, ticks);
  run<1>("B 4 compares into SGPR pairs, then 4 x (s_cmp_lg_u64; s_cbranch_scc1)", out, ticks);
  run<2>("C 8 compares, then 8 branches", out, ticks);
  run<3>("D v_cmp_ne_u64 vcc; 6 x v_fma_f64; s_cbranch_vccnz", out, ticks);
  run<6>("  (6 x v_fma_f64 alone)", out, ticks);
  run<4>("E v_readfirstlane_b32; s_cmp_lg_u32; s_cbranch_scc1", out, ticks);
  run<5>("F s_cmp_lg_u32; s_cbranch_scc1", out, ticks);
  return 0;
}
