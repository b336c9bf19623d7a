// Diagnostic (not product): what does an fp64 compare + select cost a LONE wave per SIMD?  Straight-line inline-asm
// sequences, 1024 waves (one per SIMD), ITER x 32 repetitions each; cycles per repetition from s_memtime.
//   A  v_cmp_gt_f64 vcc                      (compare alone)
//   B  v_cndmask_b32 x2 on a fixed vcc       (select alone)
//   C  v_cmp_gt_f64 vcc ; s_nop 1 ; v_cndmask_b32 x2        (what hipcc emits for  x = c ? a : b  on doubles)
//   D  v_cmp_gt_f64 s[10:11] ; v_cndmask_b32 x2 with s[10:11]
//   E  v_max_f64                              (clip as min/max)
//   F  v_min_f64 + v_max_f64                  (a full clip)
//   G  C with 4 independent v_fma_f64 between compare and selects
//   H  v_cmp_lt_f32 vcc ; s_nop 1 ; v_cndmask_b32           (fp32 compare for reference)
//   I  v_fma_f64                               (baseline)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP4(s) s s s s
#define REP32(s) REP4(REP4(s)) REP4(REP4(s))
template <int M>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *ticks, int iters) {
  double x = threadIdx.x * 0.5 + 1.0, y = 31.0 - threadIdx.x, z = 2.0, w = 3.0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (M == 0) asm volatile(REP32("v_cmp_gt_f64 vcc, %0, %1\n") : : "v"(x), "v"(y) : "vcc");
    if (M == 1) asm volatile(REP32("v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n") : "+v"(((int *)&z)[0]), "+v"(((int *)&z)[1]) : "v"(((int *)&x)[0]), "v"(((int *)&y)[0]) : "vcc");
    if (M == 2) asm volatile(REP32("v_cmp_gt_f64 vcc, %2, %3\n s_nop 1\n v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n")
                             : "+v"(((int *)&z)[0]), "+v"(((int *)&z)[1]) : "v"(x), "v"(y), "v"(((int *)&x)[0]), "v"(((int *)&y)[0]) : "vcc");
    if (M == 3) asm volatile(REP32("v_cmp_gt_f64 s[10:11], %2, %3\n s_nop 1\n v_cndmask_b32 %0, %4, %5, s[10:11]\n v_cndmask_b32 %1, %4, %5, s[10:11]\n")
                             : "+v"(((int *)&z)[0]), "+v"(((int *)&z)[1]) : "v"(x), "v"(y), "v"(((int *)&x)[0]), "v"(((int *)&y)[0]) : "s10", "s11");
    if (M == 4) asm volatile(REP32("v_max_f64 %0, %0, %1\n") : "+v"(z) : "v"(y));
    if (M == 5) asm volatile(REP32("v_max_f64 %0, %0, %1\n v_min_f64 %0, %0, %2\n") : "+v"(z) : "v"(y), "v"(x));
    if (M == 6) asm volatile(REP32("v_cmp_gt_f64 vcc, %3, %4\n v_fma_f64 %2, %2, %3, %4\n v_fma_f64 %2, %2, %3, %4\n v_fma_f64 %2, %2, %3, %4\n v_fma_f64 %2, %2, %3, %4\n"
                                   " v_cndmask_b32 %0, %5, %6, vcc\n v_cndmask_b32 %1, %5, %6, vcc\n")
                             : "+v"(((int *)&z)[0]), "+v"(((int *)&z)[1]), "+v"(w) : "v"(x), "v"(y), "v"(((int *)&x)[0]), "v"(((int *)&y)[0]) : "vcc");
    if (M == 7) asm volatile(REP32("v_cmp_lt_f32 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %0, %1, %2, vcc\n") : "+v"(((int *)&z)[0]) : "v"(((int *)&x)[1]), "v"(((int *)&y)[1]) : "vcc");
    if (M == 8) asm volatile(REP32("v_fma_f64 %0, %0, %1, %2\n") : "+v"(w) : "v"(x), "v"(y));
    if (M == 9) asm volatile(REP32("v_cndmask_b32 %0, %1, %2, vcc\n") : "+v"(((int *)&z)[0]) : "v"(((int *)&x)[0]), "v"(((int *)&y)[0]) : "vcc");
    if (M == 10) asm volatile(REP32("v_cndmask_b32 %0, %1, %2, s[10:11]\n") : "+v"(((int *)&z)[0]) : "v"(((int *)&x)[0]), "v"(((int *)&y)[0]) : "s10", "s11");
    if (M == 11) asm volatile(REP32("v_cndmask_b32 %0, %2, %3, vcc\n v_fma_f64 %1, %1, %4, %5\n") : "+v"(((int *)&z)[0]), "+v"(w) : "v"(((int *)&x)[0]), "v"(((int *)&y)[0]), "v"(x), "v"(y) : "vcc");
    if (M == 12) asm volatile(REP32("v_mov_b32 %0, %1\n") : "+v"(((int *)&z)[0]) : "v"(((int *)&x)[0]));
    if (M == 13) asm volatile(REP32("v_add_u32 %0, %0, %1\n") : "+v"(((int *)&z)[0]) : "v"(((int *)&x)[0]));
    if (M == 14) asm volatile(REP32("v_and_b32 %0, %1, %2\n v_and_b32 %3, %1, %2\n") : "+v"(((int *)&z)[0]), "+v"(((int *)&z)[1]) : "v"(((int *)&x)[0]), "v"(((int *)&y)[0]));
    if (M == 15) asm volatile(REP32("v_cmp_gt_f64 vcc, %2, %3\n v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n")
                             : "+v"(((int *)&z)[0]), "+v"(((int *)&z)[1]) : "v"(x), "v"(y), "v"(((int *)&x)[0]), "v"(((int *)&y)[0]) : "vcc");
    if (M == 16) asm volatile(REP32("v_cmp_gt_f64 vcc, %2, %3\n s_nop 1\n v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_fma_f64 %6, %6, %2, %3\n")
                             : "+v"(((int *)&z)[0]), "+v"(((int *)&z)[1]) : "v"(x), "v"(y), "v"(((int *)&x)[0]), "v"(((int *)&y)[0]), "v"(w) : "vcc");
    if (M == 17) asm volatile(REP32("v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n s_nop 7\n") : "+v"(((int *)&z)[0]), "+v"(((int *)&z)[1]) : "v"(((int *)&x)[0]), "v"(((int *)&y)[0]) : "vcc");
    if (M == 18) asm volatile(REP32("v_mul_f64 %0, %0, %1\n") : "+v"(w) : "v"(x));
    if (M == 19) asm volatile(REP32("v_add_f64 %0, %0, %1\n") : "+v"(w) : "v"(x));
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = x + y + z + w;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int M> static void run(const char *name, double *out, unsigned long long *ticks) {
  const int iters = 100, waves = 1024;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<M>), dim3(waves), dim3(64), 0, 0, out, ticks, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<M>), dim3(waves), dim3(64), 0, 0, out, ticks, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[1024]; CK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
  double sum = 0; for (int i = 0; i < waves; i++) sum += (double)h[i];
  printf("%-58s %7.1f us  %6.2f ticks per repetition\n", name, ms * 1e3, sum / waves / (iters * 32.0));
}
int main() {
  double *out; unsigned long long *ticks;
  CK(hipMalloc(&out, 65536 * 8)); CK(hipMalloc(&ticks, 1024 * 8));
  run<8>("I v_fma_f64", out, ticks);
  run<0>("A v_cmp_gt_f64 vcc", out, ticks);
  run<1>("B 2 x v_cndmask_b32 (vcc)", out, ticks);
  run<2>("C v_cmp_gt_f64 vcc; s_nop 1; 2 x v_cndmask_b32", out, ticks);
  run<3>("D v_cmp_gt_f64 sgpr pair; s_nop 1; 2 x v_cndmask_b32", out, ticks);
  run<4>("E v_max_f64", out, ticks);
  run<5>("F v_max_f64 + v_min_f64", out, ticks);
  run<6>("G v_cmp_gt_f64; 4 x v_fma_f64; 2 x v_cndmask_b32", out, ticks);
  run<7>("H v_cmp_lt_f32 vcc; s_nop 1; v_cndmask_b32", out, ticks);
  run<9>("J v_cndmask_b32 (vcc)", out, ticks);
  run<10>("K v_cndmask_b32 (sgpr pair)", out, ticks);
  run<11>("L v_cndmask_b32 (vcc) + v_fma_f64", out, ticks);
  run<12>("M v_mov_b32", out, ticks);
  run<13>("N v_add_u32", out, ticks);
  run<14>("O 2 x v_and_b32", out, ticks);
  run<15>("P v_cmp_gt_f64 vcc; 2 x v_cndmask_b32 (no s_nop)", out, ticks);
  run<16>("Q C + one v_fma_f64 after the selects", out, ticks);
  run<17>("R 2 x v_cndmask_b32 (vcc); s_nop 7", out, ticks);
  run<18>("S v_mul_f64", out, ticks);
  run<19>("T v_add_f64", out, ticks);
  return 0;
}
