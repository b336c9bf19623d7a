// Diagnostic (not product): copy rate of the stepper's access shape (one wave per SIMD, each wave streams C columns of
// its 64 plants, 512 B per column, in place) under different arena layouts:
//   soa      column c of all plants contiguous (pitch N x 8 B): wave w, column c at  c * N * 8 + w * 512
//   blocked  all columns of one wave's 64 plants contiguous: wave w, column c at  w * BS + c * 512, for several BS
// Loads by LDS-DMA (two columns per instruction) one batch ahead, stores of 512 B from registers, as in the stepper.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int B>
__global__ __launch_bounds__(64) void copy_dma(double* __restrict__ a, size_t colstride, size_t blockstride, int C) {
  __shared__ __attribute__((aligned(16))) double lds[72 * 64];
  const int lane = threadIdx.x;
  double* base = a + (size_t)blockIdx.x * blockstride;
  const double* g = base + (size_t)(lane >> 5) * colstride + (size_t)(lane & 31) * 2;
#pragma unroll
  for (int k = 0; k < B; k += 2) __builtin_amdgcn_global_load_lds((gptr_t*)(g + (size_t)k * colstride), (lptr_t*)(lds + k * 64), 16, 0, 0);
  for (int c0 = 0; c0 < C; c0 += B) {
    double v[B];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < B; k++) v[k] = lds[k * 64 + lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (c0 + B < C) {
#pragma unroll
      for (int k = 0; k < B; k += 2)
        __builtin_amdgcn_global_load_lds((gptr_t*)(g + (size_t)(c0 + B + k) * colstride), (lptr_t*)(lds + k * 64), 16, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < B; k++) base[(size_t)(c0 + k) * colstride + lane] = v[k] + 1.0;
  }
  asm volatile("; pad" ::: "v255", "a255");
}
// the same copy, but columns >= nt_from carry the non-temporal bit on their LDS-DMA loads and on their stores
template <int B>
__global__ __launch_bounds__(64) void copy_dma_nt(double* __restrict__ a, size_t colstride, size_t blockstride, int C, int nt_from) {
  __shared__ __attribute__((aligned(16))) double lds[72 * 64];
  const int lane = threadIdx.x;
  double* base = a + (size_t)blockIdx.x * blockstride;
  const double* g = base + (size_t)(lane >> 5) * colstride + (size_t)(lane & 31) * 2;
#pragma unroll
  for (int k = 0; k < B; k += 2) __builtin_amdgcn_global_load_lds((gptr_t*)(g + (size_t)k * colstride), (lptr_t*)(lds + k * 64), 16, 0, 0);
  for (int c0 = 0; c0 < C; c0 += B) {
    double v[B];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < B; k++) v[k] = lds[k * 64 + lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (c0 + B < C) {
      if (c0 + B >= nt_from) {
#pragma unroll
        for (int k = 0; k < B; k += 2)
          __builtin_amdgcn_global_load_lds((gptr_t*)(g + (size_t)(c0 + B + k) * colstride), (lptr_t*)(lds + k * 64), 16, 0, 2);
      } else {
#pragma unroll
        for (int k = 0; k < B; k += 2)
          __builtin_amdgcn_global_load_lds((gptr_t*)(g + (size_t)(c0 + B + k) * colstride), (lptr_t*)(lds + k * 64), 16, 0, 0);
      }
    }
    if (c0 >= nt_from) {
#pragma unroll
      for (int k = 0; k < B; k++) __builtin_nontemporal_store(v[k] + 1.0, &base[(size_t)(c0 + k) * colstride + lane]);
    } else {
#pragma unroll
      for (int k = 0; k < B; k++) base[(size_t)(c0 + k) * colstride + lane] = v[k] + 1.0;
    }
  }
  asm volatile("; pad" ::: "v255", "a255");
}
static void run_nt(const char* name, double* a, size_t colstride, size_t blockstride, int C, size_t waves, int nt_from) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL(copy_dma_nt<32>, dim3(waves), dim3(64), 0, 0, a, colstride, blockstride, C, nt_from);
  CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0;
  for (int i = 0; i < 20; i++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(copy_dma_nt<32>, dim3(waves), dim3(64), 0, 0, a, colstride, blockstride, C, nt_from);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
  }
  const double bytes = 2.0 * waves * 64 * C * 8;
  printf("%-58s mean %7.1f us %5.2f TB/s   best %7.1f us %5.2f TB/s\n", name, sum / 20 * 1e3, bytes / (sum / 20 * 1e-3) / 1e12, best * 1e3,
         bytes / (best * 1e-3) / 1e12);
}
static void run(const char* name, double* a, size_t colstride, size_t blockstride, int C, size_t waves) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(copy_dma<32>, dim3(waves), dim3(64), 0, 0, a, colstride, blockstride, C);
  CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0;
  for (int i = 0; i < 20; i++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(copy_dma<32>, dim3(waves), dim3(64), 0, 0, a, colstride, blockstride, C);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
  }
  const double bytes = 2.0 * waves * 64 * C * 8;
  printf("%-58s mean %7.1f us %5.2f TB/s   best %7.1f us %5.2f TB/s\n", name, sum / 20 * 1e3, bytes / (sum / 20 * 1e-3) / 1e12, best * 1e3,
         bytes / (best * 1e-3) / 1e12);
}
int main() {
  const size_t N = 65536, waves = N / 64; const int C = 512;
  double* a; const size_t maxbytes = (size_t)N * 768 * 8 + (1 << 20);
  CK(hipMalloc(&a, maxbytes > N * C * 8 ? maxbytes : N * C * 8)); CK(hipMemset(a, 0, maxbytes));
  run("soa (pitch 512 KB)", a, N, 64, C, waves);
  run("blocked, block stride 256 KB (= C x 512 B, power of two)", a, 64, (size_t)C * 64, C, waves);
  run("blocked, block stride 256 KB + 256 B", a, 64, (size_t)C * 64 + 32, C, waves);
  run("blocked, block stride 256 KB + 512 B", a, 64, (size_t)C * 64 + 64, C, waves);
  run("blocked, block stride 256 KB + 4 KB", a, 64, (size_t)C * 64 + 512, C, waves);
  run("blocked, block stride 256 KB + 4 KB + 256 B", a, 64, (size_t)C * 64 + 512 + 32, C, waves);
  run("soa again", a, N, 64, C, waves);
  printf("-- working set sweep (soa, in place): the Infinity Cache is 256 MB\n");
  for (int c = 128; c <= 768; c += 64) {
    char name[64]; snprintf(name, sizeof name, "soa, %d columns = %.0f MB", c, c * 512.0 * 1024 / 1e6);
    run(name, a, N, 64, c, waves);
  }
  printf("-- 640 columns (336 MB): columns >= K streamed with the nt bit (loads and stores)\n");
  for (int k : {640, 576, 512, 448, 384, 320, 0}) {
    char name[64]; snprintf(name, sizeof name, "nt from column %d (%.0f MB kept cacheable)", k, k * 512.0 * 1024 / 1e6);
    run_nt(name, a, N, 64, 640, waves, k);
  }
  return 0;
}
