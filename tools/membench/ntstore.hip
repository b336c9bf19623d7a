// Diagnostic (not product): does a write-only output stream evict the resident arena from the Infinity Cache, and does the
// non-temporal bit on its stores prevent that?  In-place copy of C columns (as layout.hip) plus X extra write-only
// columns per wave into a separate buffer, stored (a) normally, (b) with "nt", (c) with "sc0 sc1 nt".
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__device__ __forceinline__ void out_store(double *p, double v) {
  if (MODE == 0) *p = v;
  if (MODE == 1) asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
  if (MODE == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(v) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(64) void k(double* __restrict__ a, double* __restrict__ out, size_t N, int C, int X) {
  __shared__ __attribute__((aligned(16))) double lds[72 * 64];
  const int B = 32;
  const int lane = threadIdx.x;
  double* base = a + (size_t)blockIdx.x * 64;
  const double* g = base + (size_t)(lane >> 5) * N + (size_t)(lane & 31) * 2;
#pragma unroll
  for (int kk = 0; kk < B; kk += 2) __builtin_amdgcn_global_load_lds((gptr_t*)(g + (size_t)kk * N), (lptr_t*)(lds + kk * 64), 16, 0, 0);
  double acc = 0;
  for (int c0 = 0; c0 < C; c0 += B) {
    double v[B];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int kk = 0; kk < B; kk++) v[kk] = lds[kk * 64 + lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (c0 + B < C) {
#pragma unroll
      for (int kk = 0; kk < B; kk += 2)
        __builtin_amdgcn_global_load_lds((gptr_t*)(g + (size_t)(c0 + B + kk) * N), (lptr_t*)(lds + kk * 64), 16, 0, 0);
    }
#pragma unroll
    for (int kk = 0; kk < B; kk++) { base[(size_t)(c0 + kk) * N + lane] = v[kk] + 1.0; acc += v[kk]; }
  }
  for (int x = 0; x < X; x++) out_store<MODE>(out + (size_t)x * N + (size_t)blockIdx.x * 64 + lane, acc + x);
  asm volatile("; pad" ::: "v255", "a255");
}
template <int MODE> static void run(const char* name, double* a, double* out, size_t N, int C, int X) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<MODE>, dim3(N / 64), dim3(64), 0, 0, a, out, N, C, X);
  CK(hipDeviceSynchronize());
  float sum = 0;
  for (int i = 0; i < 20; i++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(N / 64), dim3(64), 0, 0, a, out, N, C, X);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms;
  }
  printf("%-64s %7.1f us\n", name, sum / 20 * 1e3);
}
int main() {
  const size_t N = 65536;
  double *a, *out;
  CK(hipMalloc(&a, N * 600 * 8)); CK(hipMalloc(&out, N * 64 * 8)); CK(hipMemset(a, 0, N * 600 * 8));
  for (int C : {448, 480, 512}) {
    char nm[96];
    snprintf(nm, sizeof nm, "%d columns in place (%.0f MB), no extra output", C, C * 512.0 * 1024 / 1e6); run<0>(nm, a, out, N, C, 0);
    snprintf(nm, sizeof nm, "%d columns + 32 output columns (17 MB), plain stores", C); run<0>(nm, a, out, N, C, 32);
    snprintf(nm, sizeof nm, "%d columns + 32 output columns, nt stores", C); run<1>(nm, a, out, N, C, 32);
    snprintf(nm, sizeof nm, "%d columns + 32 output columns, sc0 sc1 nt stores", C); run<2>(nm, a, out, N, C, 32);
  }
  return 0;
}
