// Diagnostic (not product): what copy rate does the SoA arena's access shape reach at ONE wave per SIMD?
// Every wave owns 64 plants and copies C columns (512 B per column per wave) in batches of B.
//   mode 0: B x global_load_dwordx2 -> regs, then B x global_store_dwordx2
//   mode 1: LDS-DMA (global_load_lds_dwordx4, two columns per instruction) -> ds_read -> global_store_dwordx2
//   mode 2: like 0 but software-pipelined: loads of batch b+1 are issued before the stores of batch b
// build: hipcc -O3 --offload-arch=gfx950 -o membench membench.hip ; run: ./membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int B, int REGS_PER_WAVE_PAD>
__global__ __launch_bounds__(64) void copy_regs(const double* __restrict__ src, double* __restrict__ dst, size_t N, int C) {
  const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
  double v[B];
  for (int c0 = 0; c0 < C; c0 += B) {
#pragma unroll
    for (int k = 0; k < B; k++) v[k] = src[(size_t)(c0 + k) * N + p];
#pragma unroll
    for (int k = 0; k < B; k++) dst[(size_t)(c0 + k) * N + p] = v[k] + 1.0;
  }
  // pad VGPR usage so that only one wave fits per SIMD, as in the step kernel
  if (REGS_PER_WAVE_PAD) { asm volatile("; pad" ::: "v255", "a255"); }
}

template <int B>
__global__ __launch_bounds__(64) void copy_pipe(const double* __restrict__ src, double* __restrict__ dst, size_t N, int C) {
  const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
  double v[B], w[B];
#pragma unroll
  for (int k = 0; k < B; k++) v[k] = src[(size_t)k * N + p];
  for (int c0 = 0; c0 < C; c0 += B) {
    if (c0 + B < C) {
#pragma unroll
      for (int k = 0; k < B; k++) w[k] = src[(size_t)(c0 + B + k) * N + p];
    }
#pragma unroll
    for (int k = 0; k < B; k++) dst[(size_t)(c0 + k) * N + p] = v[k] + 1.0;
#pragma unroll
    for (int k = 0; k < B; k++) v[k] = w[k];
  }
  asm volatile("; pad" ::: "v255", "a255");
}

template <int B>
__global__ __launch_bounds__(64) void copy_dma(const double* __restrict__ src, double* __restrict__ dst, size_t N, int C) {
  __shared__ __attribute__((aligned(16))) double lds[72 * 64];
  const int lane = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * 64;
  const size_t p = base + lane;
  const double* g = src + (size_t)(lane >> 5) * N + base + (size_t)(lane & 31) * 2;
  // stage batch 0
#pragma unroll
  for (int k = 0; k < B; k += 2) __builtin_amdgcn_global_load_lds((gptr_t*)(g + (size_t)k * N), (lptr_t*)(lds + k * 64), 16, 0, 0);
  for (int c0 = 0; c0 < C; c0 += B) {
    double v[B];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < B; k++) v[k] = lds[k * 64 + lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (c0 + B < C) {
#pragma unroll
      for (int k = 0; k < B; k += 2)
        __builtin_amdgcn_global_load_lds((gptr_t*)(g + (size_t)(c0 + B + k) * N), (lptr_t*)(lds + k * 64), 16, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < B; k++) dst[(size_t)(c0 + k) * N + p] = v[k] + 1.0;
  }
  asm volatile("; pad" ::: "v255", "a255");
}

// write-only, 8 B per lane (512 B per instruction), values from registers
__global__ __launch_bounds__(64) void wr_x2(const double* __restrict__ src, double* __restrict__ dst, size_t N, int C) {
  const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
  double v = (double)p;
  for (int c0 = 0; c0 < C; c0 += 8) {
#pragma unroll
    for (int k = 0; k < 8; k++) dst[(size_t)(c0 + k) * N + p] = v + k;
  }
}
// write-only, 16 B per lane: lanes 0-31 write column c (2 plants each), lanes 32-63 column c+1 (1 KB per instruction)
__global__ __launch_bounds__(64) void wr_x4(const double* __restrict__ src, double* __restrict__ dst, size_t N, int C) {
  const int lane = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * 64;
  double2 v = make_double2((double)lane, 1.0);
  double* g = dst + (size_t)(lane >> 5) * N + base + (size_t)(lane & 31) * 2;
  for (int c0 = 0; c0 < C; c0 += 16) {
#pragma unroll
    for (int k = 0; k < 16; k += 2) *(double2*)(g + (size_t)(c0 + k) * N) = v;
  }
}
// read-only x2 (sum to keep the loads alive)
__global__ __launch_bounds__(64) void rd_x2(const double* __restrict__ src, double* __restrict__ dst, size_t N, int C) {
  const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
  double acc = 0;
  for (int c0 = 0; c0 < C; c0 += 16) {
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = src[(size_t)(c0 + k) * N + p];
#pragma unroll
    for (int k = 0; k < 16; k++) acc += v[k];
  }
  if (acc == 123.456) dst[p] = acc;
}
// read-only through LDS-DMA x4
__global__ __launch_bounds__(64) void rd_dma(const double* __restrict__ src, double* __restrict__ dst, size_t N, int C) {
  __shared__ __attribute__((aligned(16))) double lds[72 * 64];
  const int lane = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * 64;
  const double* g = src + (size_t)(lane >> 5) * N + base + (size_t)(lane & 31) * 2;
  double acc = 0;
  for (int c0 = 0; c0 < C; c0 += 32) {
#pragma unroll
    for (int k = 0; k < 32; k += 2) __builtin_amdgcn_global_load_lds((gptr_t*)(g + (size_t)(c0 + k) * N), (lptr_t*)(lds + k * 64), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 32; k++) acc += lds[k * 64 + lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (acc == 123.456) dst[base + lane] = acc;
}
// copy: LDS-DMA x4 loads, stores as x4 through an LDS transposition (ds_write_b64 own value, ds_read_b128 pair layout)
template <int B>
__global__ __launch_bounds__(64) void copy_dma_x4st(const double* __restrict__ src, double* __restrict__ dst, size_t N, int C) {
  __shared__ __attribute__((aligned(16))) double lds[72 * 64];
  const int lane = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * 64;
  const double* g = src + (size_t)(lane >> 5) * N + base + (size_t)(lane & 31) * 2;
  double* gd = dst + (size_t)(lane >> 5) * N + base + (size_t)(lane & 31) * 2;
#pragma unroll
  for (int k = 0; k < B; k += 2) __builtin_amdgcn_global_load_lds((gptr_t*)(g + (size_t)k * N), (lptr_t*)(lds + k * 64), 16, 0, 0);
  for (int c0 = 0; c0 < C; c0 += B) {
    double v[B];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < B; k++) v[k] = lds[k * 64 + lane] + 1.0;   // "compute" on own plant
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // results back to LDS (own column position), then pair layout out
#pragma unroll
    for (int k = 0; k < B; k++) lds[k * 64 + lane] = v[k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double2 w[B / 2];
#pragma unroll
    for (int k = 0; k < B; k += 2) w[k / 2] = *(const double2*)(lds + (k + (lane >> 5)) * 64 + (lane & 31) * 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (c0 + B < C) {
#pragma unroll
      for (int k = 0; k < B; k += 2)
        __builtin_amdgcn_global_load_lds((gptr_t*)(g + (size_t)(c0 + B + k) * N), (lptr_t*)(lds + k * 64), 16, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < B; k += 2) *(double2*)(gd + (size_t)(c0 + k) * N) = w[k / 2];
  }
}

template <typename F> static void timeit(const char* name, F launch, double bytes) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; i++) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  const int K = 20;
  for (int i = 0; i < K; i++) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= K;
  printf("%-44s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t N = 65536; const int C = 512;  // 512 columns ~ the stepper's 524
  double *src, *dst;
  CK(hipMalloc(&src, N * C * 8)); CK(hipMalloc(&dst, N * C * 8));
  CK(hipMemset(src, 0, N * C * 8)); CK(hipMemset(dst, 0, N * C * 8));
  const double bytes = 2.0 * N * C * 8;
  dim3 grid(N / 64), block(64);
#define RUN(NAME, KERN) timeit(NAME, [&] { hipLaunchKernelGGL(KERN, grid, block, 0, 0, src, dst, N, C); }, bytes)
  RUN("regs B=8  (1 wave/SIMD)", (copy_regs<8, 1>));
  RUN("regs B=16 (1 wave/SIMD)", (copy_regs<16, 1>));
  RUN("regs B=32 (1 wave/SIMD)", (copy_regs<32, 1>));
  RUN("regs B=64 (1 wave/SIMD)", (copy_regs<64, 1>));
  RUN("regs B=32 (occupancy free)", (copy_regs<32, 0>));
  RUN("pipelined regs B=16", (copy_pipe<16>));
  RUN("pipelined regs B=32", (copy_pipe<32>));
  RUN("lds-dma B=16", (copy_dma<16>));
  RUN("lds-dma B=32", (copy_dma<32>));
  RUN("lds-dma B=64", (copy_dma<64>));
  RUN("lds-dma x4 loads + x4 stores via LDS, B=16", (copy_dma_x4st<16>));
  RUN("lds-dma x4 loads + x4 stores via LDS, B=32", (copy_dma_x4st<32>));
  printf("-- one direction only (bytes = one pass)\n");
  timeit("write-only x2 (512 B / instr)", [&] { hipLaunchKernelGGL(wr_x2, grid, block, 0, 0, src, dst, N, C); }, bytes / 2);
  timeit("write-only x4 (1 KB / instr)", [&] { hipLaunchKernelGGL(wr_x4, grid, block, 0, 0, src, dst, N, C); }, bytes / 2);
  timeit("read-only x2", [&] { hipLaunchKernelGGL(rd_x2, grid, block, 0, 0, src, dst, N, C); }, bytes / 2);
  timeit("read-only lds-dma x4", [&] { hipLaunchKernelGGL(rd_dma, grid, block, 0, 0, src, dst, N, C); }, bytes / 2);
  return 0;
}
