// launchcost -- what a small "gate" kernel launched behind a long one costs on the stream (kernel time by events, K launches
// of [long, gate] back to back), by grid size, register budget, scratch use and kernel-argument size.  Behind the maintenance
// rule kernel's design: its usual work is to read a few flag words and leave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Big { double a[300]; };
__global__ __launch_bounds__(64) void busy(double *x, int iters) {
  double v = x[threadIdx.x];
  for (int i = 0; i < iters; i++) v = v * 1.0000001 + 1e-9;
  x[blockIdx.x * 64 + threadIdx.x] = v;
}
__global__ __launch_bounds__(64) void gate_light(const unsigned *flags, unsigned n, unsigned *out) {
  unsigned m = 0;
  for (unsigned w = blockIdx.x; w < n; w += gridDim.x) m |= flags[w * 8] | flags[w * 8 + 4];
  if (m) out[blockIdx.x] = m;
}
__global__ __launch_bounds__(64) void gate_bigarg(Big b, const unsigned *flags, unsigned n, unsigned *out) {
  unsigned m = 0;
  for (unsigned w = blockIdx.x; w < n; w += gridDim.x) m |= flags[w * 8] | flags[w * 8 + 4];
  if (m) out[blockIdx.x] = m + (unsigned)b.a[m % 300];
}
__global__ __launch_bounds__(64) void gate_scratch(Big b, const unsigned *flags, unsigned n, unsigned *out) {
  unsigned m = 0;
  for (unsigned w = blockIdx.x; w < n; w += gridDim.x) m |= flags[w * 8] | flags[w * 8 + 4];
  if (!m) return;
  volatile double loc[64];            // private array with dynamic indexing: scratch
  for (int i = 0; i < 64; i++) loc[i] = b.a[i] * m;
  out[blockIdx.x] = (unsigned)loc[m % 64];
}
int main() {
  const int K = 300;
  double *x; unsigned *flags, *out;
  hipMalloc(&x, 4096 * 64 * 8); hipMalloc(&flags, 4096 * 8 * 4); hipMalloc(&out, 4096 * 4);
  hipMemset(x, 0, 4096 * 64 * 8); hipMemset(flags, 0, 4096 * 8 * 4);
  Big b = {};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char *name, int which, int grid) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0, 0);
      for (int k = 0; k < K; k++) {
        hipLaunchKernelGGL(busy, dim3(1024), dim3(64), 0, 0, x, 20000);
        if (which == 1) hipLaunchKernelGGL(gate_light, dim3(grid), dim3(64), 0, 0, flags, 1024u, out);
        if (which == 2) hipLaunchKernelGGL(gate_bigarg, dim3(grid), dim3(64), 0, 0, b, flags, 1024u, out);
        if (which == 3) hipLaunchKernelGGL(gate_scratch, dim3(grid), dim3(64), 0, 0, b, flags, 1024u, out);
      }
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%-34s grid %4d  %.3f us per [long + gate]\n", name, grid, ms * 1e3 / K);
    }
  };
  run("long kernel alone", 0, 0);
  for (int g : {1, 16, 64, 256, 1024}) run("+ gate, light", 1, g);
  for (int g : {64, 256}) run("+ gate, 2.4 KB of arguments", 2, g);
  for (int g : {1, 16, 64, 256}) run("+ gate, arguments + scratch", 3, g);
  return 0;
}
