// Micro-benchmark behind the xI_scatt deposit design: FP64 global atomics to random 64-byte lines of a
// 128 MB array (MALL/HBM resident), (A) one lane per line, 5 consecutive doubles by 5 instructions,
// (B) the same values transposed so that 8 consecutive lanes cover one line in ONE instruction.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_line_bench.hip -o /tmp/alb && /tmp/alb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void kA(double* a, uint32_t n_lines, int iters, int nval) {
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    s = hash(s + it);
    double* p = a + (size_t)(s % n_lines) * 8;
    for (int t = 0; t < nval; ++t) unsafeAtomicAdd(p + t, 1.0);
  }
}
__global__ void kB(double* a, uint32_t n_lines, int iters, int nval) {
  // the wave handles 64 (lane,iteration) deposits per 8 instructions: lane group g = lane/8 covers line g of round r
  const int lane = threadIdx.x & 63, t = lane & 7, g = lane >> 3;
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    s = hash(s + it);
    const uint32_t my_line = s % n_lines;
    for (int r = 0; r < 8; ++r) {
      const uint32_t line = __shfl(my_line, 8 * r + g);
      if (t < nval) unsafeAtomicAdd(a + (size_t)line * 8 + t, 1.0);
    }
  }
}
int main() {
  const uint32_t n_lines = 2u << 20;  // 128 MB
  double* a;
  hipMalloc(&a, (size_t)n_lines * 64);
  hipMemset(a, 0, (size_t)n_lines * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8, threads = 256, iters = 2000;
  for (int nval : {1, 3, 5, 8}) {
    for (int v = 0; v < 2; ++v) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (v == 0) hipLaunchKernelGGL(kA, dim3(blocks), dim3(threads), 0, 0, a, n_lines, iters, nval);
        else hipLaunchKernelGGL(kB, dim3(blocks), dim3(threads), 0, 0, a, n_lines, iters, nval);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 1) {
          const double deposits = (double)blocks * threads * iters;
          printf("%s nval=%d: %.2f ms, %.3e deposits/s, %.3e value-atomics/s\n", v ? "B(8 lanes/line)" : "A(1 lane/line) ",
                 nval, ms, deposits / (ms * 1e-3), deposits * nval / (ms * 1e-3));
        }
      }
    }
  }
  return 0;
}
