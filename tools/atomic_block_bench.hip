// Micro-benchmark: what is the unit of a memory-side atomic operation -- a 64-byte line, or something larger?
// Default-real global atomics to random places of a 256 MB array; the lanes of one instruction cover aligned blocks of
// W bytes (W = 64: 16 lanes per block, 4 blocks per instruction; 128: 32 lanes, 2 blocks; 256: 64 lanes, 1 block), and
// for comparison the same number of lanes spread over 64-byte lines that are NOT neighbours.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_block_bench.hip -o tools/atomic_block_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// lanes_per_block lanes cover one aligned block of lanes_per_block * 4 bytes; `scatter`: every 16 lanes their own random line
// (the block's lines are not neighbours); `active`: lanes of a 16-lane group that really add (10 of 16 = a line of ten observers)
__global__ void k(float* a, uint32_t n_lines, int iters, int lanes_per_block, int scatter, int active) {
  const int lane = threadIdx.x & 63;
  const int grp = lane / lanes_per_block, in = lane - grp * lanes_per_block;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t s = wave * 977u + 13u;
  const uint32_t lines_per_block = lanes_per_block / 16;
  for (int it = 0; it < iters; ++it) {
    s = hash(s + it);
    uint32_t line;
    if (scatter) line = hash(s ^ (0x9e3779b9u * (uint32_t)(lane >> 4))) % n_lines;
    else line = (hash(s ^ (0x9e3779b9u * (uint32_t)grp)) % (n_lines / lines_per_block)) * lines_per_block + (in >> 4);
    if ((lane & 15) < active) unsafeAtomicAdd(a + (size_t)line * 16 + (lane & 15), 1.0f);
  }
}

int main() {
  const uint32_t n_lines = 4u << 20;  // 256 MB
  float* a;
  hipMalloc(&a, (size_t)n_lines * 64);
  hipMemset(a, 0, (size_t)n_lines * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8, threads = 256, iters = 4000;
  for (int active : {16, 10})
    for (int lpb : {16, 32, 64})
      for (int scatter = 0; scatter < 2; ++scatter) {
        if (lpb == 16 && scatter) continue;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, a, n_lines, iters, lpb, scatter, active);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          hipEventElapsedTime(&ms, e0, e1);
        }
        const double instr = (double)blocks * threads / 64 * iters;
        printf("%2d of 16 lanes active, %s of %3d bytes: %8.2f ms  %.3e lines/s  %.3e blocks/s\n", active,
               scatter ? "scattered lines, groups" : "aligned blocks         ", lpb * 4, ms, instr * 4 / (ms * 1e-3),
               instr * (64 / lpb) / (ms * 1e-3));
      }
  return 0;
}
