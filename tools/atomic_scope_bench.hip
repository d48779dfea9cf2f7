// Micro-benchmark: where do global atomics execute?  One atomic per lane to a random 8-byte word of an array of
// `mb` megabytes, for FP64 adds (unsafeAtomicAdd: global_atomic_add_f64) and 64-bit integer adds, at agent scope and
// at workgroup scope (no sc1 bit: the XCD's own L2 may serve it).  Reports atomics per second.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_scope_bench.hip -o tools/atomic_scope_bench && tools/atomic_scope_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ void k(unsigned long long* a, uint32_t n_words, int iters) {
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    s = hash(s + it);
    unsigned long long* p = a + (s % n_words);
    if (MODE == 0) unsafeAtomicAdd(reinterpret_cast<double*>(p), 1.0);
    else if (MODE == 1) __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 2) __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 3) __hip_atomic_fetch_add(reinterpret_cast<double*>(p), 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 4) __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(p), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
int main() {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8, threads = 256, iters = 1000;
  const char* names[5] = {"f64 add, agent (unsafeAtomicAdd)", "u64 add, agent", "u64 add, workgroup scope", "f64 add, workgroup scope", "u32 add, workgroup scope"};
  for (int mb : {6, 48, 512}) {
    const uint32_t n_words = (uint32_t)mb << 17;
    unsigned long long* a;
    hipMalloc(&a, (size_t)n_words * 8);
    hipMemset(a, 0, (size_t)n_words * 8);
    for (int mode = 0; mode < 5; ++mode) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        switch (mode) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, a, n_words, iters); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, a, n_words, iters); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, a, n_words, iters); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, a, n_words, iters); break;
          default: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(threads), 0, 0, a, n_words, iters); break;
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      // checksum: every add must have landed (lost updates would show a short sum)
      printf("%4d MB  %-34s %8.2f ms  %.3e atomics/s\n", mb, names[mode], ms, (double)blocks * threads * iters / (ms * 1e-3));
    }
    hipFree(a);
  }
  return 0;
}
