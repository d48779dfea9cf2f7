// How fast does a CPU thread read pinned host memory (hipHostMalloc) compared with the heap?  (round 6: the host side of a
// launch's tail reads its table copies from the pinned arena the device copied them into.)
// hipcc -O2 -o tools/pinned_read_bench tools/pinned_read_bench.hip && tools/pinned_read_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double chase(const unsigned int* a, size_t n, size_t steps) {
  unsigned int i = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (size_t s = 0; s < steps; ++s) i = a[i];
  const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
  if (i == 0xFFFFFFFFu) printf("!");
  return ns / steps;
}
int main() {
  const size_t n = (1u << 18);   // 1 MB of indices: fits the core's L2
  unsigned int* heap = (unsigned int*)malloc(n * 4);
  for (size_t i = 0; i < n; ++i) heap[i] = (unsigned int)((i * 2654435761ull + 12345) % n);
  unsigned int *pin_default, *pin_coherent, *pin_noncoh;
  hipHostMalloc((void**)&pin_default, n * 4, hipHostMallocDefault);
  hipHostMalloc((void**)&pin_coherent, n * 4, hipHostMallocCoherent);
  hipHostMalloc((void**)&pin_noncoh, n * 4, hipHostMallocNonCoherent);
  memcpy(pin_default, heap, n * 4); memcpy(pin_coherent, heap, n * 4); memcpy(pin_noncoh, heap, n * 4);
  unsigned int* reg = (unsigned int*)aligned_alloc(4096, n * 4);
  memcpy(reg, heap, n * 4);
  hipHostRegister(reg, n * 4, hipHostRegisterDefault);
  for (int rep = 0; rep < 2; ++rep)
    printf("dependent reads, ns each: heap %.1f  hipHostMalloc default %.1f  coherent %.1f  non-coherent %.1f  hipHostRegister %.1f\n",
           chase(heap, n, 1 << 22), chase(pin_default, n, 1 << 22), chase(pin_coherent, n, 1 << 22), chase(pin_noncoh, n, 1 << 22), chase(reg, n, 1 << 22));
  return 0;
}
