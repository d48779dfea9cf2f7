// Micro-benchmark (tuning aid): throughput of LDS atomic adds with random
// addresses in a 7000-entry array, one 512-thread workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void __launch_bounds__(512) k(int iters, int n, double* out) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = 0.0;
  __syncthreads();
  uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    int idx = (s >> 8) % n;
    double v = 1.0e-3 * (double)(s & 1023);
    // some dependent FP64 work to mimic a crossing (about 40 FMAs)
    double q = v;
#pragma unroll
    for (int j = 0; j < 40; ++j) q = q * 1.0000001 + 1e-9;
    acc += q;
    if (MODE == 0) atomicAdd(&lds[idx], v);                                         // ds_add_f64
    if (MODE == 1) atomicAdd((unsigned long long*)&lds[idx], (unsigned long long)(v * 1048576.0));  // ds_add_u64
    if (MODE == 2) atomicAdd((float*)&lds[idx], (float)v);                          // ds_add_f32
    if (MODE == 3) atomicAdd((unsigned int*)&lds[idx], (unsigned int)(v * 1024.0)); // ds_add_u32
    if (MODE == 4) { }                                                              // none
    if (MODE == 5) { double t = lds[idx]; lds[idx] = t + v; }                       // non-atomic RMW
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = acc + lds[0];
}
template <int MODE>
void run(const char* name, int iters) {
  double* out; hipMalloc(&out, 4096 * sizeof(double));
  int n = 7000; size_t sh = n * sizeof(double);
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<256, 512, sh>>>(iters / 10, n, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<MODE><<<256, 512, sh>>>(iters, n, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double ops = 256.0 * 512 * iters;
  printf("%-14s %8.2f ms  %.3e lane-ops/s  %.1f cycles/wave-op/CU(8 waves)\n", name, ms, ops / (ms * 1e-3),
         ms * 1e-3 * 2.4e9 / (iters * 8.0));
  hipFree(out);
}
int main() {
  int iters = 20000;
  run<4>("none", iters); run<0>("ds_add_f64", iters); run<1>("ds_add_u64", iters); run<2>("ds_add_f32", iters);
  run<3>("ds_add_u32", iters); run<5>("plain_rmw_f64", iters);
  return 0;
}
