// Micro-benchmark of the binned-deposit path (mcfost_amd/csrc/mc_binned.hip.h) against one global FP64 atomic per
// deposit: 256 persistent workgroups of 1024 lanes, every lane deposits `iters` values into uniformly random cells of
// a 720 000-cell array, with `work` dependent FP64 multiply-adds between two deposits (0: the deposit path alone).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -o tools/binned_deposit_bench tools/binned_deposit_bench.hip
//   tools/binned_deposit_bench [iters] [work] [shift]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../mcfost_amd/csrc/mc_binned.hip.h"

using namespace mcgpu;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ inline unsigned int rnd(unsigned int& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

template <bool BINNED>
__global__ void __launch_bounds__(1024) k_bench(BinLog L, double* E, int n_cells, int iters, int work) {
  extern __shared__ double lds[];
  BinStage S = bin_carve(lds, L.n_buckets);
  if (BINNED) { bin_init(S, L.n_buckets); __syncthreads(); }
  const int lane = threadIdx.x & 63;
  unsigned int s = 1234567u + 7919u * (blockIdx.x * blockDim.x + threadIdx.x);
  double acc = 1.0;
  BinLane P;
  bin_lane_init(P);
  for (int it = 0; it < iters; ++it) {
    const int ic = (int)(((unsigned long long)rnd(s) * (unsigned long long)n_cells) >> 32);
    for (int w = 0; w < work; ++w) acc = acc * 1.0000001 + 1e-9;
    const double v = 1.0 + 1e-30 * acc;
    if (BINNED) bin_deposit(S, L, E, lane, P, true, ic, v);
    else atomic_add_f64(&E[ic], v);
  }
  if (BINNED) { bin_settle(S, L, E, lane, P); __syncthreads(); bin_drain(S, L, E); }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const int work = argc > 2 ? atoi(argv[2]) : 0;
  const int shift = argc > 3 ? atoi(argv[3]) : 14;
  const int n_cells = 720000;
  const int nb = (n_cells + (1 << shift) - 1) >> shift;
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount, threads = 1024;
  const double n_dep = (double)blocks * threads * iters;
  const size_t total_blocks = (size_t)(n_dep / BIN_H * 1.5) + 64 * nb;
  BinLog L{};
  unsigned int *off, *cap;
  CHK(hipMalloc(&L.keys, total_blocks * BIN_H * sizeof(unsigned int)));
  CHK(hipMalloc(&L.vals, total_blocks * BIN_H * sizeof(double)));
  CHK(hipMalloc(&L.count, (size_t)nb * blocks * sizeof(unsigned int)));
  CHK(hipMalloc(&off, nb * sizeof(unsigned int)));
  CHK(hipMalloc(&cap, nb * sizeof(unsigned int)));
  CHK(hipMalloc(&L.stats, 2 * sizeof(unsigned long long)));
  CHK(hipMemset(L.stats, 0, 2 * sizeof(unsigned long long)));
  CHK(hipMemset(L.count, 0, (size_t)nb * blocks * sizeof(unsigned int)));
  std::vector<unsigned int> hoff(nb), hcap(nb);
  for (int b = 0; b < nb; ++b) { hcap[b] = (unsigned int)(total_blocks / nb / blocks); hoff[b] = b * hcap[b] * blocks; }
  CHK(hipMemcpy(off, hoff.data(), nb * 4, hipMemcpyHostToDevice));
  CHK(hipMemcpy(cap, hcap.data(), nb * 4, hipMemcpyHostToDevice));
  L.off = off; L.cap = cap; L.n_buckets = nb; L.shift = shift; L.n_parts = blocks;
  double *E0, *E1;
  CHK(hipMalloc(&E0, n_cells * sizeof(double)));
  CHK(hipMalloc(&E1, n_cells * sizeof(double)));
  const size_t lds = bin_lds_bytes(nb);
  CHK(hipFuncSetAttribute((const void*)k_bench<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CHK(hipFuncSetAttribute((const void*)k_fold_bins, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) << shift)));
  hipEvent_t e0, e1, e2;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); CHK(hipEventCreate(&e2));
  printf("deposits %.3g, buckets %d (shift %d), staging %zu B of LDS, log %.2f GB, work %d\n", n_dep, nb, shift, lds,
         total_blocks * BIN_H * 12.0 / 1e9, work);
  for (int rep = 0; rep < 2; ++rep) {
    CHK(hipMemset(E0, 0, n_cells * sizeof(double)));
    CHK(hipMemset(E1, 0, n_cells * sizeof(double)));
    CHK(hipMemset(L.count, 0, (size_t)nb * blocks * sizeof(unsigned int)));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_bench<false>, dim3(blocks), dim3(threads), 0, 0, L, E0, n_cells, iters, work);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms_atomic;
    CHK(hipEventElapsedTime(&ms_atomic, e0, e1));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_bench<true>, dim3(blocks), dim3(threads), lds, 0, L, E1, n_cells, iters, work);
    CHK(hipEventRecord(e1));
    const int split = 8;
    hipLaunchKernelGGL(k_fold_bins, dim3(nb * split), dim3(1024), sizeof(double) << shift, 0, L, E1, n_cells, split);
    CHK(hipEventRecord(e2));
    CHK(hipEventSynchronize(e2));
    float ms_bin, ms_fold;
    CHK(hipEventElapsedTime(&ms_bin, e0, e1));
    CHK(hipEventElapsedTime(&ms_fold, e1, e2));
    std::vector<double> h0(n_cells), h1(n_cells);
    CHK(hipMemcpy(h0.data(), E0, n_cells * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(h1.data(), E1, n_cells * 8, hipMemcpyDeviceToHost));
    double s0 = 0, s1 = 0, dmax = 0;
    for (int i = 0; i < n_cells; ++i) { s0 += h0[i]; s1 += h1[i]; const double d = fabs(h0[i] - h1[i]); if (d > dmax) dmax = d; }
    unsigned long long st[2];
    CHK(hipMemcpy(st, L.stats, 16, hipMemcpyDeviceToHost));
    printf("atomics %.2f ms = %.3g dep/s | binned %.2f ms + fold %.2f ms = %.3g dep/s (stage alone %.3g) | sum %.6g vs %.6g, max |diff| %.3g, overflow blocks %llu, drained %llu\n",
           ms_atomic, n_dep / ms_atomic * 1e3, ms_bin, ms_fold, n_dep / (ms_bin + ms_fold) * 1e3, n_dep / ms_bin * 1e3, s0, s1, dmax, st[0], st[1]);
  }
  return 0;
}
