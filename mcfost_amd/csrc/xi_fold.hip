// Translation unit of the deposit log's fold (mc_xilog.hip.h): hipCUB's radix sort + k_xi_segfold.
#include "mc_xilog.hip.h"

#include <hipcub/hipcub.hpp>

namespace mcgpu {

// One wave per chunk of sorted records and per window of 64 (observer, slot) values: lane -> value `slot0 + lane` of the
// nv * nRT Stokes values (column of the row = the value's index) followed, with contributions, by the nRT copies of I.
__global__ void __launch_bounds__(256) k_xi_segfold(const unsigned int* __restrict__ keys, const unsigned long long* __restrict__ vals,
                                                    unsigned long long n, const float* __restrict__ rows, int nRT, int nv, int contrib,
                                                    unsigned int n_bins, float* xI, Xi32Lay xi) {
  const int lane = threadIdx.x & 63;
  const unsigned long long wave = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const unsigned long long r_lo = wave * XI_SEG_CHUNK;
  if (r_lo >= n) return;
  const unsigned long long r_hi = (r_lo + XI_SEG_CHUNK < n) ? r_lo + XI_SEG_CHUNK : n;
  const int n_stokes = nv * nRT, n_vals = n_stokes + (contrib ? nRT : 0);
  const int val = (int)blockIdx.y * 64 + lane;
  const bool stokes = val < n_stokes, copy = !stokes && val < n_vals;
  const int q = stokes ? val / nv : (copy ? val - n_stokes : 0);
  const int slot = stokes ? val - q * nv : 0;
  // where this lane's sums go (mc_xi32.hip.h): a Stokes value's place (none for I where it is the sum of the origins),
  // a copy's two places
  const int o_stokes = stokes ? xi32_offset(xi, q, slot, nv) : -1;
  const int o_star = copy ? xi32_offset(xi, q, nv + 1, nv) : -1, o_thermal = copy ? xi32_offset(xi, q, nv + 3, nv) : -1;
  const int col = stokes ? val : q * nv;            // which default real of the row this lane multiplies
  const size_t row_floats = (size_t)n_stokes;
  float acc = 0.0f, acc_star = 0.0f;   // (copy lanes: acc = thermal origin, acc_star = stellar origin)
  unsigned int cur = 0xFFFFFFFFu;
  auto flush = [&](unsigned int bin) {
    float* rec = xI + (size_t)bin * xi.binf;
    if (o_stokes >= 0 && acc != 0.0f) atomicAdd(rec + o_stokes, acc);
    if (copy) { if (acc != 0.0f) atomicAdd(rec + o_thermal, acc); if (acc_star != 0.0f) atomicAdd(rec + o_star, acc_star); }
    acc = 0.0f; acc_star = 0.0f;
  };
  for (unsigned long long r0 = r_lo; r0 < r_hi; r0 += XI_SEG_UNROLL) {
    unsigned int key[XI_SEG_UNROLL];
    float l[XI_SEG_UNROLL], w[XI_SEG_UNROLL];
#pragma unroll
    for (int t = 0; t < XI_SEG_UNROLL; ++t) {
      const unsigned long long r = (r0 + t < r_hi) ? r0 + t : r_hi - 1;   // (wave-uniform addresses)
      const unsigned int k = keys[r];
      key[t] = (r0 + t < r_hi && (k & 0x7FFFFFFFu) < n_bins) ? k : 0xFFFFFFFFu;   // (unused entries sort behind the sub-bins)
      const unsigned long long v = vals[r];
      l[t] = __uint_as_float((unsigned int)(v >> 32));
      w[t] = (key[t] != 0xFFFFFFFFu && (stokes || copy)) ? rows[(size_t)(unsigned int)v * row_floats + col] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < XI_SEG_UNROLL; ++t) {
      if (key[t] == 0xFFFFFFFFu) break;
      const unsigned int bin = key[t] & 0x7FFFFFFFu;
      if (bin != cur) { if (cur != 0xFFFFFFFFu) flush(cur); cur = bin; }
      const float d = l[t] * w[t];
      if (copy && (key[t] >> 31)) acc_star += d; else acc += d;
    }
  }
  if (cur != 0xFFFFFFFFu) flush(cur);
}

size_t xi_sort_temp_bytes(size_t n, int end_bit) {
  size_t bytes = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned int*)nullptr, (unsigned int*)nullptr, (const unsigned long long*)nullptr,
                                     (unsigned long long*)nullptr, (int)n, 0, end_bit);
  return bytes;
}

int xi_sort_fold(hipStream_t stream, const unsigned int* keys, const unsigned long long* vals, unsigned int* keys2,
                 unsigned long long* vals2, size_t n, int end_bit, void* temp, size_t temp_bytes, const float* rows, int nRT,
                 int nv, int contrib, unsigned int n_bins, float* xI, Xi32Lay xi) {
  if (n == 0) return (int)hipSuccess;
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys2, vals, vals2, (int)n, 0, end_bit, stream);
  if (e != hipSuccess) return (int)e;
  const unsigned long long n_waves = (n + XI_SEG_CHUNK - 1) / XI_SEG_CHUNK;
  const int n_vals = nv * nRT + (contrib ? nRT : 0);
  dim3 grid((unsigned int)((n_waves + 3) / 4), (unsigned int)((n_vals + 63) / 64));
  hipLaunchKernelGGL(k_xi_segfold, grid, dim3(256), 0, stream, keys2, vals2, (unsigned long long)n, rows, nRT, nv, contrib, n_bins,
                     xI, xi);
  return (int)hipGetLastError();
}

}  // namespace mcgpu
