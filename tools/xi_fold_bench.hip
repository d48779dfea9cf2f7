// xi_fold_bench.hip -- round 6: the SED commit pass's deposits as LOG + FOLD instead of atomics, costed standalone.
//
// Today a crossing of the commit pass adds l * (4 Stokes weights + the copy of I) for each of nRT observers into
// xI_scatt[cell][psup][phik][iRT][8] with global atomics: default-real records, two observers per 64-byte line ->
// nRT / 2 line operations per crossing at 2.37e10 line-ops/s for the whole chip (tools/atomic_line_bench.hip).
// The alternative measured here, at BASELINE config 2's statistics (7000 cells x 2 x 45 sub-bins, 10 observers, flights of
// a few crossings through neighbouring cells):
//   * the flying lane appends ONE 12-byte record per crossing (bin, flight id, path length) and, once per flight, the row of
//     its nRT x 4 default-real weights (what angles_scatt_rt1 already computes per flight in the default-real commit pass);
//   * the records are sorted by bin (here: hipcub radix sort as a stand-in for two partition passes of the product's
//     bucket staging -- its cost is reported separately);
//   * a fold workgroup owns the accumulators of a range of bins in LDS (padded against bank conflicts), gathers the
//     flight's row per record, does the nRT x 5 ds_add_f32 and writes its range of xI_scatt once.
// A/B: the same records deposited with global atomics in the product's arrangement (K = 5 lanes per record and line).
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/xi_fold_bench.hip -o tools/xi_fold_bench && tools/xi_fold_bench [flights] [mean crossings]
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int N_CELLS = 7000, N_RAD = 100, N_PSUP = 2, N_PHIK = 45, NRT = 10, XI_LINE = 8;
constexpr int N_BINS = N_CELLS * N_PSUP * N_PHIK;   // 630 000
constexpr int Q_STRIDE = 9;                         // floats per (bin, observer) in LDS: 8 slots + 1 of padding
constexpr int BIN_STRIDE = NRT * Q_STRIDE + 1;      // 91 floats per bin in LDS (odd: bins spread over the banks)
constexpr int SLICE_BINS = 352;                     // bins a fold workgroup owns: 352 x 91 x 4 = 128 KB of LDS
constexpr int FOLD_THREADS = 640;                   // 64 records x 10 observers per pass
#ifndef FOLD_UNROLL
#define FOLD_UNROLL 4
#endif

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// ---- the generator: one flight per thread, its row and its records (unsorted, in time order as the transport kernel
// would append them: the lanes of a wave write their crossing i side by side) -------------------------------------------
__global__ void k_generate(uint32_t n_flights, float mean_len, float4* rows, uint32_t* keys, uint2* vals, unsigned long long* n_rec,
                           uint32_t cap) {
  const uint32_t fid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint32_t s = hash(fid * 2654435761u + 12345u);
  int len = 0, cell = 0, phik = 0;
  if (fid < n_flights) {
    // geometric length (mean mean_len), at least one crossing, at most 96
    const float u = (float)(s >> 8) * (1.0f / 16777216.0f);
    len = 1 + (int)(-logf(1.0f - u * 0.999999f) * (mean_len - 1.0f));
    if (len > 96) len = 96;
    s = hash(s);
    // where flights are: the inner, opaque cells take most of them (a power law over the radial index)
    const float r = (float)(s >> 8) * (1.0f / 16777216.0f);
    const int ri = (int)(N_RAD * r * r);
    s = hash(s);
    const int zj = (int)((s >> 8) % 70u);
    cell = ri + N_RAD * zj;
    s = hash(s);
    phik = (int)((s >> 8) % N_PHIK);
    for (int q = 0; q < NRT; ++q) {
      s = hash(s);
      const float w = (float)(s >> 8) * (1.0f / 16777216.0f);
      rows[(size_t)fid * NRT + q] = make_float4(w, 0.1f * w, -0.05f * w, 0.01f * w);
    }
  }
  const int max_len = __reduce_max_sync(~0ull, len);
  for (int i = 0; i < max_len; ++i) {
    const bool on = i < len;
    const unsigned long long m = __ballot(on);
    if (!m) break;
    unsigned long long base = 0;
    const int leader = __ffsll((long long)m) - 1;
    if (lane == leader) base = atomicAdd(n_rec, (unsigned long long)__popcll(m));
    base = __shfl(base, leader);
    if (on) {
      const unsigned long long at = base + __popcll(m & ((1ull << lane) - 1ull));
      s = hash(s + i);
      // the next cell: a neighbour (radial or vertical), the azimuthal sub-bin drifts slowly, above / below at random
      const int step = (int)(s & 3u);
      if (step == 0 && cell % N_RAD < N_RAD - 1) cell += 1;
      else if (step == 1 && cell % N_RAD > 0) cell -= 1;
      else if (step == 2 && cell + N_RAD < N_CELLS) cell += N_RAD;
      else if (step == 3 && cell >= N_RAD) cell -= N_RAD;
      if ((s >> 4 & 7u) == 0u) phik = (phik + 1) % N_PHIK;
      const int psup = (int)(s >> 8 & 1u);
      const uint32_t bin = ((uint32_t)cell * N_PSUP + psup) * N_PHIK + phik;
      const float l = 0.5f + (float)(s >> 12 & 1023u) * (1.0f / 1024.0f);
      if (at < cap) { keys[at] = bin | ((s >> 30 & 1u) << 31); vals[at] = make_uint2(fid, __float_as_uint(l)); }   // bit 31: flag_star
    }
  }
}

// ---- A: global atomics in the product's arrangement: K = 5 lanes per record, two observers per 64-byte line ------------
__global__ void k_atomics(const uint32_t* keys, const uint2* vals, unsigned long long n_rec, const float4* rows, float* xI) {
  // a wave takes 6 records per pass of a pair of observers: 12 lanes per record (2 observers x (4 Stokes + the copy of I)) -> 72 > 64:
  // 5 records x 12 lanes = 60 lanes per instruction, like deposit_rt1_wave_f32's NR = 64 / (2 K)
  const int lane = threadIdx.x & 63;
  const int rl = lane / 10, jj = lane - rl * 10, hh = jj / 5, j = jj - hh * 5;
  const unsigned long long n_waves = (unsigned long long)gridDim.x * (blockDim.x >> 6);
  const unsigned long long wave = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  for (unsigned long long r0 = wave * 6; r0 < n_rec; r0 += n_waves * 6) {
    const unsigned long long r = r0 + rl;
    const bool ok = rl < 6 && r < n_rec;
    const uint32_t key = ok ? keys[r] : 0u;
    const uint2 v = ok ? vals[r] : make_uint2(0u, 0u);
    const uint32_t bin = key & 0x7FFFFFFFu;
    const float l = __uint_as_float(v.y);
    for (int q0 = 0; q0 < NRT; q0 += 2) {
      const float4 w = rows[(size_t)v.x * NRT + q0 + hh];
      const float val = l * (j == 0 || j == 4 ? w.x : (j == 1 ? w.y : (j == 2 ? w.z : w.w)));
      const int slot = j < 4 ? j : ((key >> 31) ? 5 : 7);
      if (ok) atomicAdd(xI + ((size_t)bin * NRT + q0 + hh) * XI_LINE + slot, val);
    }
  }
}

// ---- B: fold.  offs[b] .. offs[b + 1]: the sorted records of slice b (bins b * SLICE_BINS ...) -----------------------------
struct FoldItem { unsigned long long r_lo, r_hi; int slice, shared; };   // shared: the slice is folded by several workgroups

__global__ void __launch_bounds__(FOLD_THREADS) k_fold(const uint32_t* keys, const uint2* vals, const FoldItem* items,
                                                      const float4* rows, float* xI) {
  extern __shared__ float acc[];   // [SLICE_BINS][BIN_STRIDE]
  const FoldItem it = items[blockIdx.x];
  const int b = it.slice;
  for (int i = threadIdx.x; i < SLICE_BINS * BIN_STRIDE; i += blockDim.x) acc[i] = 0.0f;
  __syncthreads();
  const unsigned long long r_lo = it.r_lo, r_hi = it.r_hi;
  const int q = threadIdx.x % NRT, rr = threadIdx.x / NRT;   // 64 records per pass, 10 threads each
  const uint32_t bin0 = (uint32_t)b * SLICE_BINS;
  // (FOLD_UNROLL records per thread in flight: the row gather is a dependent load behind the record's -- one at a time it
  // runs at the latency of HBM, not at its bandwidth)
  constexpr int PER_PASS = FOLD_THREADS / NRT;
  for (unsigned long long r0 = r_lo; r0 < r_hi; r0 += (unsigned long long)PER_PASS * FOLD_UNROLL) {
    uint32_t key[FOLD_UNROLL];
    uint2 v[FOLD_UNROLL];
    float4 w[FOLD_UNROLL];
#pragma unroll
    for (int t = 0; t < FOLD_UNROLL; ++t) {
      const unsigned long long r = r0 + (unsigned long long)t * PER_PASS + rr;
      const bool ok = r < r_hi;
      key[t] = ok ? keys[r] : 0xFFFFFFFFu;
      v[t] = ok ? vals[r] : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int t = 0; t < FOLD_UNROLL; ++t) w[t] = rows[(size_t)v[t].x * NRT + q];
#pragma unroll
    for (int t = 0; t < FOLD_UNROLL; ++t) {
      if (key[t] == 0xFFFFFFFFu) continue;
      const float l = __uint_as_float(v[t].y);
      float* a = acc + (size_t)((key[t] & 0x7FFFFFFFu) - bin0) * BIN_STRIDE + q * Q_STRIDE;
      atomicAdd(a + 0, l * w[t].x); atomicAdd(a + 1, l * w[t].y); atomicAdd(a + 2, l * w[t].z); atomicAdd(a + 3, l * w[t].w);
      atomicAdd(a + ((key[t] >> 31) ? 5 : 7), l * w[t].x);
    }
  }
  __syncthreads();
  // xI_scatt += the slice (each bin belongs to one workgroup: plain read-modify-write, 8 floats per (bin, observer))
  const int n_bins = (bin0 + SLICE_BINS <= (uint32_t)N_BINS) ? SLICE_BINS : (int)(N_BINS - bin0);
  for (int i = threadIdx.x; i < n_bins * NRT * XI_LINE; i += blockDim.x) {
    const int bl = i / (NRT * XI_LINE), rem = i - bl * NRT * XI_LINE, qq = rem / XI_LINE, s = rem - qq * XI_LINE;
    const float v = acc[(size_t)bl * BIN_STRIDE + qq * Q_STRIDE + s];
    if (v != 0.0f) {
      float* dst = xI + ((size_t)(bin0 + bl) * NRT + qq) * XI_LINE + s;
      if (it.shared) atomicAdd(dst, v); else *dst += v;   // (8 neighbouring lanes = one 32-byte record: one line operation)
    }
  }
}


// ---- C: segmented sums.  The records are sorted by bin, so the records of one bin are consecutive: a wave walks a chunk of
// records with the 4 Stokes weights of the 10 observers in 40 lanes (and the copy of I per observer in 10 more), one
// coalesced 160-byte row load and ONE multiply-add per lane and record, sums in registers while the bin stays the same and
// adds to xI_scatt when it changes -- no atomics but at the segments' ends (a bin may continue in the next wave's chunk).
#ifndef SEG_CHUNK
#define SEG_CHUNK 512
#endif
#ifndef SEG_UNROLL
#define SEG_UNROLL 8
#endif
__global__ void __launch_bounds__(256) k_segfold(const uint32_t* keys, const uint2* vals, unsigned long long n_rec, const float* rowsf,
                                                 float* xI) {
  const int lane = threadIdx.x & 63;
  const unsigned long long wave = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const unsigned long long r_lo = wave * SEG_CHUNK;
  if (r_lo >= n_rec) return;
  const unsigned long long r_hi = (r_lo + SEG_CHUNK < n_rec) ? r_lo + SEG_CHUNK : n_rec;
  const bool stokes = lane < 4 * NRT, copy = lane >= 4 * NRT && lane < 5 * NRT;
  const int q = stokes ? lane >> 2 : (copy ? lane - 4 * NRT : 0);
  const int col = stokes ? lane : (copy ? 4 * (lane - 4 * NRT) : 0);   // which float of the row this lane multiplies
  float acc = 0.0f, acc_star = 0.0f;   // (copy lanes: acc = thermal origin (slot 7), acc_star = stellar origin (slot 5))
  uint32_t cur = 0xFFFFFFFFu;
  auto flush = [&](uint32_t bin) {
    float* rec = xI + ((size_t)bin * NRT + q) * XI_LINE;
    if (stokes && acc != 0.0f) atomicAdd(rec + (lane & 3), acc);
    if (copy) { if (acc != 0.0f) atomicAdd(rec + 7, acc); if (acc_star != 0.0f) atomicAdd(rec + 5, acc_star); }
    acc = 0.0f; acc_star = 0.0f;
  };
  for (unsigned long long r0 = r_lo; r0 < r_hi; r0 += SEG_UNROLL) {
    uint32_t key[SEG_UNROLL];
    float l[SEG_UNROLL], w[SEG_UNROLL];
#pragma unroll
    for (int t = 0; t < SEG_UNROLL; ++t) {
      const unsigned long long r = (r0 + t < r_hi) ? r0 + t : r_hi - 1;   // (wave-uniform: scalar loads)
      key[t] = (r0 + t < r_hi) ? keys[r] : 0xFFFFFFFFu;
      const uint2 v = vals[r];
      l[t] = __uint_as_float(v.y);
      w[t] = (stokes || copy) ? rowsf[(size_t)v.x * (4 * NRT) + col] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < SEG_UNROLL; ++t) {
      if (key[t] == 0xFFFFFFFFu) break;
      const uint32_t bin = key[t] & 0x7FFFFFFFu;
      if (bin != cur) { if (cur != 0xFFFFFFFFu) flush(cur); cur = bin; }
      const float d = l[t] * w[t];
      if (copy && (key[t] >> 31)) acc_star += d; else acc += d;
    }
  }
  if (cur != 0xFFFFFFFFu) flush(cur);
}

__global__ void k_offsets(const uint32_t* keys_sorted, unsigned long long n_rec, unsigned long long* offs, int n_slices) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > n_slices) return;
  const uint32_t first = (uint32_t)b * SLICE_BINS;   // first record whose bin >= first
  unsigned long long lo = 0, hi = n_rec;
  while (lo < hi) { const unsigned long long mid = (lo + hi) >> 1; if ((keys_sorted[mid] & 0x7FFFFFFFu) < first) lo = mid + 1; else hi = mid; }
  offs[b] = lo;
}

int main(int argc, char** argv) {
  const uint32_t n_flights = argc > 1 ? (uint32_t)atof(argv[1]) : 16000000u;
  const float mean_len = argc > 2 ? (float)atof(argv[2]) : 8.0f;
  const uint32_t cap = (uint32_t)((double)n_flights * mean_len * 1.3) + 1024u;
  float4* rows; uint32_t *keys, *keys2; uint2 *vals, *vals2; unsigned long long *n_rec_d, *offs; float *xI_a, *xI_b;
  CHK(hipMalloc(&rows, (size_t)n_flights * NRT * sizeof(float4)));
  CHK(hipMalloc(&keys, (size_t)cap * 4)); CHK(hipMalloc(&keys2, (size_t)cap * 4));
  CHK(hipMalloc(&vals, (size_t)cap * 8)); CHK(hipMalloc(&vals2, (size_t)cap * 8));
  CHK(hipMalloc(&n_rec_d, 8)); CHK(hipMemset(n_rec_d, 0, 8));
  const size_t n_xi = (size_t)N_BINS * NRT * XI_LINE;
  CHK(hipMalloc(&xI_a, n_xi * 4)); CHK(hipMalloc(&xI_b, n_xi * 4));
  CHK(hipMemset(xI_a, 0, n_xi * 4)); CHK(hipMemset(xI_b, 0, n_xi * 4));
  const int n_slices = (N_BINS + SLICE_BINS - 1) / SLICE_BINS;
  CHK(hipMalloc(&offs, (size_t)(n_slices + 1) * 8));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_generate, dim3((n_flights + 255) / 256), dim3(256), 0, 0, n_flights, mean_len, rows, keys, vals, n_rec_d, cap);
  hipEventRecord(e1); CHK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
  unsigned long long n_rec = 0;
  CHK(hipMemcpy(&n_rec, n_rec_d, 8, hipMemcpyDeviceToHost));
  if (n_rec > cap) n_rec = cap;
  printf("%u flights, %llu crossings (%.2f per flight), log %.2f GB + rows %.2f GB; generated in %.1f ms\n", n_flights, n_rec,
         (double)n_rec / n_flights, n_rec * 12e-9, (double)n_flights * NRT * 16e-9, ms);
  // A: atomics
  for (int rep = 0; rep < 2; ++rep) {
    CHK(hipMemset(xI_a, 0, n_xi * 4));
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_atomics, dim3(256 * 8), dim3(256), 0, 0, keys, vals, n_rec, rows, xI_a);
    hipEventRecord(e1); CHK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
  }
  const double t_atomic = ms;
  printf("A  global atomics (5 lanes per record and line): %8.2f ms  %.3e crossings/s  %.3e line-ops/s\n", ms, n_rec / (ms * 1e-3),
         n_rec * (NRT / 2.0) / (ms * 1e-3));
  // B: sort (stand-in for two partition passes) + fold
  size_t tmp_bytes = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys2, reinterpret_cast<unsigned long long*>(vals),
                                     reinterpret_cast<unsigned long long*>(vals2), (int)n_rec, 0, 20);
  void* tmp; CHK(hipMalloc(&tmp, tmp_bytes));
  double t_sort = 0, t_fold = 0;
  CHK(hipFuncSetAttribute((const void*)k_fold, hipFuncAttributeMaxDynamicSharedMemorySize, SLICE_BINS * BIN_STRIDE * 4));
  for (int rep = 0; rep < 2; ++rep) {
    CHK(hipMemset(xI_b, 0, n_xi * 4));
    hipEventRecord(e0);
    // (the low 20 bits hold the bin: 630 000 < 2^20; the flag bit rides along unsorted)
    hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys2, reinterpret_cast<unsigned long long*>(vals),
                                       reinterpret_cast<unsigned long long*>(vals2), (int)n_rec, 0, 20);
    hipLaunchKernelGGL(k_offsets, dim3((n_slices + 256) / 256), dim3(256), 0, 0, keys2, n_rec, offs, n_slices);
    hipEventRecord(e1); CHK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
    t_sort = ms;
  }
  CHK(hipGetLastError());
  // work items: a slice with many records is folded by several workgroups (each at most `chunk` records)
  std::vector<unsigned long long> h_offs(n_slices + 1);
  CHK(hipMemcpy(h_offs.data(), offs, (size_t)(n_slices + 1) * 8, hipMemcpyDeviceToHost));
  for (unsigned long long chunk : {1ull << 62, 1ull << 17, 1ull << 15, 1ull << 13}) {
    std::vector<FoldItem> items;
    for (int b = 0; b < n_slices; ++b) {
      const unsigned long long lo = h_offs[b], hi = h_offs[b + 1];
      if (hi == lo) continue;
      const unsigned long long parts = (hi - lo + chunk - 1) / chunk;
      for (unsigned long long p = 0; p < parts; ++p)
        items.push_back(FoldItem{lo + p * chunk, (lo + (p + 1) * chunk < hi) ? lo + (p + 1) * chunk : hi, b, parts > 1 ? 1 : 0});
    }
    FoldItem* d_items;
    CHK(hipMalloc(&d_items, items.size() * sizeof(FoldItem)));
    CHK(hipMemcpy(d_items, items.data(), items.size() * sizeof(FoldItem), hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
      CHK(hipMemset(xI_b, 0, n_xi * 4));
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_fold, dim3((unsigned)items.size()), dim3(FOLD_THREADS), SLICE_BINS * BIN_STRIDE * 4, 0, keys2, vals2, d_items, rows, xI_b);
      hipEventRecord(e1); CHK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
    }
    t_fold = ms;
    unsigned long long biggest = 0;
    for (int b = 0; b < n_slices; ++b) if (h_offs[b + 1] - h_offs[b] > biggest) biggest = h_offs[b + 1] - h_offs[b];
    printf("B  fold, at most %8llu records per workgroup: %5zu workgroups (largest slice %llu records) %8.2f ms  %.3e crossings/s  row gather %.0f GB/s\n",
           chunk > (1ull << 40) ? 0ull : chunk, items.size(), biggest, ms, n_rec / (ms * 1e-3), n_rec * 160.0 / (ms * 1e6));
    CHK(hipFree(d_items));
  }
  {  // C: segmented sums over the sorted records
    float* xI_c;
    CHK(hipMalloc(&xI_c, n_xi * 4));
    const unsigned long long n_waves = (n_rec + SEG_CHUNK - 1) / SEG_CHUNK;
    for (int rep = 0; rep < 2; ++rep) {
      CHK(hipMemset(xI_c, 0, n_xi * 4));
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_segfold, dim3((unsigned)((n_waves + 3) / 4)), dim3(256), 0, 0, keys2, vals2, n_rec, reinterpret_cast<const float*>(rows), xI_c);
      hipEventRecord(e1); CHK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
    }
    CHK(hipGetLastError());
    std::vector<float> ha2(n_xi), hc(n_xi);
    CHK(hipMemcpy(ha2.data(), xI_a, n_xi * 4, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(hc.data(), xI_c, n_xi * 4, hipMemcpyDeviceToHost));
    double worst_c = 0, sc = 0;
    for (size_t i = 0; i < n_xi; ++i) {
      sc += hc[i];
      const double d = fabs((double)ha2[i] - hc[i]) / (fabs((double)ha2[i]) + 1e-3);
      if (d > worst_c) worst_c = d;
    }
    printf("C  segmented sums over the sorted records (chunks of %d, %d records in flight): %8.2f ms  %.3e crossings/s  row gather %.0f GB/s; "
           "with the sort %.2f ms = %.2fx the atomics; sum %.6e, largest relative difference %.2e\n", SEG_CHUNK, SEG_UNROLL, ms,
           n_rec / (ms * 1e-3), n_rec * 160.0 / (ms * 1e6), ms + t_sort, t_atomic / (ms + t_sort), sc, worst_c);
    CHK(hipFree(xI_c));
  }
  printf("B  sort by bin (radix, 20 bits: stand-in for 2 partition passes): %8.2f ms  %.3e crossings/s\n", t_sort, n_rec / (t_sort * 1e-3));
  printf("B  (%d slices of %d bins, %d KB of LDS, %d threads per workgroup)\n", n_slices, SLICE_BINS, SLICE_BINS * BIN_STRIDE * 4 / 1024, FOLD_THREADS);
  printf("B  sort + fold: %.2f ms = %.2fx the atomics; fold alone %.2fx\n", t_sort + t_fold, t_atomic / (t_sort + t_fold), t_atomic / t_fold);
  // same sums?
  std::vector<float> ha(n_xi), hb(n_xi);
  CHK(hipMemcpy(ha.data(), xI_a, n_xi * 4, hipMemcpyDeviceToHost));
  CHK(hipMemcpy(hb.data(), xI_b, n_xi * 4, hipMemcpyDeviceToHost));
  double sa = 0, sb = 0, worst = 0;
  for (size_t i = 0; i < n_xi; ++i) {
    sa += ha[i]; sb += hb[i];
    const double d = fabs((double)ha[i] - hb[i]) / (fabs((double)ha[i]) + 1e-3);
    if (d > worst) worst = d;
  }
  printf("sums: atomics %.6e, fold %.6e, largest relative difference of an entry %.2e (default-real summation order)\n", sa, sb, worst);
  return 0;
}
