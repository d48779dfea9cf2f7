// Binned deposits: the absorbed-energy deposits of grids that do not fit in LDS (3D cylindrical: 720 000 cells,
// Voronoi) WITHOUT one memory-side atomic per cell crossing.
//
// save_radiation_field (radiation_field.f90:53) is `xKJ_abs(icell,id) += kappa_abs * l * Stokes(1)` for every cell a
// packet crosses.  As a global_atomic_add_f64 per crossing it runs at the memory side's atomic rate -- 2.4e10 per
// second chip-wide whatever the kernel does, every 16-byte read-modify-write moving a 64-byte
// request each way.  Here a deposit is a 12-byte RECORD (cell, value):
//   1. the packet kernel appends it to one of n_buckets address-range BUCKETS staged in LDS (bucket = cell >> shift;
//      two half-buffers of BIN_H = 64 records per bucket);
//   2. the lane whose record completes a half-buffer makes its WAVE write that block to the log in HBM: 256 contiguous
//      bytes of cells + 512 contiguous bytes of values.  Every workgroup owns its own part of every bucket's region
//      and fills it front to back, so a flush needs NO global atomic (a returning atomic on a shared cursor cost the
//      flushing wave ~3000 cycles of memory-side latency per block: measured, tools/binned_deposit_bench.hip);
//   3. k_fold_bins gives every bucket to workgroups that sum its blocks into the bucket's slice of the array in LDS
//      (ds_add_f64) and add the slice to the array once.
// HBM traffic: 24 bytes per deposit, streamed, against 128 bytes of uncached requests; no global atomics but the fold's.
// The sums are the same terms in another order (the tests' rtol 1e-9 on E_abs covers it; counters are untouched).
//
// The protocol in LDS (per bucket b, half h): resv[b] hands out positions; a position's block k = pos / 64 lives in
// half k & 1 and may be written once that half has been flushed k >> 1 times (epoch[b][h]); every writer bumps
// done[b][h] after its stores, and the one that brings it to 64 owns the flush.  A lane whose half is still being
// flushed WAITS at a wave-uniform point while its wave keeps doing the flushes it owns (bin_settle), so the lowest unflushed block
// of a bucket is always writable and its flush never waits for anybody: no deadlock.  A block that finds its
// region of the log full is added to the array with plain atomics (graceful overflow: sizes are a matter of speed).
#pragma once
#include "mc_device.hip.h"

namespace mcgpu {

constexpr int BIN_H = 64;                      // records per block = lanes of the flushing wave
constexpr int BIN_MAX_BUCKETS = 96;
#ifdef MCGPU_LANE_EMULATION
constexpr int BIN_WAVE = 1;
#else
constexpr int BIN_WAVE = 64;
#endif

// (struct BinLog, the HBM side of the log: mc_device.hip.h, next to RunArgs)

// In-flight temperature (Temp_LTE reads the cell's absorbed energy, thermal_emission.f90:670): E_abs holds the deposits
// of the n_folded packets of the chunks before this one; the packets this chunk has started so far are accounted
// for by scaling -- the reference's `xKJ_abs(icell,id) * nb_proc` idea (a partial sum stands for the whole) applied
// along the run instead of across threads.  The first chunk of a run (n_folded = 0) sees E = 0 like the reference's
// first packets; the host keeps it short and lets the chunks grow geometrically.
__device__ inline double bin_energy_scale(const RunArgs& A) {
  if (!(A.n_folded > 0.0)) return 1.0;
  unsigned long long p = __hip_atomic_load(A.next_packet, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (the work counter hands out the carried records first: they are packets of earlier chunks, already in n_folded)
  const unsigned long long n_carry = A.carry_in_n ? (unsigned long long)*A.carry_in_n : 0ull;
  p = p > n_carry ? p - n_carry : 0ull;
  const double started = (double)(p < A.n_packets ? p : A.n_packets);
  return (A.n_folded + started) / A.n_folded;
}

__host__ __device__ inline size_t bin_lds_bytes(int n_buckets) {
  return (size_t)n_buckets * (2 * BIN_H * (sizeof(double) + sizeof(unsigned int)) + 6 * sizeof(unsigned int));
}

// LDS side (one per workgroup)
struct BinStage {
  double* vals;         // [n_buckets][2 * BIN_H]
  unsigned int* keys;   // [n_buckets][2 * BIN_H]
  unsigned int* resv;   // [n_buckets]
  unsigned int* epoch;  // [n_buckets][2]
  unsigned int* done;   // [n_buckets][2]
  unsigned int* wcur;   // [n_buckets] blocks this workgroup has written to its part of the bucket's region
};

__device__ inline BinStage bin_carve(void* base, int n_buckets) {
  BinStage S;
  S.vals = reinterpret_cast<double*>(base);
  S.keys = reinterpret_cast<unsigned int*>(S.vals + (size_t)n_buckets * 2 * BIN_H);
  S.epoch = S.keys + (size_t)n_buckets * 2 * BIN_H;  // (8-byte aligned: both halves' epochs are read as one word)
  S.done = S.epoch + 2 * n_buckets;
  S.resv = S.done + 2 * n_buckets;
  S.wcur = S.resv + n_buckets;
  return S;
}

// (every thread of the workgroup, before a __syncthreads())
__device__ inline void bin_init(const BinStage& S, int n_buckets) {
  for (int i = threadIdx.x; i < 6 * n_buckets; i += blockDim.x) S.epoch[i] = 0u;
}

__device__ inline unsigned int bin_ldu(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// the whole wave writes block (b, h) to the log and opens the half for the next epoch
__device__ inline void bin_flush_block(const BinStage& S, const BinLog& L, double* E, int b, int h, int lane) {
  unsigned int blk = 0u;
  if (lane == 0) blk = atomicAdd(&S.wcur[b], 1u);
  blk = __shfl(blk, 0);
  const unsigned int cap = L.cap[b];
  const bool fits = blk < cap;
  for (int l = lane; l < BIN_H; l += BIN_WAVE) {
    const int slot = (b * 2 + h) * BIN_H + l;
    const unsigned int key = S.keys[slot];
    const double v = S.vals[slot];
    if (fits) {
      const size_t at = ((size_t)L.off[b] + (size_t)cap * blockIdx.x + blk) * BIN_H + l;
      L.keys[at] = key;
      L.vals[at] = v;
    } else {
      atomic_add_f64(&E[key], v);
    }
  }
  __threadfence_block();  // the slots have been read before the half opens again
  if (lane == 0) {
    if (!fits) atomicAdd(&L.stats[0], 1ull);
    S.done[b * 2 + h] = 0u;
    __threadfence_block();
    atomicAdd(&S.epoch[b * 2 + h], 1u);
  }
}

// What a lane carries from one bin_deposit to the next: the half-buffer it last wrote to and what the count of that
// half was before its own bump.  The bump's result is not waited for where it is issued -- it has long returned when
// the next call looks at it (LDS operations of a wave return in order, and the next call's first wait is for an
// operation issued later) --, so a deposit costs the wave ONE wait on LDS instead of three.
struct BinLane {
  unsigned int seen;   // done[b][h] before this lane's bump (BIN_H - 1: the lane completed the block)
  int b, h;
  bool open;           // a bump whose result has not been looked at
};
__device__ inline void bin_lane_init(BinLane& P) { P.seen = 0u; P.b = 0; P.h = 0; P.open = false; }

// the flushes this wave owes: blocks whose last record one of its lanes wrote (converged control flow)
__device__ inline void bin_settle(const BinStage& S, const BinLog& L, double* E, int lane, BinLane& P) {
  unsigned long long m = __ballot(P.open && P.seen + 1u == (unsigned int)BIN_H);
  P.open = false;
  while (m) {
    const int leader = __ffsll((long long)m) - 1;
    const int fb = __shfl(P.b, leader), fh = __shfl(P.h, leader);
    bin_flush_block(S, L, E, fb, fh, lane);
    m &= m - 1ull;
  }
}

// One deposit per lane with `active` (cell ic, 0-based; value v).  MUST be called by every lane of the wave in
// converged control flow; bin_settle(P) must follow the last call of a loop before the wave does anything that other
// waves may wait for.
__device__ inline void bin_deposit(const BinStage& S, const BinLog& L, double* E, int lane, BinLane& P, bool active, int ic, double v) {
  const int b = active ? (ic >> L.shift) : 0;
  unsigned int pos = 0u;
  unsigned long long ep2 = 0ull;
  if (active) {
    pos = atomicAdd(&S.resv[b], 1u);
    // both halves' epochs in one read, issued with the reservation: a stale value can only be too SMALL (the wait
    // loop below reads again), never wrongly equal -- a half reaches epoch e only after its block e - 1 was flushed,
    // and stays there until block e, which needs this lane's record, is complete
    ep2 = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&S.epoch[b * 2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  bin_settle(S, L, E, lane, P);  // (the last call's bumps have returned by now)
  const unsigned int k = pos / BIN_H;
  const int h = (int)(k & 1u);
  const unsigned int e = k >> 1;
  unsigned int ep = h ? (unsigned int)(ep2 >> 32) : (unsigned int)ep2;
  bool pend = active;
  for (int spin = 0;; ++spin) {
    if (pend && ep == e) {
      const int slot = (b * 2 + h) * BIN_H + (int)(pos % BIN_H);
      S.keys[slot] = (unsigned int)ic;
      S.vals[slot] = v;
      __threadfence_block();  // the record is in place before it is counted
      P.seen = atomicAdd(&S.done[b * 2 + h], 1u);
      P.b = b; P.h = h; P.open = true;
      pend = false;
    }
    if (__ballot(pend) == 0ull) break;
    // (rare) the half is still being flushed: by another wave, or by this one -- settle first, then look again
    bin_settle(S, L, E, lane, P);
    __builtin_amdgcn_s_sleep(1);
    if (pend) ep = bin_ldu(&S.epoch[b * 2 + h]);
    if (spin > (1 << 22)) { if (pend) atomic_add_f64(&E[ic], v); pend = false; }  // (never: a logic error must not hang the GPU)
  }
}

// End of a launch (after a __syncthreads() behind the last deposit): what is left in the half-buffers goes to the
// array directly -- at most 63 records per bucket and workgroup -- and the workgroup publishes how many blocks it
// wrote to its parts of the log.
__device__ inline void bin_drain(const BinStage& S, const BinLog& L, double* E) {
  for (int b = 0; b < L.n_buckets; ++b) {
    const unsigned int r = S.resv[b];
    const unsigned int k = r / BIN_H, n = r % BIN_H;
    const int h = (int)(k & 1u);
    for (unsigned int l = threadIdx.x; l < n; l += blockDim.x) {
      const int slot = (b * 2 + h) * BIN_H + (int)l;
      atomic_add_f64(&E[S.keys[slot]], S.vals[slot]);
    }
    if (threadIdx.x == 0 && n) atomicAdd(&L.stats[1], (unsigned long long)n);
  }
  // (the blocks the workgroup WANTED to write: the fold takes min(count, cap), the plan of the next chunk the demand)
  for (int b = threadIdx.x; b < L.n_buckets; b += blockDim.x) L.count[(size_t)b * L.n_parts + blockIdx.x] = S.wcur[b];
}

// Fold: workgroup (bucket b = blockIdx.x / split, s = blockIdx.x % split) sums the blocks of the parts s, s + split, ...
// of bucket b's region into the bucket's slice in LDS and adds the slice to E.  The counts are left as they are
// (k_plan_bins reads and clears them).
#ifdef MCGPU_LANE_EMULATION
double bin_fold_slice[1 << 14];
#endif
static __global__ void __launch_bounds__(1024) k_fold_bins(const BinLog L, double* E, int n_cells, int split) {
#ifdef MCGPU_LANE_EMULATION
  double* const slice = bin_fold_slice;
#else
  extern __shared__ double slice[];
#endif
  const int b = blockIdx.x / split, s = blockIdx.x % split;
  const int n_slice = 1 << L.shift;
  for (int i = threadIdx.x; i < n_slice; i += blockDim.x) slice[i] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_waves = blockDim.x >= 64 ? (int)(blockDim.x >> 6) : 1;
  const unsigned int base = (unsigned int)b << L.shift;
  const unsigned int cap = L.cap[b];
  const size_t first = (size_t)L.off[b];
  bool any = false;
  for (int part = s; part < L.n_parts; part += split) {
    const unsigned int cnt = L.count[(size_t)b * L.n_parts + part];
    const unsigned int n_blk = cnt < cap ? cnt : cap;
    const size_t p0 = first + (size_t)cap * part;
    // (two blocks in flight per wave: the loads of the second are issued before the first is added)
    for (unsigned int blk = (unsigned int)wave; blk < n_blk; blk += 2u * (unsigned int)n_waves) {
      const bool two = blk + (unsigned int)n_waves < n_blk;
      for (int l = lane; l < BIN_H; l += BIN_WAVE) {
        const size_t at0 = (p0 + blk) * BIN_H + l;
        const size_t at1 = two ? (p0 + blk + n_waves) * BIN_H + l : at0;
        const unsigned int k0 = L.keys[at0], k1 = L.keys[at1];
        const double v0 = L.vals[at0], v1 = L.vals[at1];
        atomicAdd(&slice[k0 - base], v0);
        if (two) atomicAdd(&slice[k1 - base], v1);
      }
      any = true;
    }
  }
  if (!__syncthreads_or(any ? 1 : 0)) return;
  for (int i = threadIdx.x; i < n_slice; i += blockDim.x) {
    const double e = slice[i];
    const size_t c = (size_t)base + i;
    if (e != 0.0 && c < (size_t)n_cells) atomic_add_f64(&E[c], e);
  }
}

// The first chunk's regions: the log split evenly.  One workgroup.
static __global__ void k_plan_uniform(unsigned int* off, unsigned int* cap, int n_buckets, unsigned long long total_blocks, int n_parts) {
  const unsigned long long per = total_blocks / (unsigned long long)n_buckets / (unsigned long long)n_parts;
  for (int b = threadIdx.x; b < n_buckets; b += blockDim.x) {
    off[b] = (unsigned int)(per * n_parts * b);
    cap[b] = (unsigned int)per;
  }
}

// Plan of the next chunk's regions from what the last chunk wrote: every bucket gets blocks in proportion to its
// count times `growth` (the next chunk's packets over the last one's) plus a half plus a floor -- cut down in
// proportion when the log is smaller than that --, split evenly among n_parts_next workgroups (they take packets from
// one global counter, so their shares are even); the counts are cleared.  One workgroup.
static __global__ void __launch_bounds__(128) k_plan_bins(const BinLog L, unsigned int* off, unsigned int* cap, unsigned long long total_blocks,
                                                  double growth, int n_parts_next, double* want /* [n_buckets] scratch */) {
  for (int b = threadIdx.x; b < L.n_buckets; b += blockDim.x) {
    unsigned long long n = 0ull;
    for (int p = 0; p < L.n_parts; ++p) { n += L.count[(size_t)b * L.n_parts + p]; L.count[(size_t)b * L.n_parts + p] = 0u; }
    want[b] = (double)n * growth * 1.5 + 8.0 * n_parts_next;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sum = 0.0;
    for (int i = 0; i < L.n_buckets; ++i) sum += want[i];
    const double scale = sum > (double)total_blocks ? (double)total_blocks / sum : 1.0;
    unsigned long long o = 0ull;
    for (int i = 0; i < L.n_buckets; ++i) {
      unsigned long long c = (unsigned long long)(want[i] * scale) / (unsigned long long)n_parts_next;
      if (o + c * n_parts_next > total_blocks) c = (total_blocks - o) / n_parts_next;
      off[i] = (unsigned int)o; cap[i] = (unsigned int)c;
      o += c * n_parts_next;
    }
  }
}

}  // namespace mcgpu
