// emu_kernel.cpp -- TEST INFRASTRUCTURE: compiles the DEVICE source
// (mcfost_amd/csrc/mc_device.hip.h) for the host with a one-lane emulation of
// the few HIP builtins it uses, so that the kernel's control flow can be
// debugged and regression-tested against the oracle without a GPU.  This is
// not a CPU path of the product: nothing in mcfost_amd/ references it, it is
// built only by tests/test_kernel_emulation.py, and it runs one lane.
#define MCGPU_LANE_EMULATION 1
#define _GNU_SOURCE 1
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __shared__
#define __launch_bounds__(...)
#define __forceinline__ inline
#define __ATOMIC_RELAXED_EMU 0
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline void __threadfence_block() {}

struct emu_dim3 { unsigned x, y, z; };
static emu_dim3 threadIdx{0, 0, 0}, blockIdx{0, 0, 0}, blockDim{1, 1, 1}, gridDim{1, 1, 1};
struct double2 { double x, y; };
static inline double2 make_double2(double a, double b) { return double2{a, b}; }
static inline unsigned long long __ballot(bool p) { return p ? 1ull : 0ull; }
static inline int __ffsll(long long m) { return __builtin_ffsll(m); }
static inline int __popcll(unsigned long long m) { return __builtin_popcountll(m); }
template <class T> static inline T __shfl(T v, int) { return v; }
template <class T> static inline T __shfl_down(T, int) { return T(0); }
template <class T> static inline T __shfl_up(T, int) { return T(0); }
template <class T> static inline T __shfl_xor(T, int) { return T(0); }
static inline unsigned long long wall_clock64() { return 0ull; }
static inline unsigned int atomicAdd(unsigned int* p, unsigned int v) { unsigned int o = *p; *p += v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
  unsigned long long o = *p; *p += v; return o;
}
#include <stdio.h>
static double* g_trace_base = nullptr;
static long g_trace_n = 0;
static inline void unsafeAtomicAdd(double* p, double v) {
  if (g_trace_base && p >= g_trace_base && p < g_trace_base + g_trace_n)
    fprintf(stderr, "dep %ld %.17g\n", (long)(p - g_trace_base), v);
  *p += v;
}
static inline void __syncthreads() {}
static inline int __syncthreads_or(int p) { return p; }
static inline unsigned long long atomicExch(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p += v; return o; }
static inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
static inline int atomicExch(int* p, int v) { int o = *p; *p = v; return o; }
static inline int atomicCAS(int* p, int cmp, int v) { int o = *p; if (o == cmp) *p = v; return o; }
static inline unsigned int atomicCAS(unsigned int* p, unsigned int cmp, unsigned int v) { unsigned int o = *p; if (o == cmp) *p = v; return o; }
static inline double __longlong_as_double(long long b) { double d; memcpy(&d, &b, 8); return d; }
static inline long long __double_as_longlong(double d) { long long b; memcpy(&b, &d, 8); return b; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p += v; return o; }
static inline float atomicAdd(float* p, float v) { float o = *p; *p += v; return o; }
namespace mcgpu {  // the device source's unfused helpers (mc_device.hip.h), for the host compiler
static inline double nd_mul(double a, double b) { volatile double r = a * b; return r; }
static inline double nd_add(double a, double b) { volatile double r = a + b; return r; }
static inline double sqrt_nonneg(double x) { return std::sqrt(x); }   // (the device's Newton sequence is correctly rounded on its domain)
static inline float nf_mul(float a, float b) { volatile float r = a * b; return r; }
static inline float nf_add(float a, float b) { volatile float r = a + b; return r; }
static inline float nf_sub(float a, float b) { volatile float r = a - b; return r; }
}  // namespace mcgpu
using std::fabs; using std::floor; using std::sqrt; using std::log; using std::exp; using std::fmax;
using std::fmin; using std::atan2; using std::acos; using std::cos; using std::copysign; using std::pow;

namespace mcgpu { double lds_raw[1 << 18]; }

#include "../../mcfost_amd/csrc/mc_device.hip.h"
#include "cross_cell_literal.h"
#include "../../mcfost_amd/csrc/mc_voronoi.hip.h"
#include "../../mcfost_amd/csrc/mc_voronoi_pool.hip.h"
#include "../../mcfost_amd/csrc/mc_mono.hip.h"
#include "../../mcfost_amd/csrc/mc_mono_voronoi.hip.h"
#include "../../mcfost_amd/csrc/mc_raytrace.hip.h"
#include "../../mcfost_amd/csrc/mc_raytrace_voronoi.hip.h"
#include "../../mcfost_amd/csrc/mc_roles.hip.h"
// k_tail's hand-over to the host (mc_tail.hip.h "The last packets on the host"): one emulated lane runs the packets one
// after the other, so "few packets are left" never interrupts a packet here -- this hook does, after a given number of events
static unsigned int g_tail_hook_events = 0u;
#define MCGPU_TAIL_TEST_HOOK(events_here) (g_tail_hook_events != 0u && (events_here) >= g_tail_hook_events)
#include "../../mcfost_amd/csrc/mc_tail.hip.h"
#include "../../oracle/mc_oracle.h"

#define EMU_CONV_POOL 1
#include "emu_conv.h"

// k_tail in rounds (MCGPU_EMU_TAIL_HOST = events per round): every packet is written back as a record after that many of
// its events (tail_packet's hand-over to the host) and taken up again from the record in the next round, until none is
// left -- the records must carry a packet's whole state, at any point of its life.  `launch(A, recs, n, next)` runs k_tail.
template <typename Launch>
static int emu_tail_rounds(RunArgs A, const std::vector<Rec<true>>& first, unsigned int n_first, Launch launch, int* err) {
  const char* e = getenv("MCGPU_EMU_TAIL_HOST");
  unsigned int next = 0u;
  if (!e) { unsigned int n = n_first; launch(A, (const void*)first.data(), &n, &next); return 0; }
  g_tail_hook_events = (unsigned int)atoi(e);
  std::vector<Rec<true>> cur(first.begin(), first.begin() + n_first), out(n_first ? n_first : 1);
  unsigned int n = n_first, rounds = 0u;
  while (n && !*err) {
    unsigned int ctl[2] = {0x80000000u, 0u};   // (done: so large that "few are left" never holds; only the hook hands over)
    A.tail_host_max = n; A.tail_done = &ctl[0]; A.tail_out = out.data(); A.tail_out_n = &ctl[1];
    next = 0u;
    launch(A, (const void*)cur.data(), &n, &next);
    // (records of packets without Stokes tracking are the shorter Rec<false>: the buffers are raw bytes of the larger type)
    n = ctl[1];
    cur.swap(out);
    if (out.size() < cur.size()) out.resize(cur.size());
    if (++rounds > 1000000u) return 31;
  }
  g_tail_hook_events = 0u;
  if (getenv("MCGPU_EMU_BIN_STATS")) fprintf(stderr, "tail: %u rounds of hand-over\n", rounds);
  return 0;
}

extern "C" int emu_run_thermal(const oracle_model* m, const oracle_opts* o, const double* E_prior, double* E_abs,
                               double* sed, double* n_sent, uint64_t* counters) {
  Conv cv(m);
  const DevModel& M = cv.M;
  const VoroGrid& G = cv.G;
  const bool voro = cv.voro;
  if (lds_bytes(M) + sizeof(double) * m->n_cells > sizeof(lds_raw)) return 31;

  const size_t nsed = (size_t)9 * m->n_lambda * m->N_thet * m->N_phi;
  memset(E_abs, 0, sizeof(double) * m->n_cells);
  memset(sed, 0, sizeof(double) * nsed);
  memset(n_sent, 0, sizeof(double) * m->n_lambda);
  unsigned long long cnt[24];
  memset(cnt, 0, sizeof(cnt));
  int err = 0;
  if (getenv("MCGPU_EMU_TRACE")) { g_trace_base = E_abs; g_trace_n = m->n_cells; }
  RunArgs A;
  memset(&A, 0, sizeof(A));
  A.seed = o->seed; A.first_packet = o->first_packet; A.n_packets = o->n_packets;
  A.qscale = o->n_replicas >= 1.0 ? o->n_replicas : 1.0;
  A.frozen = o->frozen; A.E_prior = E_prior; A.E_abs = E_abs; A.sed = sed; A.n_sent = n_sent;
  A.counters = cnt; A.next_packet = cnt + 12; A.err = &err;
  A.inner_iters = 8; A.flags = 0; A.flush_every = 4; A.min_active = 0;
  const bool pola = m->lsepar_pola && m->aniso_method == 1, dark = M.dark != nullptr, l3d = m->l3D != 0;
  if (voro && getenv("MCGPU_EMU_POOL")) {   // the pool schedule (mc_voronoi_pool.hip.h) with one lane: MCGPU_EMU_POOL = log2(records)
    PoolArgs PA;
    PA.log_rec = atoi(getenv("MCGPU_EMU_POOL")); PA.cache_log_ns = 6;
    if (PA.log_rec < 1 || PA.log_rec > VP_MAX_LOG_REC) return 31;
    if (lds_bytes(M) + ((size_t)12 << PA.cache_log_ns) + vp_lds_bytes(PA.log_rec) + 64 > sizeof(lds_raw)) return 31;
    std::vector<PRec> recs((size_t)1 << PA.log_rec);
    PA.recs = recs.data();
    A.flush_every = 4;
    VpBlob blob;
    blob.M = M; blob.A = A; blob.G = G;
    if (pola) k_thermal_voro_pool<true, 512>(M, A, G, PA, &blob); else k_thermal_voro_pool<false, 512>(M, A, G, PA, &blob);
    for (int q = 0; q < ORACLE_N_COUNTERS; ++q) counters[q] = cnt[q];
    return err;
  }
  if (voro && getenv("MCGPU_EMU_ROLES")) {  // the role schedule on a Voronoi grid, one lane
    int nsp = 1, ks = 2, fi = 3, eq = 128;
    sscanf(getenv("MCGPU_EMU_ROLES"), "%d,%d,%d,%d", &nsp, &ks, &fi, &eq);
    const int n_rec = RQ_MIN_REC, log_ns = 6;
    if (lds_bytes(M) + ((size_t)12 << log_ns) + rq_lds_bytes(true, n_rec) + 64 > sizeof(lds_raw)) return 31;
    A.flush_every = 4;
    if (M.mrw) { if (pola) k_thermal_voro_roles<true, true>(M, A, G, log_ns, n_rec, nsp, ks, fi, 65, eq);
                 else k_thermal_voro_roles<false, true>(M, A, G, log_ns, n_rec, nsp, ks, fi, 65, eq); }
    else if (pola) k_thermal_voro_roles<true>(M, A, G, log_ns, n_rec, nsp, ks, fi, 65, eq);
    else k_thermal_voro_roles<false>(M, A, G, log_ns, n_rec, nsp, ks, fi, 65, eq);
    for (int q = 0; q < ORACLE_N_COUNTERS; ++q) counters[q] = cnt[q];
    return err;
  }
  if (voro && M.n_classes) {  // lvariable_dust on a Voronoi grid (deposit cache, few slots)
    if (M.mrw) { if (pola) k_thermal_voro_var<true, true>(M, A, G, 6); else k_thermal_voro_var<false, true>(M, A, G, 6); }
    else { if (pola) k_thermal_voro_var<true, false>(M, A, G, 6); else k_thermal_voro_var<false, false>(M, A, G, 6); }
    for (int q = 0; q < ORACLE_N_COUNTERS; ++q) counters[q] = cnt[q];
    return err;
  }
  if (voro) {
    if (M.mrw) {   // the walk: the single-role kernel with HBM deposits
      if (pola) k_thermal_voro_mrw<true>(M, A, G); else k_thermal_voro_mrw<false>(M, A, G);
    } else if (getenv("MCGPU_EMU_LDS")) {  // deposit cache, few slots so that hits, misses and folds all occur
      if (pola) k_thermal_voro_cache<true, 512>(M, A, G, 6); else k_thermal_voro_cache<false, 512>(M, A, G, 6);
    } else {
      if (pola) k_thermal_voro<true>(M, A, G); else k_thermal_voro<false>(M, A, G);
    }
    for (int q = 0; q < ORACLE_N_COUNTERS; ++q) counters[q] = cnt[q];
    return err;
  }
  if (M.grid_sph) {  // spherical grid: the single-role kernel with that grid's operators
    const bool ld = getenv("MCGPU_EMU_LDS") != nullptr;
    if (dark || M.n_classes) {   // a dark zone and / or dust classes on the spherical grid (k_thermal_sph_ext)
      if (M.mrw) return 31;
      const bool var = M.n_classes != 0;
      if (dark && var) return 31;
#define RUNSX(a, b) do { if (dark) { if (ld) k_thermal_sph_ext<a, b, true, true, false>(M, A); else k_thermal_sph_ext<a, b, true, false, false>(M, A); } \
                         else { if (ld) k_thermal_sph_ext<a, b, false, true, true>(M, A); else k_thermal_sph_ext<a, b, false, false, true>(M, A); } } while (0)
      if (l3d) { if (pola) RUNSX(true, true); else RUNSX(true, false); }
      else { if (pola) RUNSX(false, true); else RUNSX(false, false); }
#undef RUNSX
      for (int q = 0; q < ORACLE_N_COUNTERS; ++q) counters[q] = cnt[q];
      return err;
    }
    if (l3d) { if (pola) { if (ld) k_thermal_sph<true, true, true>(M, A); else k_thermal_sph<true, true, false>(M, A); }
               else { if (ld) k_thermal_sph<true, false, true>(M, A); else k_thermal_sph<true, false, false>(M, A); } }
    else { if (pola) { if (ld) k_thermal_sph<false, true, true>(M, A); else k_thermal_sph<false, true, false>(M, A); }
           else { if (ld) k_thermal_sph<false, false, true>(M, A); else k_thermal_sph<false, false, false>(M, A); } }
    for (int q = 0; q < ORACLE_N_COUNTERS; ++q) counters[q] = cnt[q];
    return err;
  }
  if (M.n_classes) {  // lvariable_dust: the HBM-gather variant (MCGPU_EMU_LDS: with the private grid in LDS)
#define RUNV(a, b, c) do { if (getenv("MCGPU_EMU_LDS")) k_thermal_var<a, b, c, true>(M, A); else k_thermal_var<a, b, c, false>(M, A); } while (0)
    if (l3d) { if (pola) { if (dark) RUNV(true, true, true); else RUNV(true, true, false); }
               else { if (dark) RUNV(true, false, true); else RUNV(true, false, false); } }
    else { if (pola) { if (dark) RUNV(false, true, true); else RUNV(false, true, false); }
           else { if (dark) RUNV(false, false, true); else RUNV(false, false, false); } }
#undef RUNV
    for (int q = 0; q < ORACLE_N_COUNTERS; ++q) counters[q] = cnt[q];
    return err;
  }
  if (getenv("MCGPU_EMU_BIN")) {
    // Binned deposits with chunks (mcgpu.hip::launch_binned on one lane): MCGPU_EMU_BIN = "<packets per chunk>,<log blocks>,
    // <n_srv_pref>,<k_short>,<fly_iters>".  Every chunk but the last hands its unfinished packets on (carry_*).
    if (!l3d || M.n_classes) return 31;
    long chunk = 1000, log_blocks = 64;
    int nsp = 1, ks = 2, fi = 3, tail_thr = 0;   // tail_thr > 0: the last chunk hands its last packets to k_tail
    sscanf(getenv("MCGPU_EMU_BIN"), "%ld,%ld,%d,%d,%d,%d", &chunk, &log_blocks, &nsp, &ks, &fi, &tail_thr);
    int shift = 6;
    while (shift < 14 && ((m->n_cells + (1 << shift) - 1) >> shift) > 24) ++shift;
    const int nb = (m->n_cells + (1 << shift) - 1) >> shift;
    const int n_rec = RQ_MIN_REC;
    if (lds_bytes(M) + bin_lds_bytes(nb) + rq_lds_bytes(true, n_rec) + 64 > sizeof(lds_raw)) return 31;
    std::vector<unsigned int> keys((size_t)log_blocks * BIN_H), count(nb, 0u), off(nb), cap(nb);
    std::vector<double> vals((size_t)log_blocks * BIN_H), want(nb);
    unsigned long long stats[2] = {0ull, 0ull};
    BinLog L;
    L.keys = keys.data(); L.vals = vals.data(); L.count = count.data(); L.off = off.data(); L.cap = cap.data();
    L.stats = stats; L.n_buckets = nb; L.shift = shift; L.n_parts = 1;
    const size_t carry_cap = (size_t)n_rec + 1 + PK_BATCH;
    std::vector<Rec<true>> carry[2] = {std::vector<Rec<true>>(carry_cap), std::vector<Rec<true>>(carry_cap)};
    unsigned int carry_n[2] = {0u, 0u};
    const uint64_t n_total = o->n_packets;
    uint64_t done = 0, last = 0;
    int ichunk = 0;
    while (done < n_total) {
      const uint64_t c = (uint64_t)chunk < n_total - done ? (uint64_t)chunk : n_total - done;
      gridDim.x = 1; blockDim.x = 1; threadIdx.x = 0; blockIdx.x = 0;
      if (last == 0) k_plan_uniform(off.data(), cap.data(), nb, (unsigned long long)log_blocks, 1);
      else k_plan_bins(L, off.data(), cap.data(), (unsigned long long)log_blocks, (double)c / (double)last, 1, want.data());
      cnt[12] = 0;
      A.first_packet = o->first_packet + done; A.n_packets = c;
      A.n_folded = A.frozen ? 0.0 : (double)done;
      A.bin = L;
      const int in = ichunk & 1, out = in ^ 1;
      const bool fin = done + c >= n_total;
      A.carry_in = carry[in].data(); A.carry_in_n = &carry_n[in];
      const bool to_tail = fin && tail_thr > 0;
      A.carry_out = (fin && !to_tail) ? nullptr : carry[out].data(); A.carry_out_n = &carry_n[out];
      A.carry_cap = (unsigned int)carry_cap;
      A.tail_threshold = to_tail ? tail_thr : 0;
      carry_n[out] = 0u;
#define RUNB(b, c) do { if (M.mrw) k_thermal_roles_bin<b, c, true>(M, A, n_rec, nsp, ks, fi, 65, 128); else k_thermal_roles_bin<b, c, false>(M, A, n_rec, nsp, ks, fi, 65, 128); } while (0)
      if (pola) { if (dark) RUNB(true, true); else RUNB(true, false); }
      else { if (dark) RUNB(false, true); else RUNB(false, false); }
#undef RUNB
      if (err) return err;
      for (int b = 0; b < nb; ++b) { blockIdx.x = (unsigned)b; k_fold_bins(L, E_abs, m->n_cells, 1); }
      blockIdx.x = 0;
      if (to_tail) {
        unsigned int next = 0u;
        RunArgs At = A;
        At.n_folded = 0.0;
#define RUNK(b, c) do { if (M.mrw) k_tail<true, b, c, true>(M, Ax, recs, np, nx); else k_tail<true, b, c, false>(M, Ax, recs, np, nx); } while (0)
        const int rct = emu_tail_rounds(At, carry[out], carry_n[out], [&](const RunArgs& Ax, const void* recs, unsigned int* np, unsigned int* nx) {
          if (pola) { if (dark) RUNK(true, true); else RUNK(true, false); }
          else { if (dark) RUNK(false, true); else RUNK(false, false); } }, &err);
#undef RUNK
        (void)next;
        if (rct) return rct;
        if (err) return err;
        if (getenv("MCGPU_EMU_BIN_STATS")) fprintf(stderr, "tail: %u packets\n", carry_n[out]);
      }
      done += c; last = c; ++ichunk;
    }
    for (int q = 0; q < ORACLE_N_COUNTERS; ++q) counters[q] = cnt[q];
    if (getenv("MCGPU_EMU_BIN_STATS")) fprintf(stderr, "binned: %d chunks, %d buckets, overflow blocks %llu, drained %llu\n", ichunk, nb, stats[0], stats[1]);
    return err;
  }
  if (getenv("MCGPU_EMU_ROLES")) {  // the role schedule on one lane: the one wave alternates between both roles
    // MCGPU_EMU_ROLES = "<n_srv_pref>,<k_short>,<fly_iters>,<emit_qmax>": 1,... the wave prefers to serve (it flies when
    // there is nothing to serve or emit), 0,... it prefers to fly (it serves only while the FLY ring is empty)
    int nsp = 1, ks = 2, fi = 3, eq = 128;
    sscanf(getenv("MCGPU_EMU_ROLES"), "%d,%d,%d,%d", &nsp, &ks, &fi, &eq);
    const bool ld = getenv("MCGPU_EMU_LDS") != nullptr;
    const int n_rec = RQ_MIN_REC;
    if (lds_bytes(M) + sizeof(double) * m->n_cells + rq_lds_bytes(true, n_rec) + 64 > sizeof(lds_raw)) return 31;
    A.flush_every = 4;
    if (getenv("MCGPU_EMU_TAIL") && !l3d) {   // the role kernel hands its last packets to k_tail (mc_tail.hip.h)
      const size_t carry_cap = (size_t)n_rec + 1 + PK_BATCH;
      std::vector<Rec<true>> carry(carry_cap);
      unsigned int carry_n = 0u, next = 0u;
      RunArgs Ar = A;
      Ar.carry_out = carry.data(); Ar.carry_out_n = &carry_n; Ar.carry_cap = (unsigned int)carry_cap;
      Ar.tail_threshold = atoi(getenv("MCGPU_EMU_TAIL"));
#define RUNT(b, c, mm) do { if (ld) k_thermal_roles_tail<b, c, true, mm>(M, Ar, n_rec, nsp, ks, fi, 65, eq); else k_thermal_roles_tail<b, c, false, mm>(M, Ar, n_rec, nsp, ks, fi, 65, eq); \
                            if (!err) { const int rct = emu_tail_rounds(A, carry, carry_n, [&](const RunArgs& Ax, const void* recs, unsigned int* np, unsigned int* nx) { \
                              k_tail<false, b, c, mm>(M, Ax, recs, np, nx); }, &err); if (rct) return rct; } } while (0)
      if (M.mrw) { if (pola) { if (dark) RUNT(true, true, true); else RUNT(true, false, true); } else { if (dark) RUNT(false, true, true); else RUNT(false, false, true); } }
      else { if (pola) { if (dark) RUNT(true, true, false); else RUNT(true, false, false); } else { if (dark) RUNT(false, true, false); else RUNT(false, false, false); } }
#undef RUNT
      if (getenv("MCGPU_EMU_BIN_STATS")) fprintf(stderr, "tail: %u packets\n", carry_n);
      for (int q = 0; q < ORACLE_N_COUNTERS; ++q) counters[q] = cnt[q];
      return err;
    }
    if (getenv("MCGPU_EMU_PARAM")) {   // the flight-parametric crossing in the flying role (option "crossing" = 1): 2D, LDS deposits
      if (l3d || dark || M.mrw) return 31;
      if (pola) k_thermal_roles_param<true, false>(M, A, n_rec, nsp, ks, fi, 65, eq); else k_thermal_roles_param<false, false>(M, A, n_rec, nsp, ks, fi, 65, eq);
      for (int q = 0; q < ORACLE_N_COUNTERS; ++q) counters[q] = cnt[q];
      return err;
    }
#define RUNR(a, b, c) do { if (ld) k_thermal_roles<a, b, c, true>(M, A, n_rec, nsp, ks, fi, 65, eq); else k_thermal_roles<a, b, c, false>(M, A, n_rec, nsp, ks, fi, 65, eq); } while (0)
#define RUNRM(b, c) do { if (ld) k_thermal_roles<false, b, c, true, true>(M, A, n_rec, nsp, ks, fi, 65, eq); else k_thermal_roles<false, b, c, false, true>(M, A, n_rec, nsp, ks, fi, 65, eq); } while (0)
    if (M.mrw) {
      if (l3d) return 31;
      if (pola) { if (dark) RUNRM(true, true); else RUNRM(true, false); }
      else { if (dark) RUNRM(false, true); else RUNRM(false, false); }
    } else if (l3d) {
      if (pola) { if (dark) RUNR(true, true, true); else RUNR(true, true, false); }
      else { if (dark) RUNR(true, false, true); else RUNR(true, false, false); }
    } else {
      if (pola) { if (dark) RUNR(false, true, true); else RUNR(false, true, false); }
      else { if (dark) RUNR(false, false, true); else RUNR(false, false, false); }
    }
    for (int q = 0; q < ORACLE_N_COUNTERS; ++q) counters[q] = cnt[q];
    return err;
  }
#define RUN(a, b, c) do { if (getenv("MCGPU_EMU_LDS")) k_thermal_lds<a, b, c>(M, A); else k_thermal<a, b, c>(M, A); } while (0)
#define RUNM(b, c) do { if (getenv("MCGPU_EMU_LDS")) k_thermal_lds<false, b, c, true>(M, A); else k_thermal<false, b, c, true>(M, A); } while (0)
  if (M.mrw) {
    if (l3d) return 31;
    if (pola) { if (dark) RUNM(true, true); else RUNM(true, false); }
    else { if (dark) RUNM(false, true); else RUNM(false, false); }
  } else if (l3d) {
    if (pola) { if (dark) RUN(true, true, true); else RUN(true, true, false); }
    else { if (dark) RUN(true, false, true); else RUN(true, false, false); }
  } else {
    if (pola) { if (dark) RUN(false, true, true); else RUN(false, true, false); }
    else { if (dark) RUN(false, false, true); else RUN(false, false, false); }
  }
  for (int q = 0; q < ORACLE_N_COUNTERS; ++q) counters[q] = cnt[q];
  return err;
}


// ---- one cell crossing by the product's branch-free cross_cell_lean (literal = 0) or by the branch-for-branch
// restatement of cross_cylindrical_cell kept in tests/emu/cross_cell_literal.h (literal = 1) ----------------
extern "C" int emu_cross_cell(const oracle_model* m, int literal, int n, const double* x0, const double* y0, const double* z0,
                              const double* u, const double* v, const double* w, const int* cell, double* x1, double* y1,
                              double* z1, int* next_cell, double* l) {
  Conv cv(m);
  const DevModel& M = cv.M;
  if (cv.voro || lds_bytes(M) > sizeof(lds_raw)) return 31;
  const Lds T = lds_carve(lds_raw, M);
  lds_stage(T, M);
  for (int i = 0; i < n; ++i) {
    const double a = u[i] * u[i] + v[i] * v[i];
    const double inv_a = (a > TINY_REAL) ? 1.0 / a : HUGE_REAL;
    const double inv_w = (fabs(w[i]) > TINY_REAL) ? 1.0 / w[i] : copysign(HUGE_DP, w[i]);
    const int c = cell[i] - 1;
    const int ri = m->cell_map_i[c], zj = m->cell_map_j[c], k = m->cell_map_k[c];
    int ri1, zj1, k1;
    if (M.grid_sph) {
      if (m->l3D) cross_cell_sph<true>(T, M, x0[i], y0[i], z0[i], u[i], v[i], w[i], ri, zj, k, x1[i], y1[i], z1[i], ri1, zj1, k1, l[i]);
      else cross_cell_sph<false>(T, M, x0[i], y0[i], z0[i], u[i], v[i], w[i], ri, zj, k, x1[i], y1[i], z1[i], ri1, zj1, k1, l[i]);
    } else if (m->l3D) {
      if (literal) cross_cell<true>(T, M, x0[i], y0[i], z0[i], u[i], v[i], w[i], inv_a, inv_w, ri, zj, k, x1[i], y1[i], z1[i], ri1, zj1, k1, l[i]);
      else cross_cell_lean<true>(T, M, x0[i], y0[i], z0[i], u[i], v[i], w[i], inv_a, inv_w, ri, zj, k, x1[i], y1[i], z1[i], ri1, zj1, k1, l[i]);
    } else {
      if (literal) cross_cell<false>(T, M, x0[i], y0[i], z0[i], u[i], v[i], w[i], inv_a, inv_w, ri, zj, k, x1[i], y1[i], z1[i], ri1, zj1, k1, l[i]);
      else cross_cell_lean<false>(T, M, x0[i], y0[i], z0[i], u[i], v[i], w[i], inv_a, inv_w, ri, zj, k, x1[i], y1[i], z1[i], ri1, zj1, k1, l[i]);
    }
    next_cell[i] = icell_of(M.n_rad, M.nz, M.n_az, M.l3D, ri1, zj1, k1);
  }
  return 0;
}

// ---- SED mode: the host orchestration of mcgpu_run_mono with one emulated lane --------------
extern "C" int emu_run_mono(const oracle_model* m, const oracle_mono_opts* o, double* xI, double* sed, double* n_sent,
                            uint64_t* n_sent_chunk, uint64_t* counters) {
  Conv cv(m);
  const DevModel& M = cv.M;
  const VoroGrid& G = cv.G;
  const bool voro = cv.voro;

  const size_t nsed = (size_t)9 * m->n_lambda * m->N_thet * m->N_phi;
  const int nRT = m->RT_n_incl * m->RT_n_az;
  const size_t nxI = o->rt1 ? (size_t)m->n_az_rt * m->n_theta_rt * m->N_type_flux * nRT * (size_t)m->n_cells : 0;
  memset(sed, 0, sizeof(double) * nsed);
  memset(n_sent, 0, sizeof(double) * m->n_lambda);
  const bool xi32 = getenv("MCGPU_EMU_XI_F32") != nullptr;  // default-real records (mcgpu_set_xI_precision(4))
  const bool pola_x = m->N_type_flux == 4 || m->N_type_flux == 8;
  const Xi32Lay xi_lay = xi32_layout(nRT ? nRT : 1, pola_x, m->lsepar_contrib != 0);   // (the packed default-real layout)
  const int xi_binf = xi_lay.binf;
  std::vector<double> xI_dev(nxI ? nxI / m->N_type_flux / (nRT ? nRT : 1) * (xi32 ? (size_t)(xi_binf + 1) / 2 : (size_t)nRT * XI_LINE) : 1, 0.0);  // the kernel's own layout (FP32: half of it used)
  unsigned long long cnt[24];
  memset(cnt, 0, sizeof(cnt));
  int err = 0;
  const bool pola = m->lsepar_pola && m->aniso_method == 1, dark = M.dark != nullptr, l3d = m->l3D != 0;
  MonoArgs A;
  memset(&A, 0, sizeof(A));
  A.seed = o->seed; A.lambda = o->lambda; A.p_lambda = o->p_lambda; A.capt_sup = o->capt_sup; A.rt1 = o->rt1;
  A.frac_E_stars = m->frac_E_stars[o->lambda - 1]; A.frac_E_disk = m->frac_E_disk[o->lambda - 1];
  A.prob_E_cell = m->prob_E_cell ? m->prob_E_cell + (size_t)(m->n_cells + 1) * (o->lambda - 1) : nullptr;
  A.n_chunks = o->n_chunks; A.first_chunk = o->first_chunk;
  A.RT_n_incl = m->RT_n_incl > 0 ? m->RT_n_incl : 1; A.nRT = o->rt1 ? nRT : 0;
  A.rt_u = m->tab_u_rt; A.rt_v = m->tab_v_rt; A.rt_w = m->tab_w_rt;
  A.n_az_rt = m->n_az_rt; A.n_theta_rt = m->n_theta_rt; A.N_type_flux = m->N_type_flux; A.contrib = m->lsepar_contrib;
  A.s11 = m->tab_s11_pos ? m->tab_s11_pos + (size_t)(m->nang_scatt + 1) * (o->p_lambda - 1) : nullptr;
  A.xI = xI_dev.data(); A.xI_f32 = xi32 ? 1 : 0; A.xi = xi_lay;
  A.sed = sed; A.n_sent = n_sent; A.counters = cnt; A.next_item = cnt + 8; A.err = &err;
  A.inner_iters = 8; A.min_active = 0;
#define MONO(sc_) do {                                                                     \
    if (voro) { if (pola) k_mono_voro<true, sc_>(M, A, G); else k_mono_voro<false, sc_>(M, A, G); }               \
    else if (M.grid_sph) {                                                                     \
      if (l3d) { if (pola) k_mono_sph<true, true, sc_>(M, A); else k_mono_sph<true, false, sc_>(M, A); }            \
      else { if (pola) k_mono_sph<false, true, sc_>(M, A); else k_mono_sph<false, false, sc_>(M, A); }              \
    } else if (l3d) {                                                                            \
      if (pola) { if (dark) k_mono<true, true, true, sc_>(M, A); else k_mono<true, true, false, sc_>(M, A); }      \
      else { if (dark) k_mono<true, false, true, sc_>(M, A); else k_mono<true, false, false, sc_>(M, A); }         \
    } else {                                                                              \
      if (pola) { if (dark) k_mono<false, true, true, sc_>(M, A); else k_mono<false, true, false, sc_>(M, A); }    \
      else { if (dark) k_mono<false, false, true, sc_>(M, A); else k_mono<false, false, false, sc_>(M, A); }       \
    } } while (0)
  const int nc = o->n_chunks;
  const unsigned long long lim = (unsigned long long)std::ceil(o->n_phot_lim);
  std::vector<unsigned long long> need(nc, (unsigned long long)o->n_photons2), sent(nc, 0ull), base(nc + 1, 0ull);
  std::vector<int> active;
  if (o->n_photons2 > 0 && lim > 0) for (int c = 0; c < nc; ++c) active.push_back(c);
  while (!active.empty()) {
    const int na = (int)active.size();
    const unsigned long long batch = 64;  // small batches: several scout rounds per stream
    std::vector<unsigned char> hits((size_t)na * batch, 0);
    A.active = active.data(); A.seq0 = sent.data(); A.batch = batch; A.hits = hits.data(); A.n_items = (size_t)na * batch;
    cnt[8] = 0;
    MONO(true);
    if (err) return err;
    std::vector<int> still;
    for (int a = 0; a < na; ++a) {  // what k_mono_scan does with one wave per stream
      const int c = active[a];
      unsigned long long usable = batch;
      if (sent[c] + batch >= lim) usable = lim > sent[c] ? lim - sent[c] : 0;
      unsigned long long s = 0;
      bool fin = false;
      for (; s < usable; ++s)
        if (hits[(size_t)a * batch + s] && --need[c] == 0) { ++s; fin = true; break; }
      if (!fin && usable < batch) { s = usable; fin = true; }
      sent[c] += fin ? s : batch;
      if (!fin) still.push_back(c);
    }
    active.swap(still);
  }
  for (int c = 0; c < nc; ++c) { base[c + 1] = base[c] + sent[c]; n_sent_chunk[c] = sent[c]; }
  A.item_base = base.data(); A.n_items = base[nc]; A.active = nullptr; A.seq0 = nullptr; A.hits = nullptr; A.batch = 0;
  cnt[8] = 0;
  if (A.n_items) MONO(false);
  if (nxI) {  // mcgpu_fetch_xI
    gridDim.x = 1; blockDim.x = 1; threadIdx.x = 0;
    for (size_t i = 0; i < nxI; ++i) {
      blockIdx.x = (unsigned)i;
      k_xI_fetch(xI_dev.data(), nullptr, xI, m->n_az_rt, m->n_theta_rt, m->N_type_flux, nRT, nxI, xi32 ? 1 : 0, xi_lay, pola_x ? 4 : 1);
    }
    blockIdx.x = 0;
  }
  for (int q = 0; q < 8; ++q) counters[q] = cnt[q];
  return err;
}


// ---- RT1 ray tracing: mcgpu_rt1_dust_map / mcgpu_rt1_image with one emulated lane -----------------------------
struct EmuRt {
  Conv cv;
  std::vector<double> xI_dev, J;
  RtArgs A;
  EmuRt(const oracle_model* m, const oracle_rt_opts* o, const double* xI, const float* Tdust) : cv(m) {
    const DevModel& M = cv.M;
    const int nRT = m->RT_n_incl * m->RT_n_az, ntf = m->N_type_flux;
    const size_t st_type = (size_t)m->n_az_rt * m->n_theta_rt, st_rt = st_type * ntf;
    // reference layout (n_az_rt, n_theta_rt, N_type_flux, nRT, n_cells) -> engine layout [cell][psup][phik][iRT][XI_LINE]
    xI_dev.assign((size_t)m->n_cells * st_type * nRT * XI_LINE, 0.0);
    for (int ic = 0; xI && ic < m->n_cells; ++ic)   // (xI = null: method 2, which reads its own source function)
      for (int q = 0; q < nRT; ++q)
        for (int t = 0; t < ntf; ++t)
          for (int ps = 0; ps < m->n_theta_rt; ++ps)
            for (int k = 0; k < m->n_az_rt; ++k)
              xI_dev[((((size_t)ic * m->n_theta_rt + ps) * m->n_az_rt + k) * nRT + q) * XI_LINE + t] =
                  xI[(size_t)k + (size_t)m->n_az_rt * ps + st_type * t + st_rt * ((size_t)q + (size_t)nRT * ic)];
    J.assign(m->n_cells, 0.0);
    memset(&A, 0, sizeof(A));
    A.lambda = o->lambda; A.RT_n_incl = m->RT_n_incl; A.nRT = nRT; A.n_az_rt = m->n_az_rt; A.n_theta_rt = m->n_theta_rt;
    A.N_type_flux = ntf; A.contrib = m->lsepar_contrib; A.l_sym_ima = o->l_sym_ima;
    A.wl = o->wl_um * 1.e-6;
    const double AU_to_cm = 149597870700.0 * 100.0, pc_to_AU = 648000.0 / M_PI;
    A.photon_energy = o->E_src * o->wl_um * 1.0e-6 / (o->n_sent_photons * AU_to_cm * M_PI);
    A.pix_scale = 1.0 / (o->distance * pc_to_AU);
    A.ang_disque = o->ang_disque; A.tau_dark_zone_obs = o->tau_dark_zone_obs;
    A.rmin_RT = 0.01 * o->Rmin;
    A.fact_r = std::exp((1.0 / ((double)RT_N_RAD - 1)) * std::log(2.0 * o->Rmax / A.rmin_RT));
    A.fact_A = std::sqrt(M_PI * (A.fact_r - 1.0 / A.fact_r) / RT_N_PHI);
    A.cst_phi = (o->l_sym_ima ? M_PI : 2 * M_PI) / (double)RT_N_PHI;
    A.l_far = 10. * o->Rmax;
    A.rt_u = m->tab_u_rt; A.rt_v = m->tab_v_rt; A.rt_w = m->tab_w_rt; A.rt_az = o->tab_RT_az;
    A.xI = xI_dev.data(); A.J_th = J.data();
    gridDim.x = 1; blockDim.x = 1; threadIdx.x = 0;
    for (int ic = 0; ic < m->n_cells; ++ic) { blockIdx.x = (unsigned)ic; k_calc_Jth(M, o->lambda, A.wl, Tdust, J.data()); }
    blockIdx.x = 0;
  }
};

extern "C" int emu_rt1_dust_map(const oracle_model* m, const oracle_rt_opts* o, const double* xI, const float* Tdust,
                                double* out) {
  EmuRt E(m, o, xI, Tdust);
  const int ntf = m->N_type_flux;
  memset(out, 0, sizeof(double) * (size_t)E.A.nRT * ntf);
  E.A.out = out;
  const bool pola = ntf == 4 || ntf == 8;
  if (E.cv.voro) { if (pola) k_rt1_dust_map_voro<true>(E.cv.M, E.A, E.cv.G); else k_rt1_dust_map_voro<false>(E.cv.M, E.A, E.cv.G); }
  else if (m->l3D) { if (pola) k_rt1_dust_map<true, true>(E.cv.M, E.A); else k_rt1_dust_map<true, false>(E.cv.M, E.A); }
  else { if (pola) k_rt1_dust_map<false, true>(E.cv.M, E.A); else k_rt1_dust_map<false, false>(E.cv.M, E.A); }
  return 0;
}

// step 4 of define_dark_zone: the device's ray kernel, one thread at a time; flag[n_cells] receives the cells with a ray
// that does not leave
extern "C" int emu_dark_zone_rays(const oracle_model* m, int lambda, double tau_max, int i_lo, int i_hi, const int* zj_sup,
                                  const double* r_grid, const double* z_grid, const unsigned char* dark_now, unsigned char* flag) {
  if (m->l3D || m->grid_type != 1) return 31;
  Conv cv(m);
  memset(flag, 0, m->n_cells);
  const long long n_rays = 11LL * (i_hi - i_lo + 1) * m->nz;
  gridDim.x = (unsigned)n_rays; blockDim.x = 1; threadIdx.x = 0;
  for (long long t = 0; t < n_rays; ++t) {
    blockIdx.x = (unsigned)t;
    k_dark_zone_rays(cv.M, lambda, (float)tau_max, i_lo, i_hi, zj_sup, r_grid, z_grid, dark_now, flag);
  }
  blockIdx.x = 0;
  return 0;
}

// k_init_reemission, one (class, T) row per call; kabs[class][lambda], outputs [class][T] and [class][T][lambda]
extern "C" int emu_init_reemission_ex(int n_classes, int n_T, int n_lambda, const float* tab_Temp, const double* tab_lambda,
                                      const double* tab_delta_lambda, const double* kabs, const double* dudt, const double* hnorm,
                                      double ufac, double* lq, double* cdf) {
  gridDim.x = (unsigned)(n_classes * n_T); blockDim.x = 1; threadIdx.x = 0;
  for (unsigned b = 0; b < gridDim.x; ++b) {
    blockIdx.x = b;
    k_init_reemission(n_classes, n_T, n_lambda, tab_Temp, tab_lambda, tab_delta_lambda, kabs, lq, cdf, dudt, hnorm, ufac);
  }
  blockIdx.x = 0;
  return 0;
}
extern "C" int emu_init_reemission(int n_classes, int n_T, int n_lambda, const float* tab_Temp, const double* tab_lambda,
                                   const double* tab_delta_lambda, const double* kabs, double* lq, double* cdf) {
  return emu_init_reemission_ex(n_classes, n_T, n_lambda, tab_Temp, tab_lambda, tab_delta_lambda, kabs, nullptr, nullptr, 0.0, lq, cdf);
}

extern "C" int emu_stars_map_sed(const oracle_model* m, const oracle_rt_opts* o, uint64_t seed, const double* star_flux,
                                 double* out) {
  Conv cv(m);
  RtArgs A;
  memset(&A, 0, sizeof(A));
  A.lambda = o->lambda; A.RT_n_incl = m->RT_n_incl; A.nRT = m->RT_n_incl * m->RT_n_az; A.ang_disque = o->ang_disque;
  A.rt_u = m->tab_u_rt; A.rt_v = m->tab_v_rt; A.rt_w = m->tab_w_rt; A.rt_az = o->tab_RT_az;
  if (cv.voro) A.voro = &cv.G;
  for (int q = 0; q < A.nRT; ++q) out[q] = 0.0;
  gridDim.x = (unsigned)(A.nRT * m->n_stars); blockDim.x = 1; threadIdx.x = 0;
  for (unsigned b = 0; b < gridDim.x; ++b) {
    blockIdx.x = b;
    if (m->l3D) k_stars_map_sed<true>(cv.M, A, (unsigned)seed, (unsigned)(seed >> 32), star_flux, out);
    else k_stars_map_sed<false>(cv.M, A, (unsigned)seed, (unsigned)(seed >> 32), star_flux, out);
  }
  blockIdx.x = 0;
  return 0;
}

extern "C" int emu_tau_maps(const oracle_model* m, const oracle_rt_opts* o, int npix_x, int npix_y, double map_size, double zoom,
                            float tau, float* tau_map, float* surf_map) {
  Conv cv(m);
  RtArgs A;
  memset(&A, 0, sizeof(A));
  A.lambda = o->lambda; A.RT_n_incl = m->RT_n_incl; A.nRT = m->RT_n_incl * m->RT_n_az; A.ang_disque = o->ang_disque;
  A.rt_u = m->tab_u_rt; A.rt_v = m->tab_v_rt; A.rt_w = m->tab_w_rt; A.rt_az = o->tab_RT_az;
  A.npix_x = npix_x; A.npix_y = npix_y;
  A.taille_pix = (map_size / zoom) / (double)(npix_x > npix_y ? npix_x : npix_y);
  A.l_far = 10.0 * o->Rmax;
  gridDim.x = 1; blockDim.x = 1; threadIdx.x = 0; blockIdx.x = 0;
  if (cv.voro) k_tau_maps_voro(cv.M, A, cv.G, tau, tau_map, surf_map);
  else if (m->l3D) k_tau_maps<true>(cv.M, A, tau, tau_map, surf_map);
  else k_tau_maps<false>(cv.M, A, tau, tau_map, surf_map);
  return 0;
}

extern "C" int emu_rt1_image(const oracle_model* m, const oracle_rt_opts* o, int npix_x, int npix_y, double map_size,
                             double zoom, const double* xI, const float* Tdust, double* image, int* n_rays) {
  EmuRt E(m, o, xI, Tdust);
  const int ntf = m->N_type_flux;
  memset(image, 0, sizeof(double) * (size_t)E.A.nRT * ntf * npix_x * npix_y);
  unsigned long long rays = 0;
  E.A.npix_x = npix_x; E.A.npix_y = npix_y;
  E.A.npix_x_max = o->l_sym_ima ? npix_x / 2 + npix_x % 2 : npix_x;
  E.A.taille_pix = (map_size / zoom) / (double)(npix_x > npix_y ? npix_x : npix_y);
  E.A.image = image; E.A.n_rays = &rays;
  const bool pola = ntf == 4 || ntf == 8;
  if (E.cv.voro) { if (pola) k_rt1_image_voro<true>(E.cv.M, E.A, E.cv.G); else k_rt1_image_voro<false>(E.cv.M, E.A, E.cv.G); }
  else if (m->l3D) { if (pola) k_rt1_image<true, true>(E.cv.M, E.A); else k_rt1_image<true, false>(E.cv.M, E.A); }
  else { if (pola) k_rt1_image<false, true>(E.cv.M, E.A); else k_rt1_image<false, false>(E.cv.M, E.A); }
  if (n_rays) *n_rays = (int)rays;
  return 0;
}

// ray tracing method 2 (the source function of inclination ibin as mcgpu_rt2_source leaves it): the SED sampling (image =
// null, out[N_type_flux]) or an image(npix_x, npix_y, N_type_flux)
extern "C" int emu_rt2_map(const oracle_model* m, const oracle_rt_opts* o, const float* eps2, const float* eps2_star, int nang_rt,
                           int nang_star, int ibin, const double* z_grid, const float* Tdust, int npix_x, int npix_y,
                           double map_size, double zoom, double* out, double* image, int* n_rays) {
  if (m->l3D || m->grid_type == 3) return 31;
  EmuRt E(m, o, nullptr, Tdust);
  const int ntf = m->N_type_flux;
  E.A.method2 = 1; E.A.q_only = ibin - 1; E.A.nang_rt = nang_rt; E.A.nang_star = nang_star;
  E.A.eps2 = eps2; E.A.eps2_star = eps2_star; E.A.z_grid = z_grid;
  const bool pola = ntf == 4 || ntf == 8;
  std::vector<double> all((size_t)E.A.nRT * ntf, 0.0);
  if (!image) {
    E.A.out = all.data();
    if (pola) k_rt1_dust_map<false, true>(E.cv.M, E.A); else k_rt1_dust_map<false, false>(E.cv.M, E.A);
    for (int t = 0; t < ntf; ++t) out[t] = all[(size_t)(ibin - 1) * ntf + t];
    return 0;
  }
  std::vector<double> img((size_t)E.A.nRT * ntf * npix_x * npix_y, 0.0);
  unsigned long long rays = 0;
  E.A.npix_x = npix_x; E.A.npix_y = npix_y;
  E.A.npix_x_max = o->l_sym_ima ? npix_x / 2 + npix_x % 2 : npix_x;
  E.A.taille_pix = (map_size / zoom) / (double)(npix_x > npix_y ? npix_x : npix_y);
  E.A.image = img.data(); E.A.n_rays = &rays;
  if (pola) k_rt1_image<false, true>(E.cv.M, E.A); else k_rt1_image<false, false>(E.cv.M, E.A);
  const int n_az = E.A.nRT / E.A.RT_n_incl;
  for (int t = 0; t < ntf; ++t)
    for (size_t p = 0; p < (size_t)npix_x * npix_y; ++p)
      image[(size_t)t * npix_x * npix_y + p] = img[(((size_t)t * n_az + 0) * E.A.RT_n_incl + (ibin - 1)) * npix_x * npix_y + p];
  if (n_rays) *n_rays = (int)rays;
  return 0;
}

// az_sector_certain (mc_device.hip.h) beside the reference's expression for the azimuthal sector of a point
// (cylindrical_grid.f90:1121-1126): certain[i] = the default-real decision claimed certainty, k_fast / k_ref the two sectors.
extern "C" int emu_az_sector(int n, int n_az, const double* x, const double* y, int* k_fast, int* k_ref, int* certain) {
  for (int i = 0; i < n; ++i) {
    int kf = 0;
    certain[i] = az_sector_certain(x[i], y[i], n_az, kf) ? 1 : 0;
    k_fast[i] = kf;
    const double phi = modulo_d(atan2(y[i], x[i]), 2 * PI);
    int kk = (int)floor(phi * (1.0 / (2.0 * PI)) * (double)(float)n_az) + 1;
    if (kk == n_az + 1) kk = n_az;
    k_ref[i] = kk;
  }
  return 0;
}
