// The thermal packet loop on a Voronoi grid as a POOL of packets per workgroup (round 5).
//
// Why.  One packet per lane through the whole state machine (thermal_body_voro, mc_voronoi.hip.h) runs the wave at 0.30
// lane utilisation on the 1e6-site disk (tests/devtools/voro_diag.py, profiles/r05_voro_diag.log): a crossing is a scan of
// the cell's neighbour list (Voronoi.f90:879-917) whose length differs from lane to lane -- 13 trips of four neighbours
// per crossing round for the longest list of the wave, 24 of 64 lanes busy in an average trip -- and a packet interacts
// every 1.6 crossings, so 37 of 64 lanes sit in the interaction code, itself split between scattering and absorption.
// No assignment of packets to lanes that is fixed for a packet's life can fill a wave here.
//
// What.  A workgroup owns R packet RECORDS in HBM/L2 (128 B each, never shared with another workgroup) and a handful of
// index QUEUES in LDS: FREE, INT (stopped packets: interaction + the start of the next flight) and one CROSS queue per
// CLASS of neighbour-list length (the scan's trip count).  A wave repeatedly picks the fullest queue, pops up to 64
// record indices, loads those records, runs ONE phase for all of them -- 64 crossings of cells whose lists have about
// the same length, or 64 interactions, or 64 emissions -- writes back what changed and pushes every index onto the queue
// of its packet's next phase.  The class of the next cell comes from a byte per (cell, neighbour) next to the inlined
// neighbour records (VoroGrid::nb_cls), so routing needs no look-ahead load.
// Every value is computed by the very functions of the one-packet-per-lane kernel (voro_cross_cell, interact,
// emit_packet, capteur) from the same counter-based random numbers (keyed by packet id and event number), so a packet's
// history does not depend on the schedule: the frozen parity tests hold packet for packet, and tests/emu runs this
// scheduler with one lane on the CPU.
//
// Memory model: records are written and read by waves of ONE workgroup, i.e. on one CU sharing its vector L1, so
// workgroup-scope fences (s_waitcnt vmcnt(0)) between a record's stores and the publication of its index suffice.
#pragma once
#include "mc_voronoi.hip.h"

#if MCGPU_VORO_DIAG == 5   // stage timing (tests/devtools/voro_pool_diag.py time): every stamp waits for the wave's memory operations first
#define VP_T(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = clock64(); vt[i] += t_ - vt0; vt0 = t_; } while (0)
#else
#define VP_T(i) do { } while (0)
#endif
#ifndef VP_BATCH
#define VP_BATCH 16   // neighbour records requested at once by a crossing pass (voro_cross_cell<BATCH>)
#endif
namespace mcgpu {

constexpr int VP_NC = 6;            // crossing classes (by the trips of four neighbours a scan takes)
constexpr int VP_NQ = VP_NC + 2;    // + FREE + INT
enum : int { VP_FREE = 0, VP_INT = 1, VP_CROSS0 = 2 };
constexpr int VP_MAX_LOG_REC = 12;  // <= 4096 records per workgroup: a ring entry is (4-bit lap tag | 12-bit index)

__host__ __device__ inline int vp_class_of(int count) {
  const int t = (count + 3) >> 2;
  return t <= 3 ? 0 : (t == 4 ? 1 : (t == 5 ? 2 : (t == 6 ? 3 : (t <= 8 ? 4 : 5))));
}

// a packet between two phases
struct alignas(16) PRec {
  double x, y, z, u, v, w, extr, S0;
  int icell, prev_cell, lambda, star_icell;
  unsigned int flags, pk_cross, p_lo, p_hi;   // flags: bit 0 flag_star, 1 flag_scatt, 2 flag_ism, bits 8-15 the cell's class
  double S1, S2, S3;
  unsigned int event, pad;
};
static_assert(sizeof(PRec) == 128, "one 128-byte record per packet");

struct PoolArgs {
  PRec* recs;          // [gridDim.x][1 << log_rec]
  int log_rec;         // records per workgroup = ring capacity
  int cache_log_ns;    // the deposit cache's slots (mc_device.hip.h: DepCache)
};

// what the emission phase reads: a copy of the launch's arguments in HBM (the phase is a function of its own, see vp_emit_phase)
struct VpBlob {
  DevModel M;
  RunArgs A;
  VoroGrid G;
};

struct VpCtl {
  unsigned int head[VP_NQ], tail[VP_NQ];
  int n_live;      // packets emitted and not finished (raised before their ids are taken from the global counter)
  int ids_done;    // the global work counter has run out
  int abort_flag;
  int beat;        // bumped by every pass: the idle waves' sign of life
  int pad[12];
};
static_assert(sizeof(VpCtl) == 128, "control block");

__host__ __device__ inline size_t vp_lds_bytes(int log_rec) { return sizeof(VpCtl) + ((size_t)VP_NQ << log_rec) * sizeof(unsigned short); }

__device__ inline int vp_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline unsigned int vp_ldu(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void vp_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline unsigned int vp_ld16(const unsigned short* p) {
  return (unsigned int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ inline void vp_st16(unsigned short* p, unsigned int v) {
  __hip_atomic_store(p, (unsigned short)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ inline unsigned int vp_entry(unsigned int at, int log_cap, int rid) { return ((((at >> log_cap) & 7u) + 1u) << 12) | (unsigned int)rid; }
__device__ inline int vp_count(const VpCtl* Q, int q) { return (int)(vp_ldu(&Q->tail[q]) - vp_ldu(&Q->head[q])); }

// Push the record of every lane with `want` onto its queue q (per lane).  A ring is never full: it has one slot per
// record.  The records must have been written before the call (the fence orders those stores in front of the entries).
__device__ inline void vp_push(VpCtl* Q, unsigned short* rings, int log_cap, int lane, bool want, int q, int rid) {
  unsigned long long todo = __ballot(want);
  if (todo == 0ull) return;
  __threadfence_block();
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int ql = __shfl(q, leader);
    const bool mine = want && q == ql;
    const unsigned long long m = __ballot(mine);
    const int n = __popcll(m), rank = __popcll(m & ((1ull << lane) - 1ull));
    unsigned int pos = 0;
    if (lane == leader) pos = atomicAdd(&Q->tail[ql], (unsigned int)n);
    pos = __shfl(pos, leader);
    if (mine) {
      const unsigned int at = pos + (unsigned int)rank;
      vp_st16(&rings[((size_t)ql << log_cap) + (at & ((1u << log_cap) - 1u))], vp_entry(at, log_cap, rid));
    }
    todo &= ~m;
  }
}

// Pop up to one index for each of the wave's first `wl` lanes from ring q; returns the index or -1.  Entries whose
// producer has reserved but not yet written them end the batch early (the lap tag tells).
__device__ inline int vp_pop(VpCtl* Q, const unsigned short* rings, int log_cap, int q, int lane, int wl) {
  for (int attempt = 0; attempt < 8; ++attempt) {
    const unsigned int h = vp_ldu(&Q->head[q]), t = vp_ldu(&Q->tail[q]);  // (same address in every lane)
    const int avail = (int)(t - h);
    if (avail <= 0) return -1;
    int c = wl < avail ? wl : avail;
    const bool mine = lane < c;
    const unsigned int at = h + (unsigned int)lane;
    unsigned int word = 0;
    if (mine) word = vp_ld16(&rings[((size_t)q << log_cap) + (at & ((1u << log_cap) - 1u))]);
    const bool ok = mine && (word >> 12) == (((at >> log_cap) & 7u) + 1u);
    const unsigned long long bad = __ballot(mine && !ok);
    if (bad) c = __ffsll((long long)bad) - 1;   // the entries in front of the first unpublished one
    if (c == 0) return -1;
    unsigned int old = 0;
    if (lane == 0) old = atomicCAS(&Q->head[q], h, h + (unsigned int)c);
    old = __shfl(old, 0);
    if (old == h) {
      __threadfence_block();
      return lane < c ? (int)(word & 0xFFFu) : -1;
    }
  }
  return -1;
}

// ---------------------------------------------------------------------------
// EMISSION (mc_photon_loop body, dust_transfer.f90:529-541) + the start of the first flight, as a function of its own
// (noinline): inlined next to the crossing it shares enough code with it (the walls of the box, the nearest-site
// searches) for the compiler to hoist that out of both -- the kernel then needs 229 registers instead of the 144 of its
// two hot phases.  The phase runs 0.017 times per packet, so the call and the scalar loads of the arguments from the
// blob cost nothing.  Returns VPE_* bits | the queue the record goes to.
// ---------------------------------------------------------------------------
enum : unsigned int { VPE_FIN = 0x100u, VPE_ESC = 0x200u, VPE_FLIGHT = 0x400u, VPE_ERR = 0x800u };
template <bool POLA>
__device__ __attribute__((noinline)) unsigned int vp_emit_phase(const VpBlob* blob, PRec* R, unsigned int id_lo, unsigned int id_hi) {
#ifndef MCGPU_LANE_EMULATION
  blob = reinterpret_cast<const VpBlob*>(((unsigned long long)__builtin_amdgcn_readfirstlane((int)((unsigned long long)blob >> 32)) << 32) |
                                         (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned long long)blob));
  extern __shared__ double lds_raw[];
#endif
  const DevModel& M = blob->M;
  const RunArgs& A = blob->A;
  const VoroGrid& G = blob->G;
  const Lds T = lds_carve(lds_raw, M);
  Rng rng;
  rng.init(A.seed, 0);
  rng.p_lo = id_lo; rng.p_hi = id_hi;
  float f[12];
  rng.emission_event(f);
  const int lambda = select_wl_em(T, M, f[0]);
  lds_count_sent(T, lambda);
  bool lintersect, flag_star, flag_ism;
  double x, y, z, u, v, w;
  int icell = 0;
  VoroEmitOps ops{G, M, icell};
  const int rc = emit_packet(M, f, lambda, T.fstar[lambda - 1], M.frac_E_disk[lambda - 1],
                             M.prob_E_cell ? M.prob_E_cell + (size_t)(M.n_cells + 1) * (lambda - 1) : nullptr,
                             ops, x, y, z, u, v, w, flag_star, flag_ism, lintersect);
  if (rc) { *A.err = rc; return VPE_FIN | VPE_ERR; }
  if (!lintersect) {   // never entered the grid
    if (flag_ism) return VPE_FIN;
    const double S[4] = {1.0, 0.0, 0.0, 0.0};
    capteur<POLA>(M, A.sed, lambda, u, v, w, S, flag_star, false);
    return VPE_FIN | VPE_ESC;
  }
  // the first flight (dust_transfer.f90:1208-1215, optical_depth.f90:68)
  const double extr = tau_of_draw(f[8]);
  const int i_star = intersect_stars(M, x, y, z, u, v, w);
  const int cls = icell > 0 ? vp_class_of(G.cell[icell - 1].count) : 0;
  R->x = x; R->y = y; R->z = z; R->u = u; R->v = v; R->w = w; R->extr = extr; R->S0 = 1.0;
  R->icell = icell; R->prev_cell = 0; R->lambda = lambda; R->star_icell = (i_star > 0) ? M.star_cell[4 * (i_star - 1)] : 0;
  R->flags = (flag_star ? 1u : 0u) | (flag_ism ? 4u : 0u) | ((unsigned int)cls << 8);
  R->pk_cross = 0u; R->p_lo = id_lo; R->p_hi = id_hi;
  R->S1 = 0.0; R->S2 = 0.0; R->S3 = 0.0; R->event = rng.event;
  return VPE_FLIGHT | (unsigned int)(VP_CROSS0 + cls);
}

// ---------------------------------------------------------------------------
// The kernel body.  blockDim.x lanes in waves of (at most) 64; the lane emulation runs it with one lane.
// ---------------------------------------------------------------------------
template <bool POLA>
__device__ __forceinline__ void thermal_body_voro_pool(const DevModel& M, const RunArgs& A, const VoroGrid& G, const PoolArgs& P,
                                                       const VpBlob* blob, double* lds_base) {
  const Lds T = lds_carve(lds_base, M);
  lds_stage(T, M);
  DepCache DC;
  DC.log_ns = P.cache_log_ns;
  DC.val = lds_base + (lds_bytes(M) + sizeof(double) - 1) / sizeof(double);
  DC.tag = reinterpret_cast<int*>(DC.val + ((size_t)1 << P.cache_log_ns));
  VpCtl* Q = reinterpret_cast<VpCtl*>(DC.tag + ((size_t)1 << P.cache_log_ns));
  unsigned short* rings = reinterpret_cast<unsigned short*>(Q + 1);
  const int log_cap = P.log_rec, n_rec = 1 << P.log_rec;
  for (int i = threadIdx.x; i < (1 << P.cache_log_ns); i += blockDim.x) { DC.val[i] = 0.0; DC.tag[i] = 0; }
  for (int i = threadIdx.x; i < (VP_NQ << log_cap); i += blockDim.x) rings[i] = (i < n_rec) ? (unsigned short)vp_entry((unsigned)i, log_cap, i) : (unsigned short)0;
  if (threadIdx.x == 0) {
    for (int q = 0; q < VP_NQ; ++q) { Q->head[q] = 0u; Q->tail[q] = 0u; }
    Q->tail[VP_FREE] = (unsigned int)n_rec;   // every record is free
    Q->n_live = 0; Q->ids_done = 0; Q->abort_flag = 0; Q->beat = 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wl = blockDim.x < 64 ? (int)blockDim.x : 64;   // lanes of a wave (1 in the lane emulation)
  PRec* const recs = P.recs + ((size_t)blockIdx.x << log_cap);
  Rng rng;
  rng.init(A.seed, 0);   // (k0, k1: the launch's key; the packet id and event number travel in the record)

  unsigned int c_cross = 0, c_flight = 0, c_scatt = 0, c_abs = 0, c_esc = 0, c_kill = 0, c_pack = 0;
  int idle_spins = 0, last_beat = 0;
  bool d_want = false;   // the last pass's records, not yet on their queues
  int d_q = 0, d_rid = 0;
#if MCGPU_VORO_DIAG == 5
  unsigned long long vt[10] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}, vt0 = clock64();
#endif
#if MCGPU_VORO_DIAG == 4   // (tests/devtools/voro_diag.py: passes and lanes per phase, idle rounds)
  VoroDiag VDg;
  for (int i = 0; i < 10; ++i) VDg.c[i] = 0u;
#endif

  for (unsigned int pass = 0;; ++pass) {
    VP_T(0);   // (whatever follows the last stamp of a pass: the loop's end, the cache fold)
    if (vp_ld(&Q->abort_flag)) break;
    // ---- which phase: an emission while the pool has a wave's worth of free records (keeps the population at R),
    // else the fullest queue
    const bool ids_left = vp_ld(&Q->ids_done) == 0;
    const int n_free = vp_count(Q, VP_FREE);
    int best_q = -1, best_n = 0;
#pragma unroll
    for (int q = VP_INT; q < VP_NQ; ++q) {
      const int n = vp_count(Q, q);
      if (n > best_n) { best_n = n; best_q = q; }
    }
    int q;
    if (ids_left && n_free >= wl) q = VP_FREE;
    else if (best_n >= wl) q = best_q;
    else if (ids_left && n_free > 0 && 2 * best_n < wl) q = VP_FREE;
    else if (best_n > 0) q = best_q;
    else {
      if (__ballot(d_want)) { vp_push(Q, rings, log_cap, lane, d_want, d_q, d_rid); d_want = false; continue; }  // (what this wave still holds back)
      if (!ids_left && vp_ld(&Q->n_live) == 0) break;   // every packet of the launch is finished
      { const int b = vp_ld(&Q->beat); if (b != last_beat) { last_beat = b; idle_spins = 0; } }
      // (a minute or more without a pass ANYWHERE in the workgroup -- the beat is bumped by every wave in every pass it
      // works, so a wave stalled behind a long memory or atomic queue, or serialised by a profiler's counter pass, does not
      // trip this --: a packet has been lost by a logic error; error 15 ends the launch instead of hanging the GPU)
      if (++idle_spins > (1 << 27)) { *A.err = 15; vp_st(&Q->abort_flag, 1); }
#if MCGPU_VORO_DIAG == 4
      if (lane == 0) VDg.c[7]++;
#endif
      __builtin_amdgcn_s_sleep(4);
      VP_T(9);   // idle
      continue;
    }
    VP_T(1);   // the choice
    const int rid = vp_pop(Q, rings, log_cap, q, lane, wl);
    const bool have = rid >= 0;
    VP_T(2);   // the pop
#if MCGPU_VORO_DIAG == 4
    if (__ballot(have) == 0ull) { if (lane == 0) VDg.c[8]++; }
    else if (have) { const int w_ = q == VP_FREE ? 5 : (q == VP_INT ? 3 : 1); VD(VDg, w_, w_ + 1); if (q >= VP_CROSS0 + 4) VDg.c[9]++; }
#endif
    PRec* const R = recs + (have ? rid : 0);
    // The records of the LAST pass are published only now: this pass's records are touched first (one load each, which
    // brings the record's line to the CU), and the fence in front of the publication -- a wait for the wave's memory
    // operations, counted in issue order -- then waits for that load and the last pass's stores together instead of for
    // the stores alone at the end of every pass.
#ifndef MCGPU_LANE_EMULATION
    if (have && q != VP_FREE) { const double pre = __hip_atomic_load(&R->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("" :: "v"(pre)); }
#endif
    if (__ballot(d_want)) { vp_push(Q, rings, log_cap, lane, d_want, d_q, d_rid); d_want = false; }
    VP_T(3);   // first touch of the records + the last pass's publication
    if (__ballot(have) == 0ull) continue;   // (another wave was faster)
    if (lane == 0) atomicAdd(&Q->beat, 1);
    bool push = false, fin = false;     // what becomes of this lane's record: onto queue next_q / finished (back to FREE)
    int next_q = VP_INT;

    if (q == VP_FREE) {
      // ================= EMISSION (mc_photon_loop body, dust_transfer.f90:529-541) + the first flight's start
      const unsigned long long m = __ballot(have);
      const int n = __popcll(m), rank = __popcll(m & ((1ull << lane) - 1ull));
      unsigned long long base = 0;
      if (lane == 0) {
        atomicAdd(&Q->n_live, n);
        base = atomicAdd(A.next_packet, (unsigned long long)n);
        if (base + (unsigned long long)n >= A.n_packets) vp_st(&Q->ids_done, 1);
        const unsigned long long got = base >= A.n_packets ? 0ull : (A.n_packets - base < (unsigned long long)n ? A.n_packets - base : (unsigned long long)n);
        if (got < (unsigned long long)n) atomicAdd(&Q->n_live, -(int)((unsigned long long)n - got));
      }
      base = __shfl(base, 0);
      const unsigned long long my = base + (unsigned long long)rank;
      const bool served = have && my < A.n_packets;
      if (have && !served) { push = true; next_q = VP_FREE; }   // (the record goes back unused)
      if (served) {
        c_pack++;
        const unsigned long long id = A.first_packet + my;
        const unsigned int r = vp_emit_phase<POLA>(blob, R, (unsigned int)id, (unsigned int)(id >> 32));
        if (r & VPE_ERR) vp_st(&Q->abort_flag, 1);
        if (r & VPE_ESC) c_esc++;
        if (r & VPE_FLIGHT) c_flight++;
        if (r & VPE_FIN) fin = true;
        else { push = true; next_q = (int)(r & 0xFFu); }
      }
    } else if (q == VP_INT) {
      // ================= INTERACTION (dust_transfer.f90:1260-1402) + the next flight's start
      if (have) {
        rng.p_lo = R->p_lo; rng.p_hi = R->p_hi; rng.event = R->event;
        const double x = R->x, y = R->y, z = R->z;
        const double u = R->u, v = R->v, w = R->w;
        double S[4] = {R->S0, POLA ? R->S1 : 0.0, POLA ? R->S2 : 0.0, POLA ? R->S3 : 0.0};
        int lambda = R->lambda;
        const int icell = R->icell;
        const unsigned int fl = R->flags;
        bool flag_star = (fl & 1u) != 0, flag_scatt = (fl & 2u) != 0, flag_ism = (fl & 4u) != 0;
        float g[8];
        rng.interaction_event(g, M.m1 != 0);
        double u1, v1, w1;
        const int ic = icell - 1;
        interact<POLA>(T, M, g, lambda, u, v, w, u1, v1, w1, S, flag_star, flag_scatt, c_scatt, c_abs, [&]() {
          if (A.frozen) return A.E_prior[ic];
          double E = MCGPU_DIAG(A.flags, 8) ? A.E_abs[ic] : __hip_atomic_load(&A.E_abs[ic], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          E += DC.pending(ic + 1) * (double)gridDim.x;
          return E * A.qscale;
        }, M.volume + ic);
        if (!flag_scatt) flag_ism = false;
        const double extr = tau_of_draw(g[5]);
        const int i_star = intersect_stars(M, x, y, z, u1, v1, w1);
        c_flight++;
        R->u = u1; R->v = v1; R->w = w1; R->extr = extr; R->S0 = S[0];
        if (POLA) { R->S1 = S[1]; R->S2 = S[2]; R->S3 = S[3]; }
        R->prev_cell = 0; R->lambda = lambda; R->star_icell = (i_star > 0) ? M.star_cell[4 * (i_star - 1)] : 0;
        R->flags = (fl & 0xFF00u) | (flag_star ? 1u : 0u) | (flag_scatt ? 2u : 0u) | (flag_ism ? 4u : 0u);
        R->event = rng.event;
        push = true; next_q = VP_CROSS0 + (int)((fl >> 8) & 0xFFu);
      }
      VP_T(4);   // an interaction pass
    } else {
      // ================= ONE CELL CROSSING (physical_length's loop body, optical_depth.f90:77-178)
      bool exited = false;
      double x = 0, y = 0, z = 0, u = 0, v = 0, w = 1;
      int lambda = 1;
      unsigned int fl = 0u;
      if (have) {
        x = R->x; y = R->y; z = R->z; u = R->u; v = R->v; w = R->w;
        const double extr = R->extr, S0 = R->S0;
        const int icell = R->icell, prev_cell = R->prev_cell, star_icell = R->star_icell;
        lambda = R->lambda;
        fl = R->flags;
        unsigned int pk_cross = R->pk_cross;
        if (icell < 0) {  // test_exit_grid_Voronoi (:1446)
          exited = true;
        } else if (star_icell > 0 && icell == star_icell) {  // optical_depth.f90:91-97
          c_kill++;
          fin = true;
        } else {
          VP_T(5);   // the record
          const VoroCell C = G.cell[icell - 1];
          VP_T(6);   // the cell
          const double opacity = T.kappa[lambda - 1] * C.kf, kabs_c = T.kabs[lambda - 1];
          double x1, y1, z1, l, l_contrib, l_void;
          int next, cls_scan;
          voro_cross_cell<VP_BATCH>(G, M, C, x, y, z, u, v, w, icell, prev_cell, x1, y1, z1, next, l, l_contrib, l_void, &cls_scan);
          c_cross++;
          VP_T(7);   // the crossing (scan included)
          const double tau = l_contrib * opacity;
          if (tau > extr) {
            const double lc = l_contrib * (extr / tau);
            const double ls = l_void + lc;
            const double dE = kabs_c * lc * S0;
            if (dE != 0.0 && !MCGPU_DIAG(A.flags, 1)) {
              if (!DC.add(icell, dE) && !MCGPU_DIAG(A.flags, 4)) atomic_add_f64(&A.E_abs[icell - 1], dE);
            }
            R->x = nd_add(x, nd_mul(ls, u));
            R->y = nd_add(y, nd_mul(ls, v));
            R->z = nd_add(z, nd_mul(ls, w));
            R->flags = (fl & 0xFFu) | ((unsigned int)vp_class_of(C.count) << 8);
            push = true; next_q = VP_INT;
          } else {
            const double dE = kabs_c * l_contrib * S0;
            if (dE != 0.0 && !MCGPU_DIAG(A.flags, 1)) {
              if (!DC.add(icell, dE) && !MCGPU_DIAG(A.flags, 4)) atomic_add_f64(&A.E_abs[icell - 1], dE);
            }
            // the class of the cell entered: the byte next to the neighbour's record, or -- where the next cell did not
            // come out of the scan (a star's cell, the recovery search) -- its own record
            int cls = 0;
            if (next > 0) cls = (cls_scan >= 0) ? cls_scan : vp_class_of(G.cell[next - 1].count);
            R->x = x1; R->y = y1; R->z = z1; R->extr = extr - tau;
            R->icell = next; R->prev_cell = icell;
            R->flags = (fl & 0xFFu) | ((unsigned int)cls << 8);
            x = x1; y = y1; z = z1;
            if (next < 0) exited = true;
            else if (star_icell > 0 && next == star_icell) { c_kill++; fin = true; }
            else { push = true; next_q = VP_CROSS0 + cls; }
          }
          if (++pk_cross > 200000000u) {  // a packet that never leaves: flag it, drop it
            *A.err = 13;
            push = false; exited = false; fin = true;
          }
          R->pk_cross = pk_cross;
        }
      }
      VP_T(8);   // stop / pass, deposit, the record's stores
      if (exited) {   // capteur (output.f90:294): the packet has left the grid
        if (!(fl & 4u)) {
          const double S[4] = {R->S0, POLA ? R->S1 : 0.0, POLA ? R->S2 : 0.0, POLA ? R->S3 : 0.0};
          capteur<POLA>(M, A.sed, lambda, u, v, w, S, (fl & 1u) != 0, (fl & 2u) != 0);
          c_esc++;
        }
        fin = true;
      }
    }

    // ---- where the records go
    {
      const unsigned long long mf = __ballot(fin);
      if (mf && lane == (__ffsll((long long)mf) - 1)) atomicAdd(&Q->n_live, -__popcll(mf));
      d_want = push || fin; d_q = fin ? VP_FREE : next_q; d_rid = rid;   // (published at the start of the next pass)
    }

    // barrier-free partial fold of the deposit cache (see thermal_body_voro)
    if (((pass + 1) % (unsigned int)A.flush_every) == 0) {
      const int n_waves = (blockDim.x + 63) >> 6, wave = threadIdx.x >> 6;
      const int slice = (wave + (int)((pass + 1) / (unsigned int)A.flush_every)) % n_waves;
      const int ns = 1 << P.cache_log_ns, per = (ns + n_waves - 1) / n_waves;
      const int i0 = slice * per, i1 = (i0 + per < ns) ? i0 + per : ns;
      for (int i = i0 + lane; i < i1; i += 64) {
        const int t = DC.tag[i];
        if (t == 0) continue;
        const unsigned long long bits = atomicExch(reinterpret_cast<unsigned long long*>(&DC.val[i]), 0ull);
        const double e = __longlong_as_double((long long)bits);
        if (e != 0.0) atomic_add_f64(&A.E_abs[t - 1], e);
      }
    }
  }

  __syncthreads();  // every wave of the workgroup is done emitting and depositing
  lds_flush_sent(T, M, A.n_sent);
  for (int i = threadIdx.x; i < (1 << P.cache_log_ns); i += blockDim.x) {  // final fold
    const double e = DC.val[i];
    if (e != 0.0) atomic_add_f64(&A.E_abs[DC.tag[i] - 1], e);
  }
  unsigned int cs[8] = {c_pack, c_cross, c_flight, c_scatt, c_abs, c_esc, c_kill, 0u};
#if MCGPU_VORO_DIAG == 5
  if (lane == 0) for (int q = 0; q < 10; ++q) atomicAdd(&A.counters[q], vt[q]);
  for (int q = 0; q < 8; ++q) cs[q] = 0u;
#endif
#if MCGPU_VORO_DIAG == 4
  for (int q = 1; q < 8; ++q) cs[q] = VDg.c[q];
  {
    unsigned long long v8 = VDg.c[8], v9 = VDg.c[9];
    for (int off = 32; off > 0; off >>= 1) { v8 += __shfl_down(v8, off); v9 += __shfl_down(v9, off); }
    if (lane == 0 && v8) atomicAdd(&A.counters[8], v8);
    if (lane == 0 && v9) atomicAdd(&A.counters[9], v9);
  }
#endif
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    unsigned long long vsum = cs[q];
    for (int off = 32; off > 0; off >>= 1) vsum += __shfl_down(vsum, off);
    if (lane == 0 && vsum) atomicAdd(&A.counters[q], vsum);
  }
}

template <bool POLA, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_thermal_voro_pool(const DevModel M, const RunArgs A, const VoroGrid G, const PoolArgs P,
                                                             const VpBlob* blob) {
  extern __shared__ double lds_raw[];
  thermal_body_voro_pool<POLA>(M, A, G, P, blob, lds_raw);
}

}  // namespace mcgpu
