// The default schedule of the thermal packet loop on cylindrical grids (MCGPU_ROLES, see mcgpu.hip):
// the waves of a workgroup take ROLES and pass packets to each other through queues in LDS.
//
// Why: the flight lengths are heavy-tailed (most flights of a packet random-walking in the thick
// inner disk are 1-2 cell crossings, a few are 100+, and those few hold 94 % of all crossings).
// With one packet per lane and every lane doing everything (thermal_body), a wavefront leaves the
// crossing loop to serve the short flights again and again while its long flights idle: measured
// lane utilisation of the crossing loop 54 %.
//
//   SERVER waves  emit packets, run the interactions and the first k_short crossings of every
//                 flight; a packet still in flight after those is a long flight: it is pushed
//                 to the FLY queue and the lane takes over a packet that waits for its
//                 interaction (SRV queue) or emits a new one.
//   FLYER waves   only cross cells: every R crossings the lanes whose packet stopped for an
//                 interaction push it to the SRV queue and the empty lanes pop the FLY queue;
//                 packets that leave the grid are binned on the spot.
//
// The queues are two record pools in LDS (structure of arrays) with index stacks, guarded by one
// workgroup spin lock taken by lane 0 of a wave for a few LDS operations; a wave exchanges all
// its pushes and pops in one step.  Every spin is bounded (the lock: 2^22 tries; a wave without work:
// 2^26 polls, i.e. minutes): on overflow the workgroup aborts with error 14 / 15 instead of hanging.  Results do not depend on who runs a packet (counter-based random
// numbers keyed by the packet id), so this schedule reproduces thermal_body packet for packet.
#pragma once
#include "mc_device.hip.h"

#ifdef MCGPU_COUNT_ITERS  // diagnostic build (tools/roles_check.py diag): statements that only count
#define RQ_DIAG(...) __VA_ARGS__
#else
#define RQ_DIAG(...)
#endif

namespace mcgpu {

constexpr int RQ_NF = 192;  // records of packets ready for a long flight
constexpr int RQ_NS = 192;  // records of packets waiting for their interaction
constexpr int RQ_N = RQ_NF + RQ_NS;

template <bool POLA>
struct RoleQ {
  int lock, n_pending, ids_done, abort_flag;
  int fly_top, fly_free_top, srv_top, srv_free_top;
  short fly_stack[RQ_NF], fly_free[RQ_NF], srv_stack[RQ_NS], srv_free[RQ_NS];
  double x[RQ_N], y[RQ_N], z[RQ_N], u[RQ_N], v[RQ_N], w[RQ_N], extr[RQ_N];
  double S[POLA ? 4 : 1][RQ_N];
  int ri[RQ_N], zj[RQ_N], k[RQ_N], lambda[RQ_N], star_key[RQ_N], p_lo[RQ_N], p_hi[RQ_N], event[RQ_N], flags[RQ_N];
  unsigned int pk_cross[RQ_N];
  float tau_rand[RQ_N];
};

struct PkState {
  double x, y, z, u, v, w, extr;
  double S[4];
  int ri, zj, k, lambda, star_key;
  Rng rng;
  bool flag_star, flag_scatt, flag_ism;
  int st;
  float tau_rand;
  unsigned int pk_cross;
};

template <bool POLA>
__device__ inline void rq_store(RoleQ<POLA>* Q, int id, const PkState& p) {
  Q->x[id] = p.x; Q->y[id] = p.y; Q->z[id] = p.z; Q->u[id] = p.u; Q->v[id] = p.v; Q->w[id] = p.w;
  Q->extr[id] = p.extr;
  Q->S[0][id] = p.S[0];
  if (POLA) { Q->S[POLA ? 1 : 0][id] = p.S[1]; Q->S[POLA ? 2 : 0][id] = p.S[2]; Q->S[POLA ? 3 : 0][id] = p.S[3]; }
  Q->ri[id] = p.ri; Q->zj[id] = p.zj; Q->k[id] = p.k; Q->lambda[id] = p.lambda; Q->star_key[id] = p.star_key;
  Q->p_lo[id] = (int)p.rng.p_lo; Q->p_hi[id] = (int)p.rng.p_hi; Q->event[id] = (int)p.rng.event;
  Q->flags[id] = p.st | (p.flag_star ? ST_STAR : 0) | (p.flag_scatt ? ST_SCATT : 0) | (p.flag_ism ? ST_ISM : 0);
  Q->pk_cross[id] = p.pk_cross;
  Q->tau_rand[id] = p.tau_rand;
}

template <bool POLA>
__device__ inline void rq_load(const RoleQ<POLA>* Q, int id, PkState& p) {
  p.x = Q->x[id]; p.y = Q->y[id]; p.z = Q->z[id]; p.u = Q->u[id]; p.v = Q->v[id]; p.w = Q->w[id];
  p.extr = Q->extr[id];
  p.S[0] = Q->S[0][id];
  if (POLA) { p.S[1] = Q->S[POLA ? 1 : 0][id]; p.S[2] = Q->S[POLA ? 2 : 0][id]; p.S[3] = Q->S[POLA ? 3 : 0][id]; }
  p.ri = Q->ri[id]; p.zj = Q->zj[id]; p.k = Q->k[id]; p.lambda = Q->lambda[id]; p.star_key = Q->star_key[id];
  p.rng.p_lo = (uint32_t)Q->p_lo[id]; p.rng.p_hi = (uint32_t)Q->p_hi[id]; p.rng.event = (uint32_t)Q->event[id];
  const int f = Q->flags[id];
  p.st = f & ST_MASK; p.flag_star = (f & ST_STAR) != 0; p.flag_scatt = (f & ST_SCATT) != 0; p.flag_ism = (f & ST_ISM) != 0;
  p.pk_cross = Q->pk_cross[id];
  p.tau_rand = Q->tau_rand[id];
}

__device__ inline int rq_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void rq_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// lane 0 of the calling wave takes / releases the workgroup lock; false: gave up (abort)
template <bool POLA>
__device__ inline bool rq_lock(RoleQ<POLA>* Q, int* err) {
  int spins = 0;
  while (atomicCAS(&Q->lock, 0, 1) != 0) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1 << 22) || rq_ld(&Q->abort_flag)) {
      rq_st(&Q->abort_flag, 1);
      *err = 14;
      return false;
    }
  }
  __threadfence_block();
  return true;
}
template <bool POLA>
__device__ inline void rq_unlock(RoleQ<POLA>* Q) {
  __threadfence_block();
  rq_st(&Q->lock, 0);
}

// One wave-wide exchange with the queues.  SERVER: pushes go to the FLY queue, pops come from the
// SRV queue; flyers the other way round.  Lanes with want_push own a packet to hand over, lanes
// with want_pop are empty; a lane that manages to push is empty afterwards and pops in the same
// step.  On return `pushed` / `popped` say what happened to this lane (popped: p holds the new
// packet).
template <bool POLA, bool SERVER>
__device__ inline void rq_exchange(RoleQ<POLA>* Q, int lane, bool want_push, bool want_pop, PkState& p,
                                   bool& pushed, bool& popped, int* err) {
  pushed = false; popped = false;
  const unsigned long long m_push = __ballot(want_push);
  if ((m_push | __ballot(want_pop)) == 0ull) return;
  short* push_stack = SERVER ? Q->fly_stack : Q->srv_stack;
  short* push_free = SERVER ? Q->fly_free : Q->srv_free;
  int* push_top = SERVER ? &Q->fly_top : &Q->srv_top;
  int* push_free_top = SERVER ? &Q->fly_free_top : &Q->srv_free_top;
  short* pop_stack = SERVER ? Q->srv_stack : Q->fly_stack;
  short* pop_free = SERVER ? Q->srv_free : Q->fly_free;
  int* pop_top = SERVER ? &Q->srv_top : &Q->fly_top;
  int* pop_free_top = SERVER ? &Q->srv_free_top : &Q->fly_free_top;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int rank_push = __popcll(m_push & lt);
  // look before locking: an idle wave must not fight for the lock when there is nothing to move
  {
    const bool can_push = (m_push != 0ull) && rq_ld(push_free_top) > 0;
    const bool can_pop = rq_ld(pop_top) > 0;  // (a lane that pushes also wants to pop)
    if (!can_push && !can_pop) return;
  }

  // ---- reserve: free records for the pushes, queued records for the pops -------------------
  int ok = 1, n_push = 0, base_free = 0;
  if (lane == 0) {
    ok = rq_lock(Q, err) ? 1 : 0;
    if (ok) {
      const int ft = rq_ld(push_free_top);
      n_push = __popcll(m_push) < ft ? __popcll(m_push) : ft;
      base_free = ft - n_push;
      rq_st(push_free_top, base_free);
    }
  }
  ok = __shfl(ok, 0);
  if (!ok) return;
  n_push = __shfl(n_push, 0);
  base_free = __shfl(base_free, 0);
  const bool do_push = want_push && rank_push < n_push;
  const bool wants = want_pop || do_push;
  const unsigned long long m_pop = __ballot(wants);
  const int rank_pop = __popcll(m_pop & lt);
  int n_pop = 0, base_pop = 0;
  if (lane == 0) {
    const int pt = rq_ld(pop_top);
    n_pop = __popcll(m_pop) < pt ? __popcll(m_pop) : pt;
    base_pop = pt - n_pop;
    rq_st(pop_top, base_pop);
  }
  n_pop = __shfl(n_pop, 0);
  base_pop = __shfl(base_pop, 0);
  const bool do_pop = wants && rank_pop < n_pop;
  int id_push = -1, id_pop = -1;
  if (do_push) id_push = ((volatile short*)push_free)[base_free + rank_push];
  if (do_pop) id_pop = ((volatile short*)pop_stack)[base_pop + rank_pop];
  if (lane == 0) rq_unlock(Q);  // (the fence inside orders the index reads above before the release)

  // ---- move the packets ------------------------------------------------------------------
  if (do_push) { rq_store(Q, id_push, p); pushed = true; p.st = S_EMIT; }
  if (do_pop) { rq_load(Q, id_pop, p); popped = true; }
  __threadfence_block();

  // ---- publish the pushed records, return the popped ones to their free list ----------------
  if (n_push > 0 || n_pop > 0) {
    int tp = 0, fp = 0;
    if (lane == 0) {
      ok = rq_lock(Q, err) ? 1 : 0;
      if (ok) {
        tp = rq_ld(push_top);
        fp = rq_ld(pop_free_top);
      }
    }
    ok = __shfl(ok, 0);
    if (!ok) return;
    tp = __shfl(tp, 0);
    fp = __shfl(fp, 0);
    if (do_push) ((volatile short*)push_stack)[tp + rank_push] = (short)id_push;
    if (do_pop) ((volatile short*)pop_free)[fp + rank_pop] = (short)id_pop;
    if (lane == 0) {
      __threadfence_block();
      rq_st(push_top, tp + n_push);
      rq_st(pop_free_top, fp + n_pop);
      rq_unlock(Q);
    }
  }
}

// One cell crossing of a packet in flight (physical_length's loop body, optical_depth.f90:77-178).  Returns the
// number of packets this lane finished (0 or 1).
// DARK: the reference tests l_dark_zone(icell0) at the top of the NEXT loop turn and then puts the packet back at
// the point where it entered the cell it has just crossed, direction reversed (:104-112).  The same thing is done
// here at the end of the crossing that leads into the dark cell -- the entry point is still at hand, so the packet
// needs no memory of it (a flight never starts inside a dark cell: packets are mirrored at its edge and the dark
// cells emit nothing, thermal_emission.f90:1817).
template <bool L3D, bool POLA, bool DARK, bool LDSE>
__device__ inline int roles_cross(const Lds& T, const DevModel& M, const RunArgs& A, double* E_lds, PkState& p,
                                  double inv_a, double inv_w, double kap, double kab, double& kf,
                                  unsigned int& c_cross, unsigned int& c_kill, unsigned int& c_dark) {
  const int n_rad = M.n_rad, nz = M.nz;
  const int azj = p.zj < 0 ? -p.zj : p.zj;
  const bool out = (p.ri == n_rad + 1) || ((azj == nz + 1) && (fabs(p.z) > M.zmaxmax));
  bool killed = false;
  if (p.star_key >= 0) {
    const int key = p.ri + (n_rad + 2) * ((p.zj + nz + 1) + (2 * nz + 3) * (p.k - 1));
    killed = (key == p.star_key);
  }
  if (out) { p.st = S_EXITED; return 0; }
  if (killed) { c_kill++; p.st = S_EMIT; return 1; }
  const bool real_cell = is_real_cell<L3D>(n_rad, nz, p.ri, p.zj);
  const int ic = real_cell ? cell_index<L3D>(n_rad, nz, p.ri, p.zj, p.k) : 0;
  double x1, y1, z1, l;
  int ri1, zj1, k1;
  MCGPU_CROSS<L3D>(T, M, p.x, p.y, p.z, p.u, p.v, p.w, inv_a, inv_w, p.ri, p.zj, p.k, x1, y1, z1, ri1, zj1, k1, l);
  c_cross++;
  if (++p.pk_cross > 200000000u) { *A.err = 13; p.st = S_EMIT; return 1; }
  // (kf was loaded at the end of the previous crossing: first used here, behind the geometry, so that the
  // latency of that load is covered by it)
  const double opacity = real_cell ? kap * kf : 0.0;
  const double tau = l * opacity;
  if (tau > p.extr) {
    const double lc = l * (p.extr / tau);
    if (real_cell && !(A.flags & 1)) deposit<LDSE>(A.E_abs, E_lds, ic, kab * lc * p.S[0]);
    p.x = p.x + lc * p.u;
    p.y = p.y + lc * p.v;
    p.z = p.z + lc * p.w;
    if (L3D) index_cell<L3D>(T, M, p.x, p.y, p.z, p.ri, p.zj, p.k);
    p.st = S_INTERACT;
  } else {
    p.extr = p.extr - tau;
    if (real_cell && !(A.flags & 1)) deposit<LDSE>(A.E_abs, E_lds, ic, kab * l * p.S[0]);
    const bool next_real = is_real_cell<L3D>(n_rad, nz, ri1, zj1);
    const int ic1 = next_real ? cell_index<L3D>(n_rad, nz, ri1, zj1, k1) : 0;
    if (DARK && next_real && M.dark[ic1]) {
      p.u = -p.u; p.v = -p.v; p.w = -p.w;  // back at the entry point of this cell, an interaction follows there
      c_dark++;
      p.st = S_INTERACT;
    } else {
      p.x = x1; p.y = y1; p.z = z1;
      p.ri = ri1; p.zj = zj1; p.k = k1;
      kf = next_real ? M.kappa_factor[ic1] : 0.0;
    }
  }
  return 0;
}

template <bool L3D, bool POLA, bool DARK, bool LDSE>
__device__ __forceinline__ void roles_body(const DevModel& M, const RunArgs& A, double* lds_base, int n_flyers,
                                           int k_short, int fly_iters, int fly_idle, int emit_qmax, int emit_min) {
  double* const E_lds = lds_base;
  const Lds T = lds_carve(lds_base + (LDSE ? M.n_cells : 0), M);
  lds_stage(T, M);
  RoleQ<POLA>* Q = reinterpret_cast<RoleQ<POLA>*>(lds_base + (LDSE ? M.n_cells : 0) + (lds_bytes(M) + 7) / 8);
  if (LDSE)
    for (int i = threadIdx.x; i < M.n_cells; i += blockDim.x) E_lds[i] = 0.0;
  for (int i = threadIdx.x; i < RQ_NF; i += blockDim.x) Q->fly_free[i] = (short)i;
  for (int i = threadIdx.x; i < RQ_NS; i += blockDim.x) Q->srv_free[i] = (short)(RQ_NF + i);
  if (threadIdx.x == 0) {
    Q->lock = 0; Q->n_pending = 0; Q->ids_done = 0; Q->abort_flag = 0;
    Q->fly_top = 0; Q->fly_free_top = RQ_NF; Q->srv_top = 0; Q->srv_free_top = RQ_NS;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_rad = M.n_rad, nz = M.nz;
  bool flyer = wave < n_flyers;  // n_flyers >= 100: every wave picks its role anew in each round
  const bool auto_roles = n_flyers >= 100;
  const int fly_fill = auto_roles ? n_flyers - 100 : 0;  // policy A: become a flyer when so many lanes can fly
  PkState p;
  p.x = p.y = p.z = p.u = p.v = 0.0; p.w = 1.0; p.extr = 0.0;
  p.S[0] = 1.0; p.S[1] = p.S[2] = p.S[3] = 0.0;
  p.ri = 0; p.zj = 1; p.k = 1; p.lambda = 1; p.star_key = -1;
  p.rng.init(A.seed, 0);
  p.flag_star = p.flag_scatt = p.flag_ism = false;
  p.st = S_EMIT;  // S_EMIT = the lane holds no packet
  p.tau_rand = 0.0f; p.pk_cross = 0;
  double inv_a = 0.0, inv_w = 0.0, kf = 0.0;
  double kap = 0.0, kab = 0.0;  // kappa(lambda), kappa_abs(lambda) of the packet in this lane: constants of a flight
  unsigned int c_cross = 0, c_flight = 0, c_scatt = 0, c_abs = 0, c_esc = 0, c_kill = 0, c_pack = 0, c_dark = 0;
  unsigned long long pk_next = 0, pk_end = 0;
  bool no_more_ids = false;  // wave-uniform: the global id counter is exhausted
  int idle_spins = 0;        // consecutive rounds without work: bounded, a lost packet must not hang the GPU
  RQ_DIAG(unsigned int d_fly_iters = 0, d_srv_iters = 0, d_fly_cross = 0, d_srv_rounds = 0, d_fly_rounds = 0;)
  RQ_DIAG(unsigned int d_in_flight = 0, d_handed = 0, d_popped = 0, d_empty = 0;)

  // diagnostics (A.flags bit 1): wall-clock (100 MHz) of this wave's start, of the moment the packet ids ran out
  // and of its end, summed over the waves into counters[10..14] (tools/wave_timeline.py)
  const unsigned long long t_start = (A.flags & 2) ? wall_clock64() : 0ull;
  unsigned long long t_ids_out = 0;
  for (int ep = 0;; ++ep) {
    if (rq_ld(&Q->abort_flag)) break;
    if ((A.flags & 2) && no_more_ids && !t_ids_out) t_ids_out = wall_clock64();
    int finished = 0;  // packets this lane finished in this round
    if (auto_roles) {
      // fly when the lanes can be (nearly) filled with packets in flight -- the wave's own plus the queue's --
      // and the packets that wait here for their interaction can be handed over; serve otherwise
      const int nF = __popcll(__ballot(p.st == S_FLIGHT)), nI = __popcll(__ballot(p.st == S_INTERACT));
      const int ft = rq_ld(&Q->fly_top), sfree = rq_ld(&Q->srv_free_top);
      if (n_flyers < 200) {  // policy A: fly when at least fly_fill lanes can fly
        const int room = 64 - nF;
        flyer = (nF + (ft < room ? ft : room) >= fly_fill) && (sfree >= nI);
      } else {  // policy B (default): take the role in which more of the 64 lanes have work in this round
        const int stq = rq_ld(&Q->srv_top), ffree = rq_ld(&Q->fly_free_top);
        const int keepI = nI - (nI < sfree ? nI : sfree);          // waiting packets a flyer could not hand over
        const int roomF = 64 - nF - keepI;
        const int fly_pot = nF + (ft < roomF ? ft : roomF);
        const int keepF = nF - (nF < ffree ? nF : ffree);          // flights a server could not hand over
        const int roomS = 64 - nI - keepF;
        const bool can_emit = !no_more_ids && (ft + stq) <= emit_qmax;  // (new packets fill the rest)
        const int srv_pot = nI + (can_emit ? roomS : (stq < roomS ? stq : roomS));
        flyer = fly_pot >= srv_pot;
      }
      // the tail: flights left in the queue when there is nothing to serve or emit any more must still be flown
      if (!flyer && ft > 0 && nI == 0 && no_more_ids && rq_ld(&Q->srv_top) == 0) flyer = true;
    }

    if (flyer) {
      // ---- FLYER: hand over the packets that stopped, take long flights from the queue ------
      bool pushed, popped;
      rq_exchange<POLA, false>(Q, lane, p.st == S_INTERACT, p.st == S_EMIT, p, pushed, popped, A.err);
      if (popped) {  // per-flight constants (cylindrical_grid.f90:941-952) and the cell's opacity factor
        const double a = p.u * p.u + p.v * p.v;
        inv_a = (a > TINY_REAL) ? 1.0 / a : HUGE_REAL;
        inv_w = (fabs(p.w) > TINY_REAL) ? 1.0 / p.w : copysign(HUGE_DP, p.w);
        kf = is_real_cell<L3D>(n_rad, nz, p.ri, p.zj) ? M.kappa_factor[cell_index<L3D>(n_rad, nz, p.ri, p.zj, p.k)] : 0.0;
        kap = T.kappa[p.lambda - 1]; kab = T.kabs[p.lambda - 1];
      }
      if (p.st == S_EXITED) {  // (left over from a round as a server)
        if (!p.flag_ism) { capteur<POLA>(M, A.sed, p.lambda, p.u, p.v, p.w, p.S, p.flag_star, p.flag_scatt); c_esc++; }
        p.st = S_EMIT;
        finished++;
      }
      const bool nothing_to_fly = __ballot(p.st == S_FLIGHT) == 0ull;
      if (nothing_to_fly) {
        if (__ballot(finished > 0) == 0ull && __ballot(p.st == S_INTERACT) == 0ull && rq_ld(&Q->ids_done) &&
            rq_ld(&Q->n_pending) == 0)
          break;  // (every operand is wave-uniform)
        __builtin_amdgcn_s_sleep(32);  // nothing to fly: wait for the servers
        if (++idle_spins > (1 << 26)) { *A.err = 15; rq_st(&Q->abort_flag, 1); }  // (minutes: a lost packet, not a long tail)
      } else {
      idle_spins = 0;
      RQ_DIAG(if (lane == 0) d_fly_rounds++;)
#pragma unroll 1
      for (int it = 0; it < fly_iters; ++it) {
        // back to the queues as soon as enough lanes have nothing to fly (or after fly_iters crossings)
        if (it > 0 && __popcll(__ballot(p.st != S_FLIGHT)) >= fly_idle) break;
        RQ_DIAG(if (lane == 0) d_fly_iters++; if (p.st == S_FLIGHT) d_fly_cross++;)
        if (p.st == S_FLIGHT) finished += roles_cross<L3D, POLA, DARK, LDSE>(T, M, A, E_lds, p, inv_a, inv_w, kap, kab, kf, c_cross, c_kill, c_dark);
        if (p.st == S_EXITED) {  // binned on the spot (capteur)
          if (!p.flag_ism) { capteur<POLA>(M, A.sed, p.lambda, p.u, p.v, p.w, p.S, p.flag_star, p.flag_scatt); c_esc++; }
          p.st = S_EMIT;
          finished++;
        }
      }
      }  // something to fly
    } else {
      // ---- SERVER ------------------------------------------------------------------------------
      RQ_DIAG(if (lane == 0) d_srv_rounds++;)
      // long flights go to the flyers, empty lanes take packets that wait for their interaction
      bool pushed, popped;
      RQ_DIAG(if (p.st == S_FLIGHT) d_in_flight++;)  // lanes that come into the server round with a flight
      rq_exchange<POLA, true>(Q, lane, p.st == S_FLIGHT && n_flyers > 0, p.st == S_EMIT, p, pushed, popped, A.err);
      RQ_DIAG(if (pushed) d_handed++; if (popped) d_popped++; if (p.st == S_EMIT) d_empty++;)

      // EMIT: lanes that are still empty start new packets (mc_photon_loop body, dust_transfer.f90:529-541)
      {
        const bool need = (p.st == S_EMIT);
        const unsigned long long mask = __ballot(need);
        // new packets only while the queues are not loaded: a workgroup that keeps every lane AND both queues
        // full cannot move packets between its waves any more
        // ... and only for several lanes at a time (the emission code costs the wave the same for 1 lane or 64),
        // unless the wave has nothing else to do
        const int n_need = __popcll(mask);
        const bool emit_now = n_need >= emit_min || __ballot(p.st != S_EMIT) == 0ull;
        if (mask && emit_now && !no_more_ids && (rq_ld(&Q->fly_top) + rq_ld(&Q->srv_top)) <= emit_qmax) {
          if (pk_next >= pk_end) {
            const int leader = __ffsll((long long)mask) - 1;
            unsigned long long base = 0;
            if (lane == leader) {
              // pending is raised BEFORE the ids are taken and corrected afterwards, so that nobody can
              // see "ids exhausted and nothing pending" while a reservation is under way
              atomicAdd(&Q->n_pending, (int)PK_BATCH);
              __threadfence_block();
              base = atomicAdd(A.next_packet, (unsigned long long)PK_BATCH);
            }
            base = __shfl(base, leader);
            pk_next = base < A.n_packets ? base : A.n_packets;
            pk_end = (base + PK_BATCH < A.n_packets) ? base + PK_BATCH : A.n_packets;
            if (pk_end < pk_next) pk_end = pk_next;
            if (lane == leader) {
              const int got = (int)(pk_end - pk_next);
              if (got < (int)PK_BATCH) atomicAdd(&Q->n_pending, got - (int)PK_BATCH);
              if (got == 0) { __threadfence_block(); rq_st(&Q->ids_done, 1); }
            }
            if (pk_end == pk_next) no_more_ids = true;
          }
          const unsigned long long avail = pk_end - pk_next;
          const unsigned long long rank = (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
          const unsigned long long cnt = (unsigned long long)__popcll(mask);
          const unsigned long long my = pk_next + rank;
          const bool served = need && (rank < avail);
          pk_next += (cnt < avail) ? cnt : avail;
          if (served) {
            p.rng.init(A.seed, A.first_packet + my);
            c_pack++;
            p.pk_cross = 0;
            float f[12];
            p.rng.emission_event(f);
            p.tau_rand = f[8];
            p.lambda = select_wl_em(T, M, f[0]);
            atomic_add_f64(&A.n_sent[p.lambda - 1], 1.0);
            bool lintersect;
            p.flag_scatt = false;
            p.S[0] = 1.0; p.S[1] = 0.0; p.S[2] = 0.0; p.S[3] = 0.0;
            CylEmitOps<L3D> ops{T, M, p.ri, p.zj, p.k};
            const int rc = emit_packet(M, f, p.lambda, T.fstar[p.lambda - 1], M.frac_E_disk[p.lambda - 1],
                                       M.prob_E_cell ? M.prob_E_cell + (size_t)(M.n_cells + 1) * (p.lambda - 1) : nullptr,
                                       ops, p.x, p.y, p.z, p.u, p.v, p.w, p.flag_star, p.flag_ism, lintersect);
            if (rc) { *A.err = rc; rq_st(&Q->abort_flag, 1); }
            p.st = lintersect ? S_NEWFLIGHT : S_EXITED;
          }
        }
      }
      if (p.st == S_EXITED) {  // capteur
        if (!p.flag_ism) { capteur<POLA>(M, A.sed, p.lambda, p.u, p.v, p.w, p.S, p.flag_star, p.flag_scatt); c_esc++; }
        p.st = S_EMIT;
        finished++;
      }
      if (p.st == S_INTERACT) {  // dust_transfer.f90:1260-1402
        float g[8];
        p.rng.interaction_event(g);
        p.tau_rand = g[5];
        double u1, v1, w1;
        const int ic = cell_index<L3D>(n_rad, nz, p.ri, p.zj, p.k);
        interact<POLA>(T, M, g, p.lambda, p.u, p.v, p.w, u1, v1, w1, p.S, p.flag_star, p.flag_scatt, c_scatt, c_abs, [&]() {
          double E;
          if (A.frozen) E = A.E_prior[ic];
          else {
            E = __hip_atomic_load(&A.E_abs[ic], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (LDSE) E += E_lds[ic] * (double)gridDim.x;
            E *= A.qscale;
          }
          return E;
        }, M.volume + ic);
        if (!p.flag_scatt) p.flag_ism = false;
        p.u = u1; p.v = v1; p.w = w1;
        p.st = S_NEWFLIGHT;
      }
      if (p.st == S_NEWFLIGHT) {
        const float rand = p.tau_rand;
        p.extr = (rand > 1.0e-6f) ? -log(1.0 - (double)rand) : (double)rand;
        const int i_star = intersect_stars(M, p.x, p.y, p.z, p.u, p.v, p.w);
        p.star_key = -1;
        if (i_star > 0) {
          const int* sc = &M.star_cell[4 * (i_star - 1)];
          p.star_key = sc[0] + (n_rad + 2) * ((sc[1] + nz + 1) + (2 * nz + 3) * (sc[2] - 1));
        }
        c_flight++;
        p.st = S_FLIGHT;
      }
      // per-flight constants of whatever flies in this lane now (new flight, or one kept because the
      // FLY queue was full)
      {
        const double a = p.u * p.u + p.v * p.v;
        inv_a = (a > TINY_REAL) ? 1.0 / a : HUGE_REAL;
        inv_w = (fabs(p.w) > TINY_REAL) ? 1.0 / p.w : copysign(HUGE_DP, p.w);
        kap = T.kappa[p.lambda - 1]; kab = T.kabs[p.lambda - 1];
        if (p.st == S_FLIGHT)
          kf = is_real_cell<L3D>(n_rad, nz, p.ri, p.zj) ? M.kappa_factor[cell_index<L3D>(n_rad, nz, p.ri, p.zj, p.k)] : 0.0;
      }
      if (__ballot(p.st != S_EMIT) == 0ull) {  // the wave holds no packet at all
        if (no_more_ids && rq_ld(&Q->n_pending) == 0) break;
        if (no_more_ids) {  // packets are with the flyers: wait for them to come back
          __builtin_amdgcn_s_sleep(32);
          if (++idle_spins > (1 << 26)) { *A.err = 15; rq_st(&Q->abort_flag, 1); }
        }
      } else {
        idle_spins = 0;
      }
      // the first crossings of every flight
#pragma unroll 1
      for (int it = 0; it < k_short; ++it) {
        if (__ballot(p.st == S_FLIGHT) == 0ull) break;
        RQ_DIAG(if (lane == 0) d_srv_iters++;)
        if (p.st == S_FLIGHT) finished += roles_cross<L3D, POLA, DARK, LDSE>(T, M, A, E_lds, p, inv_a, inv_w, kap, kab, kf, c_cross, c_kill, c_dark);
      }
    }

    // ---- bookkeeping common to both roles ---------------------------------------------------------
    {
      const int fin = __popcll(__ballot(finished > 0)) + __popcll(__ballot(finished > 1));
      if (fin > 0 && lane == 0) atomicAdd(&Q->n_pending, -fin);
    }
    if (LDSE && ((ep + 1) % A.flush_every) == 0) {  // barrier-free partial fold (see thermal_body)
      const int n_waves = (blockDim.x + 63) >> 6;
      const int slice = (wave + (ep + 1) / A.flush_every) % n_waves;
      const int per = (M.n_cells + n_waves - 1) / n_waves;
      const int i0 = slice * per, i1 = (i0 + per < M.n_cells) ? i0 + per : M.n_cells;
      for (int i = i0 + lane; i < i1; i += 64) {
        const unsigned long long bits = atomicExch(reinterpret_cast<unsigned long long*>(&E_lds[i]), 0ull);
        const double e = __longlong_as_double((long long)bits);
        if (e != 0.0) atomic_add_f64(&A.E_abs[i], e);
      }
    }
  }

  if ((A.flags & 2) && lane == 0) {
    const unsigned long long t_end = wall_clock64();
    atomicAdd(&A.counters[10], t_end - t_start);
    atomicAdd(&A.counters[11], (t_ids_out ? t_ids_out : t_end) - t_start);
    atomicMax(&A.counters[12], ~t_start);
    atomicMax(&A.counters[13], t_end);
    atomicAdd(&A.counters[14], 1ull);
  }
  if (LDSE) {
    __syncthreads();
    for (int i = threadIdx.x; i < M.n_cells; i += blockDim.x) {
      const double e = E_lds[i];
      if (e != 0.0) atomic_add_f64(&A.E_abs[i], e);
    }
  }
  unsigned int cs[8] = {c_pack, c_cross, c_flight, c_scatt, c_abs, c_esc, c_kill, c_dark};
#ifdef MCGPU_COUNT_ITERS  // the eight counters carry the schedule's statistics instead
  cs[0] = d_in_flight; cs[1] = d_handed; cs[6] = d_popped; cs[3] = d_empty;  // lanes, summed over server rounds
  cs[2] = d_srv_rounds; cs[5] = d_fly_rounds;                                 // rounds
  cs[4] = d_srv_iters; cs[7] = d_fly_iters;                                   // crossing iterations
#endif
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    unsigned long long vsum = cs[q];
    for (int off = 32; off > 0; off >>= 1) vsum += __shfl_down(vsum, off);
    if (lane == 0 && vsum) atomicAdd(&A.counters[q], vsum);
  }
}

template <bool L3D, bool POLA, bool DARK, bool LDSE>
__global__ void __launch_bounds__(MCGPU_LDS_BLOCK) k_thermal_roles(const DevModel M, const RunArgs A, int n_flyers,
                                                                   int k_short, int fly_iters, int fly_idle,
                                                                   int emit_qmax, int emit_min) {
  extern __shared__ double lds_raw[];
  roles_body<L3D, POLA, DARK, LDSE>(M, A, lds_raw, n_flyers, k_short, fly_iters, fly_idle, emit_qmax, emit_min);
}

}  // namespace mcgpu
