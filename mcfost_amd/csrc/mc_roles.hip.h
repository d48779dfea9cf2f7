// The default schedule of the thermal packet loop on cylindrical grids: the waves of a workgroup take
// ROLES and pass packets to each other through lock-free queues in LDS.
//
// Why roles: the flight lengths are heavy-tailed (most flights of a packet random-walking in the thick
// inner disk are 1-2 cell crossings, a few are 100+, and those few hold 94 % of all crossings).  With one
// packet per lane and every lane doing everything (thermal_body), a wavefront leaves the crossing loop to
// serve the short flights again and again while its long flights idle: lane utilisation of the crossing
// loop 54 %.
// Why records: the crossing loop needs ~90 VGPRs, the interaction code ~230 when it holds a whole packet
// in registers next to its own temporaries -- fused, the kernel ran at 2 waves per SIMD (256 VGPRs).  Here
// the SERVING role never holds a packet in registers: a packet that is not in flight lives in a RECORD in
// LDS, and the serving code runs as short phases (emission / interaction / new flight / first crossings)
// that each load the few fields they need from the record and store what they changed.  The kernel is
// compiled for 128 VGPRs: 1024-thread workgroups, 4 waves per SIMD.
//
//   SERVING   a lane owns a record; per round: interaction (dust_transfer.f90:1260-1402), optical depth
//             and star test of the next flight, the first k_short crossings.  A packet still in flight
//             after those is a long flight: the record's index goes to the FLY ring (no copy) and the
//             lane takes a record from the SRV ring (a packet that waits for its interaction) or a free
//             record for a new packet (emission, dust_transfer.f90:529-541).
//   FLYING    a lane holds a packet in registers and only crosses cells (physical_length,
//             optical_depth.f90:77-178).  Every fly_iters crossings (or when fly_idle lanes have
//             nothing to fly) the lanes whose packet stopped SWAP it with a record from the FLY ring
//             (field by field, so the record now holds the stopped packet) and put that index on the
//             SRV ring; empty lanes load a FLY record and return its index to the FREE ring.
//
// Rings: three index rings in LDS (FREE, FLY, SRV) of RQ_CAP >= n_rec entries.  An entry is one 32-bit word
// (16-bit position tag | record index) written with one store, so publishing is atomic; a ring can never be
// full (there are only n_rec indices), so pushing is one ds_add_rtn on the tail plus the entry stores;
// popping is a compare-and-swap on the head after the entries' tags have been checked.  No lock anywhere.
// Results do not depend on who runs a packet (counter-based random numbers keyed by the packet id), so
// this schedule reproduces thermal_body packet for packet.  Every wait is bounded (error 15) so that a
// logic error cannot hang the GPU.
//
// Chunks without tails (BIN kernels; RunArgs::carry_*).  A run with binned deposits is a sequence of launches (the log
// is folded between them), and a persistent kernel's launch ends with a TAIL: a packet is sequential, a lone packet
// costs ~2.6 us per event, and the slowest of a few million packets has 10^4 events -- measured: 65 of the 95 ms of a
// 3.75e6-packet chunk of ref4.1_3D.  So a chunk does not finish its packets: when a workgroup finds the global work
// counter exhausted, every wave writes the packets it holds (registers or owned records, and the work items it had
// reserved but not started) to carry_out as records, the workgroup sweeps the FLY and SRV rings into it, and the
// launch ends.  The next chunk's work items are [carried records | new packet ids]: a serving lane that takes an item
// below *carry_in_n loads that record instead of emitting a packet.  Only the last chunk runs to the end.  A packet's
// random numbers and its whole state travel in the record, so the results are the single launch's, packet for packet.
#pragma once
#include "mc_device.hip.h"
#include "mc_voronoi.hip.h"
#include "mc_binned.hip.h"

#ifndef MCGPU_3D_BRANCHY  // 1: the 3D crossing through roles_cross / cross_cell_lean (A/B builds), 0: fly_step_3d
#define MCGPU_3D_BRANCHY 0
#endif

#ifdef MCGPU_COUNT_ITERS  // diagnostic build (tools/roles_check.py diag): statements that only count
#define RQ_DIAG(...) __VA_ARGS__
#else
#define RQ_DIAG(...)
#endif

namespace mcgpu {

constexpr int RQ_CAP = 512;          // entries per ring (power of two, >= records per workgroup)
constexpr int RQ_MIN_REC = 96;       // fewer records than this: the single-role kernel runs instead
enum : int { RQ_FREE = 0, RQ_FLY = 1, RQ_SRV = 2 };

// a packet that is not in a flyer's registers (array of structures: every field is an immediate offset
// from one per-lane address; strides 136 / 120 B keep consecutive records on different banks)
template <bool POLA>
struct alignas(8) Rec {
  double x, y, z, u, v, w, extr;
  double S[POLA ? 4 : 1];
  int ri, zj, k, lambda, star_key;
  uint32_t p_lo, p_hi, event;
  int flags;  // state | ST_STAR | ST_SCATT | ST_ISM
  uint32_t pk_cross;
  float tau_rand;
  int pad[POLA ? 1 : 3];
};
static_assert(sizeof(Rec<true>) == 136 && sizeof(Rec<false>) == 120, "record stride");

// a record copied (src = nullptr: cleared) word by word through two registers -- a struct assignment would stage all 34
// dwords in registers, and the role kernels have none to spare
template <bool POLA>
__device__ inline void rec_copy(Rec<POLA>* dst, const Rec<POLA>* src) {
  unsigned long long* d = reinterpret_cast<unsigned long long*>(dst);
  const unsigned long long* s = reinterpret_cast<const unsigned long long*>(src);
#pragma unroll 1
  for (int i = 0; i < (int)(sizeof(Rec<POLA>) / 8); ++i) d[i] = s ? s[i] : 0ull;
}

struct RqCtl {
  int n_pending, ids_done, abort_flag, suspend;  // suspend: the workgroup is handing its packets over (CARRY)
  unsigned int head[3], tail[3];
  int n_srv, cooldown;  // waves [0, n_srv) serve; adapted by wave 0 (see roles_body)
  int idle_f, idle_s;   // lanes the flying / serving waves could not fill since the last adaptation
  int beat;             // bumped by every wave in every round in which it had work: the idle waves' sign of life
  int pad1;
};
static_assert(sizeof(RqCtl) == 64, "control block");

__host__ __device__ inline size_t rq_lds_bytes(bool pola, int n_rec) {
  return sizeof(RqCtl) + (size_t)3 * RQ_CAP * sizeof(unsigned int) + (size_t)n_rec * (pola ? sizeof(Rec<true>) : sizeof(Rec<false>));
}
// records that fit into `free_bytes` of LDS (0: not enough for this schedule)
__host__ __device__ inline int rq_records_that_fit(bool pola, size_t free_bytes) {
  const size_t fixed = sizeof(RqCtl) + (size_t)3 * RQ_CAP * sizeof(unsigned int);
  if (free_bytes <= fixed) return 0;
  size_t n = (free_bytes - fixed) / (pola ? sizeof(Rec<true>) : sizeof(Rec<false>));
  if (n > (size_t)RQ_CAP) n = RQ_CAP;
  return n >= (size_t)RQ_MIN_REC ? (int)n : 0;
}

__device__ inline int rq_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline unsigned int rq_ldu(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void rq_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void rq_stu(unsigned int* p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline int rq_count(const RqCtl* Q, int q) { return (int)(rq_ldu(&Q->tail[q]) - rq_ldu(&Q->head[q])); }

// Push the indices of the lanes with `want` onto ring q (never full).  The records they name must have been
// written before the call (the fence orders those stores in front of the entries).
__device__ inline void rq_push(RqCtl* Q, unsigned int* rings, int q, int lane, bool want, int rid) {
  const unsigned long long m = __ballot(want);
  if (m == 0ull) return;
  __threadfence_block();
  const int n = __popcll(m), rank = __popcll(m & ((1ull << lane) - 1ull));
  const int leader = __ffsll((long long)m) - 1;
  unsigned int pos = 0;
  if (lane == leader) pos = atomicAdd(&Q->tail[q], (unsigned int)n);
  pos = __shfl(pos, leader);
  if (want) {
    const unsigned int at = pos + (unsigned int)rank;
    rq_stu(&rings[q * RQ_CAP + (at & (RQ_CAP - 1))], ((at + 1u) << 16) | (unsigned int)rid);
  }
}

// Pop up to one index per lane with `want` from ring q; returns the index or -1.  Lanes are served in lane
// order; entries whose producer has reserved but not yet written them end the batch early.
__device__ inline int rq_pop(RqCtl* Q, unsigned int* rings, int q, int lane, bool want) {
  const unsigned long long m = __ballot(want);
  if (m == 0ull) return -1;
  const int n = __popcll(m), rank = __popcll(m & ((1ull << lane) - 1ull));
  const int leader = __ffsll((long long)m) - 1;
  for (int attempt = 0; attempt < 8; ++attempt) {
    const unsigned int h = rq_ldu(&Q->head[q]), t = rq_ldu(&Q->tail[q]);  // (same address in every lane)
    const int avail = (int)(t - h);
    if (avail <= 0) return -1;
    int c = n < avail ? n : avail;
    const bool mine = want && rank < c;
    const unsigned int at = h + (unsigned int)rank;
    unsigned int word = 0;
    if (mine) word = rq_ldu(&rings[q * RQ_CAP + (at & (RQ_CAP - 1))]);
    const bool ok = mine && (word >> 16) == ((at + 1u) & 0xFFFFu);
    const unsigned long long bad = __ballot(mine && !ok);
    if (bad) c = __popcll(m & ((1ull << (__ffsll((long long)bad) - 1)) - 1ull));  // entries in front of the first unpublished one
    if (c == 0) return -1;
    unsigned int old = 0;
    if (lane == leader) old = atomicCAS(&Q->head[q], h, h + (unsigned int)c);
    old = __shfl(old, leader);
    if (old == h) {
      __threadfence_block();
      return (want && rank < c) ? (int)(word & 0xFFFFu) : -1;
    }
  }
  return -1;
}

// what a lane needs to cross cells (registers)
struct Flight {
  double x, y, z, u, v, w, extr, S0, inv_a, inv_w, kf, kap, kab;
  int ri, zj, k, star_key, st;
  unsigned int pk_cross;
  int lam;  // (VAR: the packet's wavelength, for the per-cell opacities)
  int ic;   // (2D crossing: the 0-based index of the cell (ri, zj), n_cells outside the real cells; flight_constants sets it)
};

__device__ inline void flight_clear(Flight& F) {
  F.x = F.y = F.z = F.u = F.v = 0.0; F.w = 1.0; F.extr = 0.0; F.S0 = 1.0; F.inv_a = F.inv_w = F.kf = F.kap = F.kab = 0.0;
  F.ri = 0; F.zj = 1; F.k = 1; F.star_key = -1; F.st = S_EMIT; F.pk_cross = 0u; F.lam = 1; F.ic = 0;
}

// VAR (lvariable_dust): the opacities change from cell to cell.  DevModel::v_kk holds, per (cell, wavelength), the pair
// (kappa(p_icell, lambda) * kappa_factor(icell), kappa_abs_LTE(p_icell, lambda)) -- one 16-byte gather per cell entered
// instead of the 8 bytes of kappa_factor; the flight then carries kap = 1, kf = the product, kab = the cell's value.
__device__ inline void var_cell_opacities(const DevModel& M, Flight& F, int ic) {
  const double2 kk = M.v_kk[(size_t)ic * M.n_lambda + (F.lam - 1)];
  F.kf = kk.x;
  F.kab = kk.y;
}

// REUSE (the tail kernel, whose trapped packets start flight after flight in one cell): F.ic / F.kf still hold the last
// flight's cell and its kappa_factor (F.ic = -1: nothing yet), and the load from HBM is skipped while the cell is the same
template <bool L3D, bool VAR = false, bool REUSE = false>
__device__ inline void flight_constants(const Lds& T, const DevModel& M, Flight& F, int lambda) {
  const double a = F.u * F.u + F.v * F.v;  // cylindrical_grid.f90:941-952
  F.inv_a = (a > TINY_REAL) ? 1.0 / a : HUGE_REAL;
  F.inv_w = (fabs(F.w) > TINY_REAL) ? 1.0 / F.w : copysign(HUGE_DP, F.w);
  const int ic_new = is_real_cell<L3D>(M.n_rad, M.nz, F.ri, F.zj) ? cell_index<L3D>(M.n_rad, M.nz, F.ri, F.zj, F.k) : M.n_cells;
  if (REUSE && !VAR && ic_new == F.ic) {
    F.kap = T.kappa[lambda - 1];
    F.kab = T.kabs[lambda - 1];
    return;
  }
  F.ic = ic_new;
  if (VAR) {
    F.lam = lambda;
    F.kap = 1.0;
    var_cell_opacities(M, F, F.ic);
    return;
  }
  F.kap = T.kappa[lambda - 1];
  F.kab = T.kabs[lambda - 1];
  F.kf = M.kappa_factor[F.ic];   // (the device's array ends in a zero entry for "no cell")
}

// One cell crossing of a packet in flight (physical_length's loop body, optical_depth.f90:77-178).  Returns the
// number of packets this lane finished (0 or 1).
// DARK: the reference tests l_dark_zone(icell0) at the top of the NEXT loop turn and then puts the packet back at
// the point where it entered the cell it has just crossed, direction reversed (:104-112).  The same thing is done
// here at the end of the crossing that leads into the dark cell -- the entry point is still at hand, so the packet
// needs no memory of it (a flight never starts inside a dark cell: packets are mirrored at its edge and the dark
// cells emit nothing, thermal_emission.f90:1817).
// BIN: the deposit is handed back (dep_ic >= 0: cell, dep_v: value) for the caller's bin_deposit, which needs the whole
// wave in converged control flow (mc_binned.hip.h).
template <bool L3D, bool DARK, bool LDSE, bool BIN = false>
__device__ inline int roles_cross(const Lds& T, const DevModel& M, const RunArgs& A, double* E_lds, Flight& p,
                                  unsigned int& c_cross, unsigned int& c_kill, unsigned int& c_dark, int& dep_ic, double& dep_v) {
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
  MCGPU_CROSS<L3D>(T, M, p.x, p.y, p.z, p.u, p.v, p.w, p.inv_a, p.inv_w, p.ri, p.zj, p.k, x1, y1, z1, ri1, zj1, k1, l);
  c_cross++;
  // (kf was loaded at the end of the previous crossing: first used here, behind the geometry, so that the
  // latency of that load is covered by it)
  const double opacity = real_cell ? p.kap * p.kf : 0.0;
  const double tau = l * opacity;
  if (tau > p.extr) {
    const double lc = l * (p.extr / tau);
    if (real_cell && !MCGPU_DIAG(A.flags, 1)) {
      if (BIN) { dep_ic = ic; dep_v = p.kab * lc * p.S0; }
      else deposit<LDSE>(A.E_abs, E_lds, ic, p.kab * lc * p.S0);
    }
    p.x = p.x + lc * p.u;
    p.y = p.y + lc * p.v;
    p.z = p.z + lc * p.w;
    if (L3D) index_cell<L3D>(T, M, p.x, p.y, p.z, p.ri, p.zj, p.k, p.ri);
    p.st = S_INTERACT;
  } else {
    p.extr = p.extr - tau;
    if (real_cell && !MCGPU_DIAG(A.flags, 1)) {
      if (BIN) { dep_ic = ic; dep_v = p.kab * l * p.S0; }
      else deposit<LDSE>(A.E_abs, E_lds, ic, p.kab * l * p.S0);
    }
    const bool next_real = is_real_cell<L3D>(n_rad, nz, ri1, zj1);
    const int ic1 = next_real ? cell_index<L3D>(n_rad, nz, ri1, zj1, k1) : 0;
    if (DARK && next_real && M.dark[ic1]) {
      p.u = -p.u; p.v = -p.v; p.w = -p.w;  // back at the entry point of this cell, an interaction follows there
      c_dark++;
      p.st = S_INTERACT;
    } else {
      p.x = x1; p.y = y1; p.z = z1;
      p.ri = ri1; p.zj = zj1; p.k = k1;
      p.kf = next_real ? M.kappa_factor[ic1] : 0.0;
    }
  }
  if (++p.pk_cross > 200000000u) { *A.err = 13; p.st = S_EMIT; return 1; }  // a packet that never leaves: flag it, drop it
  return 0;
}

// The same crossing for 2D grids, written for the instruction ISSUE rate that bounds this kernel (measured: the
// loop runs at ~4 cycles per wave instruction of ANY kind at 2, 3 and 4 waves per SIMD alike, so its time is its
// instruction count): one straight-line stream for all 64 lanes -- every lane computes, results are committed by
// selects on `go` -- instead of nested divergent branches (each costs s_and_saveexec / s_or / s_cbranch plus the
// moves that merge the two sides).  The only branches left are the ones whose bodies are expensive or have side
// effects: the stop (an FP64 division), the deposit (an LDS atomic) and the rare default-real zj recomputation.
// Every value that decides an index or a position is computed by the reference's expression, exactly as in
// cross_cell_lean / roles_cross above (cylindrical_grid.f90:918-1175, optical_depth.f90:77-178).
// OUT: the deposit is handed back (dep_ic >= 0, dep_v) instead of being made (the tail kernel, mc_tail.hip.h)
template <bool DARK, bool LDSE, bool MRW = false, bool OUT = false, bool VAR = false>
__device__ __forceinline__ int fly_step_2d(const Lds& T, const DevModel& M, const RunArgs& A, double* E_lds, Flight& p,
                                           unsigned int& c_cross, unsigned int& c_kill, unsigned int& c_dark,
                                           int* dep_ic = nullptr, double* dep_v = nullptr) {
  const int n_rad = M.n_rad, nz = M.nz;
  // correct_plus = 1 + e and correct_moins = 1 - e with e = 45 * 2^-52 EXACTLY (1e-14 rounds to 45 units in the last
  // place of 1.0 and 90 of the doubles below it): a * correct_plus = a + a e is then one fma(a, e, a) -- the very
  // product, rounded once -- and the choice between the two factors is the sign bit of e (section 2 below)
  static_assert((1.0 + GRID_PREC) - 1.0 == 0x1.68p-47 && 1.0 - (1.0 - GRID_PREC) == 0x1.68p-47, "grid_prec is not 45 ulp");
  const double cp = 1.0 + GRID_PREC;
  const bool active = (p.st == S_FLIGHT);
  const int ri0 = p.ri, zj0 = p.zj;
  const double x0 = p.x, y0 = p.y, z0 = p.z, u = p.u, v = p.v, w = p.w;
  const bool top = (zj0 == nz + 1);
  // test_exit_grid_cyl (cylindrical_grid.f90:680-704) in closed form; the star's cell (optical_depth.f90:91-97)
  const bool out = (ri0 == n_rad + 1) || (top && (fabs(z0) > M.zmaxmax));
  const bool killed = (p.star_key >= 0) && (ri0 + (n_rad + 2) * (zj0 + nz + 1) == p.star_key);
  const bool go = active && !out && !killed;
  const bool hole = (ri0 == 0);
  // the cell's index travels with the flight (p.ic; n_cells = "no cell"): what the last crossing computed for the
  // kappa_factor of this cell is the address of this crossing's deposit
  const int ic = p.ic;
  const bool real_cell = ic < M.n_cells;
  // the row of this radial index (lanes outside the grid read a valid row; their results are discarded)
  const RowT& R0 = T.row[ri0];

  // 1) radial wall (:959-1000); rl_in / rl_out carry the correction factors (RowT)
  const double r_2 = x0 * x0 + y0 * y0;
  const double dot = x0 * u + y0 * v;
  const double b = dot * p.inv_a;
  const double c_in = (r_2 - R0.rl_in) * p.inv_a;
  const double c_out = (r_2 - R0.rl_out) * p.inv_a;
  const double bb = b * b;
  const double d_in = bb - c_in;
  const double d_out = fmax(bb - c_out, 0.0);
  const bool use_in = hole || ((dot < 0.0) && !(d_in < 0.0));
  const double delta = use_in ? d_in : d_out;
  const int delta_rad = (use_in && !hole) ? -1 : 1;
  const double rac = sqrt_nonneg(delta);   // (in the hole delta > 0: the packet is inside that circle)
  const double s1 = (-b - rac) * cp, s2 = (-b + rac) * cp;
  const double s_pos = (s1 == 0.0) ? GRID_PREC : s1;
  const double s = (hole || (s1 < 0.0)) ? s2 : s_pos;

  // 2) vertical wall (:1003-1055), 2D: zj >= 1, the midplane mirrors
  const double dz = w * z0;
  const bool away = dz > 0.0;
  const bool flip = !away && (zj0 == 1);  // through the midplane to the mirror side
  const int jsel = away ? zj0 + 1 : (zj0 == 1 ? 2 : zj0);
  // (jsel = nz + 2, where z_lim is 1e30, only occurs for away && top, which the 1e10 below replaces)
  double zmag = (jsel <= nz) ? ((double)jsel - 1.0) * R0.ch : R0.zmax;
  // zmag * (away ? correct_plus : correct_moins), see the top of the function
  zmag = __builtin_fma(zmag, __longlong_as_double(away ? 0x3D06800000000000ll : (long long)0xBD06800000000000ull), zmag);
  zmag = (away && top) ? 1.0e10 : zmag;
  const bool neg = (z0 < 0.0) != flip;
  // zl = neg ? -zmag : zmag (zmag >= 0): the sign bit is set directly
  const double zl = __longlong_as_double(__double_as_longlong(zmag) | (neg ? (long long)0x8000000000000000ull : 0ll));
  const int delta_zj = away ? (top ? 0 : 1) : ((zj0 == 1) ? 1 : -1);
  double t = (zl - z0) * p.inv_w;
  t = (t < 0.0) ? GRID_PREC : t;
  // dz == 0: no vertical wall ahead (1e10, :1052); in the hole the reference does not look for one (:1003): the
  // same 1e10 serves, being longer than any chord of the hole
  t = ((dz == 0.0) || hole) ? 1.0e10 : t;

  // 3) nearest wall (:1098-1156)
  const bool rad = (s < t);
  const double l = fmin(s, t);  // (= rad ? s : t; both are finite)
  // (x1, y1 = x0 + l u, y0 + l v are formed in the commit below, together with the stopping point)
  // products rounded before the sum, like the reference build (see cross_cell_lean)
  double z1 = nd_add(z0, nd_mul(l, w));
  const int ri1 = rad ? ri0 + delta_rad : ri0;
  // zj of the end point for a radial move (:1116): the fast form of zj_capped -- floor(|z1| nz / zmax) + 1, at most
  // nz + 1 (the minimum is taken on the double: |z1| is unbounded above the disk); in the hole rzn = 0 gives the
  // reference's zj = 1 (:1117) -- and its rare default-real fallback, taken within 1e-4 of an integer
  const double qd = fabs(z1) * T.row[ri1].rzn;
  const double fl = floor(qd);
  int zjr = (int)fmin(fl, (double)nz) + 1;
  const bool rad_in = rad && (ri1 >= 1) && (ri1 <= n_rad);
  const double fr = qd - fl;
  if (__builtin_expect(go && rad_in && (fr < 1.0e-4 || fr > 1.0 - 1.0e-4), 0)) {  // (rare)
    const int zq = zj_from_z_real(T, nz, fabs(z1), ri1);
    zjr = zq > nz ? nz + 1 : zq;
  }
  zjr = (ri1 > n_rad) ? zj0 : zjr;
  const int zj1 = rad ? zjr : zj0 + delta_zj;
  z1 = (z1 == 0.0) ? GRID_PREC : z1;

  // 4) optical depth of the crossing, stop or go on (optical_depth.f90:102, 134-146)
  // (kf was loaded at the end of the previous crossing: first used here, behind the geometry)
  // (outside the real cells p.kf = 0: flight_constants sets it so, and the commit below reads the zero entry that
  // ends the device's kappa_factor array)
  const double opacity = p.kap * p.kf;
  const double tau = l * opacity;
  const bool stop = go && (tau > p.extr);
  double lc = l;
  if (__builtin_expect(stop, 0)) lc = l * (p.extr / tau);  // (once per flight)
  // save_radiation_field (radiation_field.f90:53)
  if (go && real_cell && !MCGPU_DIAG(A.flags, 1)) {
    if (OUT) { *dep_ic = ic; *dep_v = p.kab * lc * p.S0; }
    else deposit<LDSE>(A.E_abs, E_lds, ic, p.kab * lc * p.S0);
  }

  // the next cell; DARK: mirrored back at the wall of a dark cell (see roles_cross)
  const bool next_real = (ri1 >= 1) && (ri1 <= n_rad) && (zj1 >= 1) && (zj1 <= nz);
  const int ic1 = next_real ? (ri1 - 1) + n_rad * (zj1 - 1) : M.n_cells;  // (n_cells: the entry of "no cell", 0)
  bool mirror = false;
  if (DARK) mirror = go && !stop && next_real && M.dark[next_real ? ic1 : 0];
  const bool move = go && !stop && !mirror;
  // (VAR: the pair (kappa kappa_factor, kappa_abs_LTE) of the next cell and this wavelength, see var_cell_opacities)
  const double2 kk1 = VAR ? M.v_kk[(size_t)ic1 * M.n_lambda + (p.lam - 1)] : make_double2(M.kappa_factor[ic1], 0.0);
  const double kf1 = kk1.x;

  // 5) commit
  // x and y: one multiply-add with the length that applies (l to the wall, lc to the stopping point, 0: the packet
  // stays) instead of selecting among three finished points -- the same x0 + l u / x0 + lc u as before
  const double lf = (stop || move) ? lc : 0.0;  // (lc = l unless the packet stops)
  p.x = x0 + lf * u;
  p.y = y0 + lf * v;
  p.z = move ? z1 : z0 + lf * w;  // (the wall point z1 keeps its own rounding and zero fix; stopping point / stay as x, y)
  // (a packet that stopped, left or was not in flight does not read extr again before its next flight sets it)
  p.extr = p.extr - tau;
  p.ri = move ? ri1 : ri0;
  p.zj = move ? zj1 : zj0;
  p.ic = move ? ic1 : ic;
  p.kf = move ? kf1 : p.kf;
  if (VAR) p.kab = move ? kk1.y : p.kab;
  if (DARK) {
    p.u = mirror ? -u : u; p.v = mirror ? -v : v; p.w = mirror ? -w : w;
    c_dark += mirror ? 1u : 0u;
  }
  int st = p.st;
  st = (active && out) ? S_EXITED : st;
  st = (active && !out && killed) ? S_EMIT : st;
  st = (stop || mirror) ? S_INTERACT : st;
  c_cross += go ? 1u : 0u;
  c_kill += (active && !out && killed) ? 1u : 0u;
  p.pk_cross += go ? 1u : 0u;
  // MRW: bit 31 of the crossing counter remembers that this flight has left the cell it started in
  if (MRW) p.pk_cross |= (move || mirror) ? 0x80000000u : 0u;
  const bool runaway = go && ((MRW ? (p.pk_cross & 0x7FFFFFFFu) : p.pk_cross) > 200000000u);  // a packet that never leaves: flag it, drop it
  if (runaway) { *A.err = 13; st = S_EMIT; }
  p.st = st;
  return ((active && !out && killed) || runaway) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// The 2D crossing parametrised along the flight (round 5; option "crossing" = 1, OFF by default).
// Along one straight flight the point is r(s) = r0 + s d, and everything the crossing recomputes from the current point
// -- r^2, r.d, the discriminants of both circles -- is a function of constants of the flight: with b0 = (x0 u + y0 v) / a
// and D0 = b0^2 - (x0^2 + y0^2) / a the wall of radius R is met at s = -b0 -+ sqrt(D0 + R^2 / a), a plane z = zl at
// s = (zl - z0) / w, and the packet moves inward while s + b0 < 0.  One multiply-add per circle, one for the current height,
// no position update per crossing (the point is formed when the packet stops or leaves the flying loop): the loop's
// common path is about two thirds of fly_step_2d's.
// What it is NOT: the reference's arithmetic.  cross_cylindrical_cell (cylindrical_grid.f90:918-1175) re-derives every
// crossing from the current point, nudges it by grid_prec and takes zj through default real; this form cannot reproduce
// its golden walks bit for bit (the cells it visits are the same but for ties at the rounding level), so the frozen
// packet-for-packet tests do not apply to it and it is gated by the statistical ones only (tests/test_param_crossing.py).
// Only the flying waves of the role kernel use it; serving waves and the tail kernel keep fly_step_2d.
// ---------------------------------------------------------------------------------------------
struct FlightParam {
  double s;    // the packet is at r0 + s d; F.x, F.y, F.z hold r0 while a lane is inside the flying loop
  double b0, D0;
  double zc;   // its height there, z0 + s w
  double kabS; // kappa_abs_LTE(lambda) Stokes(1): what a crossing's length is multiplied with for its deposit
};

__device__ __forceinline__ void param_begin(const Flight& F, FlightParam& P) {
  P.s = 0.0;
  P.b0 = (F.x * F.u + F.y * F.v) * F.inv_a;
  P.D0 = P.b0 * P.b0 - (F.x * F.x + F.y * F.y) * F.inv_a;
  P.zc = F.z;
  P.kabS = F.kab * F.S0;
}
// the point the packet has reached; a packet still in flight is put just beyond the wall it stands on (the exact crossing,
// which may pick it up, decides by the point's side: the role of the reference's correct_plus / correct_moins)
__device__ __forceinline__ void param_end(Flight& F, const FlightParam& P) {
  const double sm = (F.st == S_FLIGHT) ? P.s * (1.0 + 1.0e-13) : P.s;
  F.x = F.x + sm * F.u;
  F.y = F.y + sm * F.v;
  double z = F.z + sm * F.w;
  F.z = (z == 0.0) ? GRID_PREC : z;
}

// c_cross_wave: the wave's crossings, counted with a ballot and a scalar add (the caller adds them to one lane's counter);
// any_star: some lane of the wave flies towards a star's cell (else the test of that cell is skipped); the runaway
// test of the packet's crossing counter is the caller's, once per visit of the rings
template <bool LDSE>
__device__ __forceinline__ int fly_step_2d_param(const Lds& T, const DevModel& M, const RunArgs& A, double* E_lds, Flight& p,
                                                 FlightParam& P, unsigned int& c_cross_wave, unsigned int& c_kill, bool any_star) {
  const int n_rad = M.n_rad, nz = M.nz;
  const bool active = (p.st == S_FLIGHT);
  const int ri0 = p.ri, zj0 = p.zj;
  const double z0 = p.z, w = p.w;
  const double s0 = P.s;
  const double zc = P.zc;   // the current height
  const bool top = (zj0 == nz + 1);
  const bool out = (ri0 == n_rad + 1) || (top && (fabs(zc) > M.zmaxmax));
  bool killed = false;
  if (any_star) killed = (p.star_key >= 0) && (ri0 + (n_rad + 2) * (zj0 + nz + 1) == p.star_key);
  const bool go = active && !out && !killed;
  const bool hole = (ri0 == 0);
  const int ic = p.ic;
  const bool real_cell = ic < M.n_cells;
  const RowT& R0 = T.row[ri0];

  // 1) the radial wall ahead
  const double d_in = __builtin_fma(R0.rl_in, p.inv_a, P.D0);
  const double d_out = fmax(__builtin_fma(R0.rl_out, p.inv_a, P.D0), 0.0);
  const bool inward = (s0 + P.b0) < 0.0;
  const bool use_in = hole || (inward && !(d_in < 0.0));
  const bool minus = use_in && !hole;
  const double rac = sqrt_fast_nonneg(use_in ? d_in : d_out);
  const double s_rad = (minus ? -rac : rac) - P.b0;
  const int delta_rad = minus ? -1 : 1;

  // 2) the vertical wall ahead (2D: zj >= 1, the midplane mirrors).  The upper wall of layer nz is nz ch = zmax to rounding.
  const double dz = w * zc;
  const bool away = dz > 0.0;
  const bool flip = !away && (zj0 == 1);
  const int jsel = away ? zj0 + 1 : (zj0 == 1 ? 2 : zj0);
  const double zmag = ((double)jsel - 1.0) * R0.ch;
  const bool neg = (zc < 0.0) != flip;
  const double zl = __longlong_as_double(__double_as_longlong(zmag) | (neg ? (long long)0x8000000000000000ull : 0ll));
  const int delta_zj = away ? (top ? 0 : 1) : ((zj0 == 1) ? 1 : -1);
  double t = (zl - z0) * p.inv_w;
  t = ((dz == 0.0) | hole | (away & top)) ? 1.0e10 : t;

  // 3) the nearest wall, the cell behind it
  const bool rad = (s_rad < t);
  const double s1 = fmin(s_rad, t);
  const double l = fmax(s1 - s0, 0.0);   // (a wall the rounding puts behind the packet is crossed on the spot)
  const int ri1 = rad ? ri0 + delta_rad : ri0;
  const double z1 = __builtin_fma(s1, w, z0);
  int zjr = (int)fmin(floor(fabs(z1) * T.row[ri1].rzn), (double)nz) + 1;
  zjr = (ri1 > n_rad) ? zj0 : zjr;
  const int zj1 = rad ? zjr : zj0 + delta_zj;

  // 4) optical depth of the crossing, stop or go on (optical_depth.f90:102, 134-146)
  const double tau = l * (p.kap * p.kf);
  const bool stop = go && (tau > p.extr);
  double lc = l;
  if (__builtin_expect(stop, 0)) lc = l * (p.extr / tau);  // (once per flight)
  if (go && real_cell && !MCGPU_DIAG(A.flags, 1)) deposit<LDSE>(A.E_abs, E_lds, ic, P.kabS * lc);

  // (unsigned compares: 1 <= ri1 <= n_rad and 1 <= zj1 <= nz in one test each)
  const bool next_real = ((unsigned)(ri1 - 1) < (unsigned)n_rad) & ((unsigned)(zj1 - 1) < (unsigned)nz);
  const int ic1 = next_real ? (ri1 - 1) + n_rad * (zj1 - 1) : M.n_cells;
  const bool move = go && !stop;
  const double kf1 = M.kappa_factor[ic1];

  // 5) commit
  P.s = go ? s0 + lc : s0;   // (lc = l = s1 - s0 unless the packet stops)
  P.zc = move ? z1 : zc;     // (a packet that stops does not look at its height again in this loop)
  p.extr = p.extr - tau;
  p.ri = move ? ri1 : ri0;
  p.zj = move ? zj1 : zj0;
  p.ic = move ? ic1 : ic;
  p.kf = move ? kf1 : p.kf;
  int st = p.st;
  st = (active && out) ? S_EXITED : st;
  st = (active && !out && killed) ? S_EMIT : st;
  st = stop ? S_INTERACT : st;
  c_cross_wave += (unsigned int)__popcll(__ballot(go));
  p.pk_cross += go ? 1u : 0u;
  p.st = st;
  if (!any_star) return 0;
  c_kill += (active && !out && killed) ? 1u : 0u;
  return (active && !out && killed) ? 1 : 0;
}

// The crossing of a 3D cylindrical grid in the same straight-line, select-committed form (cross_cell_lean<true> +
// roles_cross above, statement for statement: cylindrical_grid.f90:918-1175 with the azimuthal walls :1058-1094,
// optical_depth.f90:77-178).  What stays a branch: the stop (a division, and the 3D re-indexing of the stopping point,
// optical_depth.f90:140 -> index_cell: a bisection and an atan2), the deposit, the azimuth of a packet that leaves the
// central hole (atan2), and the rare fallbacks of the zj recomputation.  The wall at tan(phi) = +-1e300 and the
// ordinary azimuthal wall share ONE division (numerator and denominator are selected first).
// BIN: the deposit is handed back (dep_ic >= 0, dep_v) for the caller's bin_deposit.
// DEFER (the role kernels): the re-indexing of a stopping point (optical_depth.f90:140 -> index_cell: a bisection, the
// default-real zj and an atan2, ~250 vector instructions) is NOT done here, where it would run in almost every iteration
// of the flying loop for the two or three lanes of 64 that stop in it, but by the interaction that follows, which runs
// for the stopped packets together; the packet carries the request as a negative azimuthal index k.
// MRW: bit 31 of the crossing counter remembers that the flight has left the cell it started in (fly_step_2d).
template <bool DARK, bool LDSE, bool BIN, bool VAR = false, bool DEFER = false, bool MRW = false>
__device__ __forceinline__ int fly_step_3d(const Lds& T, const DevModel& M, const RunArgs& A, double* E_lds, Flight& p,
                                           unsigned int& c_cross, unsigned int& c_kill, unsigned int& c_dark, int& dep_ic,
                                           double& dep_v) {
  const int n_rad = M.n_rad, nz = M.nz, n_az = M.n_az;
  const double cp = 1.0 + GRID_PREC;  // (correct_moins: in the rows' rl_in and in the fused product below, see fly_step_2d)
  const double r1e30 = 1.00000001504746621988e+30;
  const bool active = (p.st == S_FLIGHT);
  const int ri0 = p.ri, zj0 = p.zj, k0 = p.k;
  const double x0 = p.x, y0 = p.y, z0 = p.z, u = p.u, v = p.v, w = p.w;
  const int azj = zj0 < 0 ? -zj0 : zj0;
  const bool top = (azj == nz + 1);
  const bool out = (ri0 == n_rad + 1) || (top && (fabs(z0) > M.zmaxmax));
  const bool killed = (p.star_key >= 0) && (ri0 + (n_rad + 2) * ((zj0 + nz + 1) + (2 * nz + 3) * (k0 - 1)) == p.star_key);
  const bool go = active && !out && !killed;
  const bool hole = (ri0 == 0);
  // the cell's index travels with the flight and the radial index names ONE row of tables (see fly_step_2d)
  const int ic = p.ic;
  const bool real_cell = ic < M.n_cells;
  const RowT& R0 = T.row[ri0];

  // 1) radial wall (:959-1000)
  const double r_2 = x0 * x0 + y0 * y0;
  const double dot = x0 * u + y0 * v;
  const double b = dot * p.inv_a;
  const double c_in = (r_2 - R0.rl_in) * p.inv_a;
  const double c_out = (r_2 - R0.rl_out) * p.inv_a;
  const double bb = b * b;
  const double d_in = bb - c_in;
  const double d_out = fmax(bb - c_out, 0.0);
  const bool use_in = hole || ((dot < 0.0) && !(d_in < 0.0));
  const double delta = use_in ? d_in : d_out;
  const int delta_rad = (use_in && !hole) ? -1 : 1;
  const double rac = sqrt_nonneg(delta);
  const double s1 = (-b - rac) * cp, s2 = (-b + rac) * cp;
  const double s_pos = (s1 == 0.0) ? GRID_PREC : s1;
  const double s = (hole || (s1 < 0.0)) ? s2 : s_pos;

  // 2) vertical wall (:1003-1055), 3D: zj = -(nz+1) .. -1, 1 .. nz+1, the midplane is a wall
  const double dz = w * z0;
  const bool away = dz > 0.0;
  const bool neg = z0 < 0.0;
  const int jsel = away ? azj + 1 : azj;
  // (jsel = nz + 2, where z_lim is 1e30, only occurs for away && top, which the 1e10 below replaces)
  double zmag = (jsel <= nz) ? ((double)jsel - 1.0) * R0.ch : R0.zmax;
  // zmag * (away ? correct_plus : correct_moins) as one fma with e = +-45 * 2^-52 (fly_step_2d)
  zmag = __builtin_fma(zmag, __longlong_as_double(away ? 0x3D06800000000000ll : (long long)0xBD06800000000000ull), zmag);
  zmag = (away && top) ? 1.0e10 : zmag;
  const double zl = __longlong_as_double(__double_as_longlong(zmag) | (neg ? (long long)0x8000000000000000ull : 0ll));
  const int dzj_away = top ? 0 : (neg ? -1 : 1);
  const int dzj_back = (z0 > 0.0) ? ((zj0 == 1) ? -2 : -1) : ((zj0 == -1) ? 2 : 1);
  const int delta_zj = away ? dzj_away : dzj_back;
  double t = (zl - z0) * p.inv_w;
  t = (t < 0.0) ? GRID_PREC : t;
  t = (dz == 0.0) ? 1.0e10 : t;
  t = hole ? HUGE_REAL : t;

  // 3) azimuthal wall (:1058-1094)
  const double dp = x0 * v - y0 * u;
  const bool ccw = dp > 0.0;
  int kk = ccw ? k0 : k0 - 1;
  kk = (kk == 0) ? n_az : kk;
  const double tan_lim = T.tan_phi[kk - 1];
  const double den = v - u * tan_lim;
  const bool inf_wall = tan_lim > 1.0e299;
  const double num_w = inf_wall ? -x0 : -(y0 - x0 * tan_lim);
  const double den_w = inf_wall ? u : den;
  const bool den_ok = fabs(den_w) > (double)1.0e-6f;
  double tp = den_ok ? num_w / den_w : r1e30;
  tp = (tp < 0.0) ? r1e30 : tp;
  tp = (fabs(dp) < (double)1.0e-10f) ? r1e30 : tp;
  const double t_phi = hole ? HUGE_REAL : tp;

  // 4) nearest wall (:1098-1156)
  const bool rad = (s < t) && (s < t_phi);
  const bool vert = !rad && (t < t_phi);
  const double l = rad ? s : (vert ? t : t_phi);
  const double dv = (rad || vert) ? l : cp * t_phi;
  // products rounded before the sum, like the reference build (see cross_cell_lean)
  double z1 = nd_add(z0, nd_mul(dv, w));
  const int ri1 = rad ? ri0 + delta_rad : ri0;
  // zj of the end point of a radial (:1116, through default real) or azimuthal (:1139, FP64) move: one multiply by
  // nz / zmax decides it unless the quotient is within 1e-4 of an integer, where the reference's own expression runs
  const double qd = fabs(z1) * T.row[ri1].rzn;
  const double fl = floor(qd);
  const double fr = qd - fl;
  int zjr = (int)fmin(fl, (double)nz) + 1;   // (at most nz + 1; the minimum on the double: |z1| is unbounded above the disk)
  const bool ri1_in = (ri1 >= 1) && (ri1 <= n_rad);
  if (__builtin_expect(go && !vert && ri1_in && (fr < 1.0e-4 || fr > 1.0 - 1.0e-4), 0)) {  // (rare)
    // (far above the disk both expressions exceed nz and the cap below applies, as it does to zjr above)
    const double qe = rad ? 0.0 : floor(fabs(z1) / T.zmax[ri1 - 1] * (double)nz);
    int zq = rad ? zj_from_z_real(T, nz, fabs(z1), ri1) : (int)fmin(qe, (double)nz) + 1;
    zjr = zq > nz ? nz + 1 : zq;
  }
  zjr = (z1 < 0.0) ? -zjr : zjr;
  const int zj_rad = (ri1 == 0) ? 1 : ((ri1 > n_rad) ? zj0 : zjr);
  int zj1 = rad ? zj_rad : (vert ? zj0 + delta_zj : zjr);
  int k_phi = k0 + (ccw ? 1 : -1);
  k_phi = (k_phi == 0) ? n_az : k_phi;
  k_phi = (k_phi == n_az + 1) ? 1 : k_phi;
  int k1 = (rad || vert) ? k0 : k_phi;
  k1 = (rad && ri1 == 0) ? 1 : k1;
  if (__builtin_expect(go && rad && hole, 0)) {  // out of the central hole: the azimuth of the landing point (:1121-1126)
    const double x1h = x0 + dv * u, y1h = y0 + dv * v;
    k1 = az_sector(x1h, y1h, n_az, true);
  }
  const bool snap = vert && (M.midplane_snap != 0) && (delta_zj == 2 || delta_zj == -2);
  z1 = (snap || z1 == 0.0) ? copysign(GRID_PREC, w) : z1;

  // 5) optical depth of the crossing, stop or go on (optical_depth.f90:102, 134-146)
  const double opacity = p.kap * p.kf;  // (p.kf = 0 outside the real cells, see fly_step_2d)
  const double tau = l * opacity;
  const bool stop = go && (tau > p.extr);
  double lc = l;
  if (__builtin_expect(stop, 0)) lc = l * (p.extr / tau);  // (once per flight)
  // save_radiation_field (radiation_field.f90:53)
  if (go && real_cell && !MCGPU_DIAG(A.flags, 1)) {
    if (BIN) { dep_ic = ic; dep_v = p.kab * lc * p.S0; }
    else deposit<LDSE>(A.E_abs, E_lds, ic, p.kab * lc * p.S0);
  }

  // the next cell; DARK: mirrored back at the wall of a dark cell (see roles_cross)
  const int azj1 = zj1 < 0 ? -zj1 : zj1;
  // (k1 >= 1: a lane that is not flying may carry the DEFER request, a negative k -- its index must not leave the array)
  const bool next_real = (ri1 >= 1) && (ri1 <= n_rad) && (azj1 >= 1) && (azj1 <= nz) && (k1 >= 1);
  const int jj1 = zj1 < 0 ? zj1 + nz : zj1 + nz - 1;
  const int ic1 = next_real ? (ri1 - 1) + n_rad * (jj1 + 2 * nz * (k1 - 1)) : M.n_cells;  // (n_cells: the entry of "no cell", 0)
  bool mirror = false;
  if (DARK) mirror = go && !stop && next_real && M.dark[next_real ? ic1 : 0];
  const bool move = go && !stop && !mirror;
  // (VAR: the pair (kappa kappa_factor, kappa_abs_LTE) of the next cell and this wavelength, see var_cell_opacities)
  const double2 kk1 = VAR ? M.v_kk[(size_t)ic1 * M.n_lambda + (p.lam - 1)] : make_double2(M.kappa_factor[ic1], 0.0);
  const double kf1 = kk1.x;

  // 6) commit (x, y: one multiply-add with the length that applies, as in fly_step_2d)
  const double lf = stop ? lc : (move ? dv : 0.0);
  p.x = x0 + lf * u;
  p.y = y0 + lf * v;
  p.z = move ? z1 : z0 + lf * w;
  p.extr = p.extr - tau;
  p.ri = move ? ri1 : ri0;
  p.zj = move ? zj1 : zj0;
  p.k = move ? k1 : k0;
  p.ic = move ? ic1 : ic;
  p.kf = move ? kf1 : p.kf;
  if (VAR) p.kab = move ? kk1.y : p.kab;
  if (DEFER) {
    p.k = stop ? -p.k : p.k;   // (:140: the interaction re-indexes the stopping point, see reindex_stop)
  } else if (__builtin_expect(stop, 0)) {  // (:140: 3D re-indexes the stopping point)
    index_cell<true>(T, M, p.x, p.y, p.z, p.ri, p.zj, p.k, p.ri);
    p.ic = is_real_cell<true>(n_rad, nz, p.ri, p.zj) ? cell_index<true>(n_rad, nz, p.ri, p.zj, p.k) : M.n_cells;
    // (the re-indexed point may lie in a neighbouring cell -- zj goes through default real --: the next flight, which
    // skips the load when the cell is the one it holds, must not inherit the old cell's factor)
    if (VAR) { const double2 kk = M.v_kk[(size_t)p.ic * M.n_lambda + (p.lam - 1)]; p.kf = kk.x; p.kab = kk.y; }
    else p.kf = M.kappa_factor[p.ic];
  }
  if (DARK) {
    p.u = mirror ? -u : u; p.v = mirror ? -v : v; p.w = mirror ? -w : w;
    c_dark += mirror ? 1u : 0u;
  }
  int st = p.st;
  st = (active && out) ? S_EXITED : st;
  st = (active && !out && killed) ? S_EMIT : st;
  st = (stop || mirror) ? S_INTERACT : st;
  c_cross += go ? 1u : 0u;
  c_kill += (active && !out && killed) ? 1u : 0u;
  p.pk_cross += go ? 1u : 0u;
  if (MRW) p.pk_cross |= (move || mirror) ? 0x80000000u : 0u;
  const bool runaway = go && ((MRW ? (p.pk_cross & 0x7FFFFFFFu) : p.pk_cross) > 200000000u);  // a packet that never leaves: flag it, drop it
  if (runaway) { *A.err = 13; st = S_EMIT; }
  p.st = st;
  return ((active && !out && killed) || runaway) ? 1 : 0;
}

// One cell crossing of a packet in flight on a Voronoi grid (the crossing of thermal_body_voro, mc_voronoi.hip.h, on a
// Flight): p.ri = the cell, p.zj = the cell it came from, p.star_key = the cell of the star on the way (0: none).
template <bool CACHE, bool MRW = false>
__device__ inline int voro_roles_cross(const Lds& T, const DevModel& M, const RunArgs& A, const VoroGrid& G,
                                       const DepCache& DC, Flight& p, unsigned int& c_cross, unsigned int& c_kill) {
  const int icell = p.ri;
  if (icell < 0) { p.st = S_EXITED; return 0; }                                  // test_exit_grid_Voronoi (:1446)
  if (p.star_key > 0 && icell == p.star_key) { c_kill++; p.st = S_EMIT; return 1; }  // optical_depth.f90:91-97
  const VoroCell C = G.cell[icell - 1];
  const double opacity = p.kap * C.kf;
  double x1, y1, z1, l, l_contrib, l_void;
  int next;
  voro_cross_cell(G, M, C, p.x, p.y, p.z, p.u, p.v, p.w, icell, p.zj, x1, y1, z1, next, l, l_contrib, l_void);
  c_cross++;
  const double tau = l_contrib * opacity;
  const bool stop = tau > p.extr;
  const double lc = stop ? l_contrib * (p.extr / tau) : l_contrib;
  const double dE = p.kab * lc * p.S0;
  if (dE != 0.0 && !MCGPU_DIAG(A.flags, 1)) {
    if (!(CACHE && DC.add(icell, dE))) atomic_add_f64(&A.E_abs[icell - 1], dE);
  }
  if (stop) {
    const double ls = l_void + lc;
    p.x = nd_add(p.x, nd_mul(ls, p.u));
    p.y = nd_add(p.y, nd_mul(ls, p.v));
    p.z = nd_add(p.z, nd_mul(ls, p.w));
    p.st = S_INTERACT;
  } else {
    p.extr = p.extr - tau;
    p.x = x1; p.y = y1; p.z = z1;
    p.zj = icell;
    p.ri = next;
    if (MRW) p.pk_cross |= 0x80000000u;   // (the flight has left the cell it started in, see fly_step_2d)
  }
  ++p.pk_cross;
  if ((MRW ? (p.pk_cross & 0x7FFFFFFFu) : p.pk_cross) > 200000000u) { *A.err = 13; p.st = S_EMIT; return 1; }  // a packet that never leaves: flag it, drop it
  return 0;
}

// the compiler must not carry values from one serving phase to the next in registers: they go through the record
#ifndef MCGPU_LANE_EMULATION
#define RQ_PHASE_END() asm volatile("" ::: "memory")
#else
#define RQ_PHASE_END()
#endif

// VORO: the same schedule on a Voronoi grid (G; L3D = true, DARK = LDSE = MRW = false): a record's ri is the packet's
// cell, zj the cell it came from, star_key the cell of the star on its way; deposits go through the workgroup's
// deposit cache (DepCache, 2^cache_log_ns slots behind the tables) instead of a private grid.
// BIN: deposits go through the workgroup's staging buckets to the log in HBM (mc_binned.hip.h; grids whose
// absorbed-energy array does not fit in LDS), the staging area sits between the tables and the queues.
// CARRY: the kernel can end early and hand its unfinished packets over (RunArgs::carry_out; "Chunks without tails"
// above, and the tail kernel, mc_tail.hip.h); BIN implies it.
// VAR: lvariable_dust -- the flights read the per-cell opacities (var_cell_opacities), the interactions the tables of the
// cell's class (class_tables); cylindrical grids, no MRW.
template <bool L3D, bool POLA, bool DARK, bool LDSE, bool MRW = false, bool VORO = false, bool BIN = false, bool CARRY = BIN, bool VAR = false,
          bool PARAM = false>   // PARAM: the flying waves cross with fly_step_2d_param (2D, no dark zone, no walk, one dust class)
__device__ __forceinline__ void roles_body(const DevModel& M, const RunArgs& A, double* lds_base, int n_rec, int n_srv_pref,
                                           int k_short, int fly_iters, int fly_idle, int emit_qmax,
                                           const VoroGrid* Gp = nullptr, int cache_log_ns = 0) {
  static_assert(!VORO || (L3D && !DARK && !LDSE), "Voronoi variant");
  static_assert(!BIN || (!LDSE && !VORO), "binned deposits: grids that do not fit in LDS");
  static_assert(!(L3D && MRW) || BIN || VORO, "the walk on 3D grids: the binned role kernel");
  static_assert(!BIN || CARRY, "the chunks of a binned run hand their packets on");
  static_assert(!CARRY || !VORO, "no carry-over on Voronoi grids");
  static_assert(!VAR || (!MRW && !VORO && !BIN && !CARRY), "variable dust: the plain role kernel on cylindrical grids");
  double* const E_lds = lds_base;
  const Lds T = lds_carve(lds_base + (LDSE ? M.n_cells : 0), M);
  lds_stage(T, M);
  DepCache DC;
  DC.log_ns = cache_log_ns;
  DC.val = lds_base + (lds_bytes(M) + 7) / 8;
  DC.tag = reinterpret_cast<int*>(DC.val + ((size_t)1 << cache_log_ns));
  const size_t cache_doubles = VORO ? (((size_t)12 << cache_log_ns) + 7) / 8 : 0;
  if (VORO)
    for (int i = threadIdx.x; i < (1 << cache_log_ns); i += blockDim.x) { DC.val[i] = 0.0; DC.tag[i] = 0; }
  double* const bin_base = lds_base + (LDSE ? M.n_cells : 0) + (lds_bytes(M) + 7) / 8 + cache_doubles;
  const size_t bin_doubles = BIN ? (bin_lds_bytes(A.bin.n_buckets) + 7) / 8 : 0;
  const BinStage BS = bin_carve(bin_base, BIN ? A.bin.n_buckets : 0);
  if (BIN) bin_init(BS, A.bin.n_buckets);
  char* const qbase = reinterpret_cast<char*>(bin_base + bin_doubles);
  RqCtl* const Q = reinterpret_cast<RqCtl*>(qbase);
  unsigned int* const rings = reinterpret_cast<unsigned int*>(qbase + sizeof(RqCtl));
  Rec<POLA>* const recs = reinterpret_cast<Rec<POLA>*>(qbase + sizeof(RqCtl) + 3 * RQ_CAP * sizeof(unsigned int));
  if (LDSE)
    for (int i = threadIdx.x; i < M.n_cells; i += blockDim.x) E_lds[i] = 0.0;
  for (int i = threadIdx.x; i < 3 * RQ_CAP; i += blockDim.x)  // every record starts on the FREE ring
    rings[i] = (i < n_rec) ? ((((unsigned int)i + 1u) << 16) | (unsigned int)i) : 0u;
  if (threadIdx.x == 0) {
    Q->n_pending = 0; Q->ids_done = 0; Q->abort_flag = 0; Q->suspend = 0;
    Q->n_srv = n_srv_pref > 0 ? (n_srv_pref < 1000 ? n_srv_pref : n_srv_pref - 1000) : 0; Q->cooldown = 0; Q->idle_f = 0; Q->idle_s = 0; Q->beat = 0;
    Q->head[0] = Q->head[1] = Q->head[2] = 0u;
    Q->tail[RQ_FREE] = (unsigned int)n_rec; Q->tail[RQ_FLY] = 0u; Q->tail[RQ_SRV] = 0u;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_rad = M.n_rad, nz = M.nz;
  // n_srv_pref >= 1000: that many (minus 1000) serving waves, fixed; otherwise the starting value of the adaptive count
  const bool adapt = n_srv_pref > 0 && n_srv_pref < 1000;
  const int n_waves_wg = (int)(blockDim.x >> 6);
#ifdef MCGPU_TUNING  // parameters of the adaptation from A.flags (bits 8..): window, ratio, margin / 16 (sweeps)
  const int ad_cd = ((A.flags >> 8) & 0xFF) ? ((A.flags >> 8) & 0xFF) : 2;
  const int ad_ratio = ((A.flags >> 16) & 0xFF) ? ((A.flags >> 16) & 0xFF) : 2;
  const int ad_margin = ((A.flags >> 24) & 0x7F) ? 16 * ((A.flags >> 24) & 0x7F) : 256;
#else
  const int ad_cd = 2, ad_ratio = 2, ad_margin = 256;
#endif
  const int free_reserve = n_rec / 8 < 32 ? n_rec / 8 : 32;
  const uint32_t key0 = (uint32_t)A.seed, key1 = (uint32_t)(A.seed >> 32);
  // work items of this launch: [0, n_carry) the records the last chunk left unfinished, then the new packets
  const unsigned long long n_carry = (BIN && A.carry_in_n) ? (unsigned long long)*A.carry_in_n : 0ull;  // (chunks only)
  const unsigned long long n_items = n_carry + A.n_packets;
  const Rec<POLA>* const carry_in = reinterpret_cast<const Rec<POLA>*>(A.carry_in);
  Rec<POLA>* const carry_out = reinterpret_cast<Rec<POLA>*>(A.carry_out);
  bool suspended = false;

  // ---- per-lane state ------------------------------------------------------------------------------------
  // serving: rid >= 0 names the record this lane owns, st its packet's state (S_INTERACT between rounds)
  // flying:  rid < 0 and F, bag_* hold a packet (st = S_FLIGHT, or S_INTERACT / S_EXITED once it stopped); st = S_EMIT: empty
  int rid = -1, st = S_EMIT;
  Flight F;
  flight_clear(F);
  BinLane BP;  // (BIN) this lane's last deposit, see bin_deposit
  bin_lane_init(BP);
  double bag_S1 = 0.0, bag_S2 = 0.0, bag_S3 = 0.0;     // what a packet carries but a flight does not use
  uint32_t bag_plo = 0, bag_phi = 0, bag_event = 0;
  int bag_lambda = 1, bag_fl = 0;
  float bag_tau = 0.0f;

  unsigned int c_cross = 0, c_flight = 0, c_scatt = 0, c_abs = 0, c_esc = 0, c_kill = 0, c_pack = 0, c_dark = 0;
  unsigned int c_walks = 0, c_steps = 0;  // MRW
  unsigned int ev_max = 0;  // the longest packet this lane binned: crossings + interactions (slot 10 of the counters, a maximum)
  unsigned long long pk_next = 0, pk_end = 0;
  bool no_more_ids = false;  // wave-uniform: the global id counter is exhausted
  // consecutive rounds in which neither this wave nor any other wave of the workgroup had work (Q->beat stands
  // still): bounded, a lost packet must not hang the GPU -- while the long tail of a thick model, in which a few
  // lanes work for seconds and everybody else waits, must not trip it
  int idle_spins = 0, last_beat = 0;
  RQ_DIAG(unsigned int d_fly_iters = 0, d_srv_iters = 0, d_fly_cross = 0, d_srv_rounds = 0, d_fly_rounds = 0;)
  RQ_DIAG(unsigned int d_srv_lanes = 0, d_srv_int = 0, d_emit = 0;)
  RQ_DIAG(unsigned int d_emit_rounds = 0, d_exit_rounds = 0, d_exit = 0, d_int_rounds = 0, d_first_rounds = 0, d_first = 0;)

  for (int ep = 0;; ++ep) {
    const int stop_flag = rq_ld(&Q->abort_flag);   // 1: error, 2 (CARRY): hand the packets over
    if (stop_flag == 1) break;
    if (CARRY && carry_out && wave == 0 && lane == 0 && stop_flag == 0) {
      // Wave 0 -- the one wave that always serves -- decides when the workgroup hands its packets over: the global work
      // counter has run out (seen by an emission, or looked up every 16 rounds: a workgroup whose records are full of
      // long flights emits nothing for a while) and (tail_threshold > 0: the launch's last packets go to the tail
      // kernel) no more than that many packets are left here; tail_threshold <= 0: at once (a chunk of a binned run).
      if ((ep & 15) == 15 && !rq_ld(&Q->ids_done) &&
          __hip_atomic_load(A.next_packet, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_items) rq_st(&Q->ids_done, 1);
      if (rq_ld(&Q->ids_done) && (A.tail_threshold <= 0 || rq_ld(&Q->n_pending) <= A.tail_threshold)) atomicCAS(&Q->abort_flag, 0, 2);
    }
    if (CARRY && stop_flag == 2) { suspended = true; break; }  // (the hand-over itself: behind the loop)
    int finished = 0;  // packets this lane finished in this round

    // ---- which role this round -----------------------------------------------------------------------
    const bool any_owned = __ballot(rid >= 0) != 0ull;
    const bool any_held = __ballot(rid < 0 && st != S_EMIT) != 0ull;
    const int fly_n = rq_count(Q, RQ_FLY), srv_n = rq_count(Q, RQ_SRV);
    // (new packets leave free records for the flyers' stopped packets)
    const bool can_emit = !no_more_ids && fly_n <= emit_qmax && rq_count(Q, RQ_FREE) > free_reserve;
    // Which waves serve: waves [0, n_srv).  The share of serving work depends on the model (Pascucci: 0.15
    // interactions per packet, ref4.1: 11), so wave 0 adapts n_srv every few of its rounds to what the two kinds of
    // waves could not fill since the last time: lanes of flying waves left without a flight (the servers do not
    // produce flights fast enough: one more server) against lanes of serving waves left without a packet (one fewer).
    const int n_srv_now = rq_ld(&Q->n_srv);
    const bool prefer_server = wave < n_srv_now;
    if (adapt && wave == 0 && lane == 0) {
      const int cd = rq_ld(&Q->cooldown);
      if (cd > 0) rq_st(&Q->cooldown, cd - 1);
      else {
        const int f = atomicExch(&Q->idle_f, 0), sv = atomicExch(&Q->idle_s, 0);
        if (f > ad_ratio * sv + ad_margin && n_srv_now < n_waves_wg - 1) rq_st(&Q->n_srv, n_srv_now + 1);
        else if (sv > ad_ratio * f + ad_margin && n_srv_now > 1) rq_st(&Q->n_srv, n_srv_now - 1);
        rq_st(&Q->cooldown, ad_cd);
      }
    }
    // Liveness: a wave that holds stopped packets in registers cannot serve, and it can only put them down into a
    // free record or a FLY record.  Wave 0 NEVER takes packets into registers (it flies long flights in place,
    // below): whatever the others hold, somebody always empties the SRV ring, which turns waiting packets into FLY
    // records or free records.  A flying wave that becomes a server drains first (no new flights; its flights go
    // to free records as they become available).
    bool serve;
    if (prefer_server) {
      serve = !any_held;
    } else if (any_owned) {
      serve = true;
      // a wave that prefers to fly gives its waiting packets back when long flights pile up
      if (fly_n >= 64) {
        rq_push(Q, rings, RQ_SRV, lane, rid >= 0, rid);
        rid = -1; st = S_EMIT;
        serve = false;
      }
    } else if (any_held) {
      serve = false;
    } else {
      serve = (fly_n == 0) && ((srv_n > 0) || can_emit);
    }

#if defined(__HIP_DEVICE_COMPILE__) && defined(MCGPU_PRIO_SERVE)   // (A/B builds: the issue priority of a wave by its role)
    if (serve) __builtin_amdgcn_s_setprio(MCGPU_PRIO_SERVE); else __builtin_amdgcn_s_setprio(MCGPU_PRIO_FLY);
#endif
    if (!serve) {
      // ======================= FLYING ==================================================================
      // lanes whose packet stopped swap it for a long flight, empty lanes load one
      const bool stopped = (st == S_INTERACT || st == S_EXITED);  // (packets that left the grid are binned by the servers)
      const bool draining = prefer_server;  // on its way to the serving role: takes no new flights
      const int rin = rq_pop(Q, rings, RQ_FLY, lane, !draining && st != S_FLIGHT);
      int rout = -1;  // record that leaves with this lane's stopped packet
      if (rin >= 0) {
        Rec<POLA>& R = recs[rin];
        if (stopped) {
          // swap registers <-> record, field by field: the record then holds the stopped packet
#define RQ_SWAP(reg, fld) do { const auto t_ = (fld); (fld) = (reg); (reg) = t_; } while (0)
          RQ_SWAP(F.x, R.x); RQ_SWAP(F.y, R.y); RQ_SWAP(F.z, R.z); RQ_SWAP(F.u, R.u); RQ_SWAP(F.v, R.v); RQ_SWAP(F.w, R.w);
          RQ_SWAP(F.extr, R.extr); RQ_SWAP(F.S0, R.S[0]);
          if (POLA) { RQ_SWAP(bag_S1, R.S[POLA ? 1 : 0]); RQ_SWAP(bag_S2, R.S[POLA ? 2 : 0]); RQ_SWAP(bag_S3, R.S[POLA ? 3 : 0]); }
          RQ_SWAP(F.ri, R.ri); RQ_SWAP(F.zj, R.zj); RQ_SWAP(F.k, R.k); RQ_SWAP(bag_lambda, R.lambda); RQ_SWAP(F.star_key, R.star_key);
          RQ_SWAP(bag_plo, R.p_lo); RQ_SWAP(bag_phi, R.p_hi); RQ_SWAP(bag_event, R.event);
          RQ_SWAP(F.pk_cross, R.pk_cross); RQ_SWAP(bag_tau, R.tau_rand);
#undef RQ_SWAP
          const int fl_in = R.flags;
          R.flags = st | bag_fl;
          bag_fl = fl_in & ~ST_MASK;
          rout = rin;
        } else {
          F.x = R.x; F.y = R.y; F.z = R.z; F.u = R.u; F.v = R.v; F.w = R.w; F.extr = R.extr; F.S0 = R.S[0];
          if (POLA) { bag_S1 = R.S[POLA ? 1 : 0]; bag_S2 = R.S[POLA ? 2 : 0]; bag_S3 = R.S[POLA ? 3 : 0]; }
          F.ri = R.ri; F.zj = R.zj; F.k = R.k; bag_lambda = R.lambda; F.star_key = R.star_key;
          bag_plo = R.p_lo; bag_phi = R.p_hi; bag_event = R.event; F.pk_cross = R.pk_cross; bag_tau = R.tau_rand;
          bag_fl = R.flags & ~ST_MASK;
        }
        st = S_FLIGHT;
        if (VORO) { F.kap = T.kappa[bag_lambda - 1]; F.kab = T.kabs[bag_lambda - 1]; }
        else flight_constants<L3D, VAR>(T, M, F, bag_lambda);
      }
      // stopped packets that found no flight to swap with go to a free record
      {
        // (a draining wave also puts its flights down: they go back to the FLY ring)
        const int rf = rq_pop(Q, rings, RQ_FREE, lane, st == S_INTERACT || st == S_EXITED || (draining && st == S_FLIGHT));
        if (rf >= 0) {
          Rec<POLA>& R = recs[rf];
          R.x = F.x; R.y = F.y; R.z = F.z; R.u = F.u; R.v = F.v; R.w = F.w; R.extr = F.extr; R.S[0] = F.S0;
          if (POLA) { R.S[POLA ? 1 : 0] = bag_S1; R.S[POLA ? 2 : 0] = bag_S2; R.S[POLA ? 3 : 0] = bag_S3; }
          R.ri = F.ri; R.zj = F.zj; R.k = F.k; R.lambda = bag_lambda; R.star_key = F.star_key;
          R.p_lo = bag_plo; R.p_hi = bag_phi; R.event = bag_event; R.pk_cross = F.pk_cross; R.tau_rand = bag_tau;
          R.flags = st | bag_fl;
          rout = rf;
          st = S_EMIT;
        }
      }
      {
        const bool was_flight = rout >= 0 && (recs[rout].flags & ST_MASK) == S_FLIGHT;
        rq_push(Q, rings, RQ_SRV, lane, rout >= 0 && !was_flight, rout);
        rq_push(Q, rings, RQ_FLY, lane, rout >= 0 && was_flight, rout);
      }
      rq_push(Q, rings, RQ_FREE, lane, rin >= 0 && rout != rin, rin);  // loaded, not swapped: the record is free again

      if (adapt && !draining) {
        const int n_idle = __popcll(__ballot(st != S_FLIGHT));
        if (n_idle > 0 && lane == 0) atomicAdd(&Q->idle_f, n_idle);
      }
      if (__ballot(st == S_FLIGHT) == 0ull) {
        if (__ballot(st != S_EMIT) == 0ull && rq_ld(&Q->ids_done) && rq_ld(&Q->n_pending) == 0) break;
        __builtin_amdgcn_s_sleep(16);  // nothing to fly (or no record for a stopped packet): wait for the others
        { const int b = rq_ld(&Q->beat); if (b != last_beat) { last_beat = b; idle_spins = 0; } }
        if (++idle_spins > (1 << 21)) { *A.err = 15; rq_st(&Q->abort_flag, 1); }  // (seconds without work anywhere in the workgroup: a lost packet)
      } else {
        idle_spins = 0;
        if (lane == 0) atomicAdd(&Q->beat, 1);
        RQ_DIAG(if (lane == 0) d_fly_rounds++;)
        F.st = st;
        FlightParam FP;
        unsigned int c_cross_wave = 0u;
        bool any_star = true;
        if (PARAM) { param_begin(F, FP); any_star = __ballot(st == S_FLIGHT && F.star_key >= 0) != 0ull; }
#pragma unroll 1
        for (int it = 0; it < fly_iters; ++it) {
          // back to the rings as soon as enough lanes have nothing to fly (or after fly_iters crossings)
          if (it > 0 && __popcll(__ballot(F.st != S_FLIGHT)) >= fly_idle) break;
          RQ_DIAG(if (lane == 0) d_fly_iters++; if (F.st == S_FLIGHT) d_fly_cross++;)
          if (PARAM) {
            finished += fly_step_2d_param<LDSE>(T, M, A, E_lds, F, FP, c_cross_wave, c_kill, any_star);
          } else if (VORO) {
            if (F.st == S_FLIGHT) finished += voro_roles_cross<true, MRW>(T, M, A, *Gp, DC, F, c_cross, c_kill);
          } else if (L3D) {
            int dep_ic = -1;
            double dep_v = 0.0;
            if (MCGPU_3D_BRANCHY) { if (F.st == S_FLIGHT) finished += roles_cross<L3D, DARK, LDSE, BIN>(T, M, A, E_lds, F, c_cross, c_kill, c_dark, dep_ic, dep_v); }
            else finished += fly_step_3d<DARK, LDSE, BIN, VAR, true, MRW>(T, M, A, E_lds, F, c_cross, c_kill, c_dark, dep_ic, dep_v);
            if (BIN) bin_deposit(BS, A.bin, A.E_abs, lane, BP, dep_ic >= 0, dep_ic, dep_v);
          } else {
            finished += fly_step_2d<DARK, LDSE, MRW, false, VAR>(T, M, A, E_lds, F, c_cross, c_kill, c_dark);
          }
        }
        if (BIN) bin_settle(BS, A.bin, A.E_abs, lane, BP);
        if (PARAM) {
          param_end(F, FP);
          if (lane == 0) c_cross += c_cross_wave;
          if (F.pk_cross > 200000000u && F.st == S_FLIGHT) { *A.err = 13; F.st = S_EMIT; finished += 1; }  // a packet that never leaves: flag it, drop it
        }
        st = F.st;
      }
    } else {
      // ======================= SERVING =================================================================
      RQ_DIAG(if (lane == 0) d_srv_rounds++;)
      // ---- lanes without a record take a packet that waits for its interaction ... ----------------------
      if (__ballot(rid < 0) != 0ull) {
        const int r = rq_pop(Q, rings, RQ_SRV, lane, rid < 0);
        if (r >= 0) { rid = r; st = recs[r].flags & ST_MASK; }  // (S_INTERACT, or S_EXITED: to be binned)
      }
      // ---- ... or a free record for a new packet (mc_photon_loop body, dust_transfer.f90:529-541) -------
      {
        const unsigned long long mask = __ballot(rid < 0);
        if (mask && can_emit) {
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
            pk_next = base < n_items ? base : n_items;
            pk_end = (base + PK_BATCH < n_items) ? base + PK_BATCH : n_items;
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
          const int r = rq_pop(Q, rings, RQ_FREE, lane, rid < 0 && rank < avail);
          // ids go to the lanes that got a record, in lane order
          const unsigned long long got = __ballot(r >= 0);
          const unsigned long long my = pk_next + (unsigned long long)__popcll(got & ((1ull << lane) - 1ull));
          pk_next += (unsigned long long)__popcll(got);
          // a carried record (see "Chunks without tails"): it goes on where the last chunk left it; one that was
          // never emitted (state S_EMIT) is emitted now, with the id it carries
          unsigned long long pid = A.first_packet + (my - n_carry);
          bool fresh = r >= 0;
          if (BIN && r >= 0 && my < n_carry) {
            const Rec<POLA>& Rc = carry_in[my];
            const int fl_c = Rc.flags;
            if ((fl_c & ST_MASK) == S_EMIT) pid = ((unsigned long long)Rc.p_hi << 32) | Rc.p_lo;
            else { rec_copy(&recs[r], &Rc); rid = r; st = fl_c & ST_MASK; fresh = false; }
          }
          RQ_DIAG(if (lane == 0 && __ballot(fresh) != 0ull) d_emit_rounds++;)
          if (fresh) {
            rid = r;
            Rec<POLA>& R = recs[rid];
            Rng rng;
            rng.init(A.seed, pid);
            c_pack++;
            RQ_DIAG(d_emit++;)
            float f[12];
            rng.emission_event(f);
            const int lambda = select_wl_em(T, M, f[0]);
            lds_count_sent(T, lambda);
            bool lintersect, flag_star, flag_ism;
            double x, y, z, u, v, w;
            int ri = 0, zj = 1, k = 1;
            int rc;
            if (VORO) {
              zj = 0;
              VoroEmitOps ops{*Gp, M, ri};
              rc = emit_packet(M, f, lambda, T.fstar[lambda - 1], M.frac_E_disk[lambda - 1],
                               M.prob_E_cell ? M.prob_E_cell + (size_t)(M.n_cells + 1) * (lambda - 1) : nullptr,
                               ops, x, y, z, u, v, w, flag_star, flag_ism, lintersect);
            } else {
              CylEmitOps<L3D> ops{T, M, ri, zj, k};
              rc = emit_packet(M, f, lambda, T.fstar[lambda - 1], M.frac_E_disk[lambda - 1],
                               M.prob_E_cell ? M.prob_E_cell + (size_t)(M.n_cells + 1) * (lambda - 1) : nullptr,
                               ops, x, y, z, u, v, w, flag_star, flag_ism, lintersect);
            }
            if (rc) { *A.err = rc; rq_st(&Q->abort_flag, 1); }
            st = lintersect ? S_NEWFLIGHT : S_EXITED;
            R.x = x; R.y = y; R.z = z; R.u = u; R.v = v; R.w = w; R.extr = 0.0;
            R.S[0] = 1.0;
            if (POLA) { R.S[POLA ? 1 : 0] = 0.0; R.S[POLA ? 2 : 0] = 0.0; R.S[POLA ? 3 : 0] = 0.0; }
            R.ri = ri; R.zj = zj; R.k = k; R.lambda = lambda; R.star_key = -1;
            R.p_lo = rng.p_lo; R.p_hi = rng.p_hi; R.event = rng.event;
            R.flags = st | (flag_star ? ST_STAR : 0) | (flag_ism ? ST_ISM : 0);
            R.pk_cross = 0u;
            R.tau_rand = f[8];
          }
        }
      }
      // ---- ... or, when the wave has little to serve, a long flight that it flies in place (the record stays the
      // packet's home: loaded, crossed fly_iters times, stored back) ------------------------------------------
      if (adapt && prefer_server) {
        const int n_idle = __popcll(__ballot(rid < 0));
        if (n_idle > 0 && lane == 0) atomicAdd(&Q->idle_s, n_idle);
      }
      bool flying_in_place = false;
      if (prefer_server && __popcll(__ballot(rid >= 0)) < 32) {
        const int r = rq_pop(Q, rings, RQ_FLY, lane, rid < 0);
        if (r >= 0) { rid = r; st = S_FLIGHT; }
        flying_in_place = __ballot(r >= 0) != 0ull;
      }
      RQ_PHASE_END();
      RQ_DIAG(if (rid >= 0) d_srv_lanes++; if (st == S_INTERACT && rid >= 0) d_srv_int++;)

      if (__ballot(rid >= 0) == 0ull) {  // the wave owns no packet at all
        if (rq_ld(&Q->ids_done) && rq_ld(&Q->n_pending) == 0) break;
        __builtin_amdgcn_s_sleep(16);
        { const int b = rq_ld(&Q->beat); if (b != last_beat) { last_beat = b; idle_spins = 0; } }
        if (++idle_spins > (1 << 21)) { *A.err = 15; rq_st(&Q->abort_flag, 1); }
      } else {
        idle_spins = 0;
        if (lane == 0) atomicAdd(&Q->beat, 1);
        // ---- INTERACT: scatter or absorb + re-emit (dust_transfer.f90:1260-1402), in two phases: the event and
        // the new direction, then (Stokes tracking) the Stokes vector.  Only the new direction, the scattering
        // angle bin and one draw cross the phase boundary in registers.
        double u1 = 0.0, v1 = 0.0, w1 = 1.0;
        int itheta = 1, lambda_sc = 1, vcls = -1, igrain = 0;  // (VAR: the cell's class, the grain of scattering method 1)
        float rand2 = 0.0f;
        bool scat = false;
        const bool inter = rid >= 0 && st == S_INTERACT;
        RQ_DIAG(if (lane == 0 && __ballot(inter) != 0ull) d_int_rounds++;)
        if (inter) {
          Rec<POLA>& R = recs[rid];
          Rng rng;
          rng.k0 = key0; rng.k1 = key1; rng.p_lo = R.p_lo; rng.p_hi = R.p_hi; rng.event = R.event;
          float g[8];
          rng.interaction_event(g, M.m1 != 0);
          int lambda = R.lambda;
          lambda_sc = lambda;
          const int fl = R.flags;
          bool flag_star = (fl & ST_STAR) != 0, flag_scatt = (fl & ST_SCATT) != 0, flag_ism = (fl & ST_ISM) != 0;
          if (L3D && !VORO && R.k < 0) {   // a flight stopped here: the cell of the stopping point (fly_step_3d, DEFER)
            int ri_s, zj_s, k_s;
            index_cell<true>(T, M, R.x, R.y, R.z, ri_s, zj_s, k_s, R.ri);
            R.ri = ri_s; R.zj = zj_s; R.k = k_s;
          }
          const int ic = VORO ? R.ri - 1 : cell_index<L3D>(n_rad, nz, R.ri, R.zj, R.k);
          const int cls = VAR ? M.cell_class[ic] : -1;
          const Lds Tc = VAR ? class_tables(T, M, cls) : T;   // (lvariable_dust: this cell's tables)
          vcls = cls;
          scat = interact_direction(Tc, M, g, lambda, R.u, R.v, R.w, u1, v1, w1, flag_star, flag_scatt, c_scatt, c_abs, [&]() {
            // the cell's absorbed energy for Temp_LTE (thermal_emission.f90:649-706): what every workgroup has
            // folded into HBM so far plus (LDSE) this workgroup's not yet folded part -- the other workgroups'
            // unfolded parts are estimated by this one's, exactly the reference's partial * nb_proc
            // (thermal_emission.f90:670) with workgroups in the role of threads; * n_replicas across GPUs
            double E;
            if (A.frozen) E = A.E_prior[ic];
            else {
              E = __hip_atomic_load(&A.E_abs[ic], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (LDSE) E += E_lds[ic] * (double)gridDim.x;
              if (VORO) E += DC.pending(ic + 1) * (double)gridDim.x;
              if (BIN) E *= bin_energy_scale(A);
              E *= A.qscale;
            }
            return E;
          }, M.volume + ic, itheta, rand2, false, nullptr, -1, (VAR && M.m1) ? cls : -1, &igrain);
          if (!flag_scatt) flag_ism = false;  // absorbed and re-emitted by the dust (:1367)
          R.lambda = lambda;
          R.event = rng.event;
          R.tau_rand = g[5];
          int n_int = 0;
          if (MRW) {  // interactions in a row whose flights never left the cell (dust_transfer.f90:1244-1249), 0..7
            n_int = (fl >> ST_NINT_SHIFT) & 7;
            n_int = (R.pk_cross & 0x80000000u) ? 0 : (n_int < 7 ? n_int + 1 : 7);
            R.pk_cross &= 0x7FFFFFFFu;  // the next flight starts here
          }
          R.flags = S_NEWFLIGHT | (flag_star ? ST_STAR : 0) | (flag_scatt ? ST_SCATT : 0) | (flag_ism ? ST_ISM : 0) | (n_int << ST_NINT_SHIFT);
          if (!POLA) { R.u = u1; R.v = v1; R.w = w1; }
        }
        if (POLA) {
          RQ_PHASE_END();
          if (inter) {
            Rec<POLA>& R = recs[rid];
            double S[4] = {R.S[0], R.S[POLA ? 1 : 0], R.S[POLA ? 2 : 0], R.S[POLA ? 3 : 0]};
            interact_stokes(M, scat, lambda_sc, itheta, rand2, R.u, R.v, R.w, u1, v1, w1, S, (VAR && M.v_scatt) ? vcls : -1, igrain);
            R.S[0] = S[0]; R.S[POLA ? 1 : 0] = S[1]; R.S[POLA ? 2 : 0] = S[2]; R.S[POLA ? 3 : 0] = S[3];
            R.u = u1; R.v = v1; R.w = w1;
          }
        }
        if (inter) st = S_NEWFLIGHT;
        RQ_PHASE_END();
        if (MRW) {
          // ---- modified random walk of the packets their cell has just re-emitted for the (n_inter+1)-th time in
          // a row (dust_transfer.f90:1222-1239; mrw_walk in mc_device.hip.h) --------------------------------------
          bool walk = false;
          if (inter) {
            const int fl = recs[rid].flags;
            walk = !(fl & (ST_SCATT | ST_STAR)) && ((fl >> ST_NINT_SHIFT) & 7) > M.mrw_n_inter;
          }
          // BIN: a walk stays in its cell, so its deposits (two or three steps) are summed here and logged as ONE deposit
          // by the whole wave below (bin_deposit wants converged control flow); the walk's own temperature sees them
          int walk_ic = -1;
          double walk_dep = 0.0;
          if (__builtin_expect(walk, 0)) {  // (rare: the hint keeps its registers out of the common path, +5 % with no walks)
            Rec<POLA>& R = recs[rid];
            const int ic = VORO ? R.ri - 1 : cell_index<L3D>(n_rad, nz, R.ri, R.zj, R.k);
            double x = R.x, y = R.y, z = R.z, u = R.u, v = R.v, w = R.w;
            int lambda = R.lambda;
            bool done;
            if (VORO) {   // (distance_to_closest_wall_Voronoi, the deposit cache: as k_thermal_voro_mrw, mc_voronoi.hip.h)
              const VoroCell C = Gp->cell[ic];
              done = mrw_walk_with(T, M, key0, key1, R.p_lo, R.p_hi, R.event, ic, C.kf, R.S[0], x, y, z, u, v, w, lambda,
                [&](double px, double py, double pz) { return voro_distance_to_closest_wall(*Gp, C, px, py, pz); },
                [&]() {
                  if (A.frozen) return A.E_prior[ic];
                  double E = __hip_atomic_load(&A.E_abs[ic], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  E += DC.pending(ic + 1) * (double)gridDim.x;
                  return E * A.qscale;
                },
                [&](double e) { if (!DC.add(ic + 1, e)) atomic_add_f64(&A.E_abs[ic], e); }, c_walks, c_steps);
            } else
            done = mrw_walk(T, M, key0, key1, R.p_lo, R.p_hi, R.event, R.ri, R.zj, ic, R.S[0], x, y, z, u, v, w, lambda,
              [&]() {
                double E;
                if (A.frozen) E = A.E_prior[ic];
                else {
                  E = __hip_atomic_load(&A.E_abs[ic], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  if (LDSE) E += E_lds[ic] * (double)gridDim.x;
                  if (BIN) E = E * bin_energy_scale(A) + walk_dep;
                  E *= A.qscale;
                }
                return E;
              },
              [&](double e) { if (BIN) walk_dep += e; else deposit<LDSE>(A.E_abs, E_lds, ic, e); }, c_walks, c_steps, L3D ? R.k : 1);
            if (done) { R.x = x; R.y = y; R.z = z; R.u = u; R.v = v; R.w = w; R.lambda = lambda; }
            walk_ic = ic;
          }
          if (BIN) {
            bin_deposit(BS, A.bin, A.E_abs, lane, BP, walk_ic >= 0 && walk_dep != 0.0, walk_ic, walk_dep);
            bin_settle(BS, A.bin, A.E_abs, lane, BP);
          }
          RQ_PHASE_END();
        }
        // ---- NEWFLIGHT: optical depth to the next event (dust_transfer.f90:1208-1215, tau in FP64) and the
        // star on the way (optical_depth.f90:68) ---------------------------------------------------------
        if (rid >= 0 && st == S_NEWFLIGHT) {
          Rec<POLA>& R = recs[rid];
          const float rand = R.tau_rand;
          R.extr = tau_of_draw(rand);
          const int i_star = intersect_stars(M, R.x, R.y, R.z, R.u, R.v, R.w);
          int key = -1;
          if (i_star > 0) {
            const int* sc = &M.star_cell[4 * (i_star - 1)];
            key = VORO ? sc[0] : sc[0] + (n_rad + 2) * ((sc[1] + nz + 1) + (2 * nz + 3) * (sc[2] - 1));
          }
          if (VORO) R.zj = 0;  // a new flight has no previous cell (prev_cell = 0)
          R.star_key = key;
          c_flight++;
          st = S_FLIGHT;
        }
        RQ_PHASE_END();
        // ---- the first crossings of every flight ----------------------------------------------------------
        if (__ballot(rid >= 0 && st == S_FLIGHT) != 0ull) {
          const bool fly = rid >= 0 && st == S_FLIGHT;
          RQ_DIAG(if (lane == 0) d_first_rounds++; if (fly) d_first++;)
          Rec<POLA>& R = recs[fly ? rid : 0];
          flight_clear(F);
          F.st = fly ? S_FLIGHT : S_EMIT;
          if (fly) {
            F.x = R.x; F.y = R.y; F.z = R.z; F.u = R.u; F.v = R.v; F.w = R.w; F.extr = R.extr; F.S0 = R.S[0];
            F.ri = R.ri; F.zj = R.zj; F.k = R.k; F.star_key = R.star_key; F.pk_cross = R.pk_cross;
            if (VORO) { F.kap = T.kappa[R.lambda - 1]; F.kab = T.kabs[R.lambda - 1]; }
            else flight_constants<L3D, VAR>(T, M, F, R.lambda);
          }
          const int n_it = flying_in_place ? fly_iters : k_short;
#pragma unroll 1
          for (int it = 0; it < n_it; ++it) {
            if (__ballot(F.st == S_FLIGHT) == 0ull) break;
            RQ_DIAG(if (lane == 0) d_srv_iters++;)
            if (VORO) {
              if (F.st == S_FLIGHT) finished += voro_roles_cross<true, MRW>(T, M, A, *Gp, DC, F, c_cross, c_kill);
            } else if (L3D) {
              int dep_ic = -1;
              double dep_v = 0.0;
              if (MCGPU_3D_BRANCHY) { if (F.st == S_FLIGHT) finished += roles_cross<L3D, DARK, LDSE, BIN>(T, M, A, E_lds, F, c_cross, c_kill, c_dark, dep_ic, dep_v); }
              else finished += fly_step_3d<DARK, LDSE, BIN, VAR, true, MRW>(T, M, A, E_lds, F, c_cross, c_kill, c_dark, dep_ic, dep_v);
              if (BIN) bin_deposit(BS, A.bin, A.E_abs, lane, BP, dep_ic >= 0, dep_ic, dep_v);
            } else {
              finished += fly_step_2d<DARK, LDSE, MRW, false, VAR>(T, M, A, E_lds, F, c_cross, c_kill, c_dark);
            }
          }
          if (BIN) bin_settle(BS, A.bin, A.E_abs, lane, BP);
          if (fly) {
            R.x = F.x; R.y = F.y; R.z = F.z; R.extr = F.extr;
            if (DARK) { R.u = F.u; R.v = F.v; R.w = F.w; }
            R.ri = F.ri; R.zj = F.zj; R.k = F.k; R.pk_cross = F.pk_cross;
            st = F.st;
            R.flags = (R.flags & ~ST_MASK) | st;
          }
        }
        RQ_PHASE_END();
        // ---- packets that left the grid: capteur (output.f90:294-597) --------------------------------------
        RQ_DIAG(if (lane == 0 && __ballot(rid >= 0 && st == S_EXITED) != 0ull) d_exit_rounds++;)
        if (rid >= 0 && st == S_EXITED) {
          RQ_DIAG(d_exit++;)
          const Rec<POLA>& R = recs[rid];
          const int fl = R.flags;
          if (!(fl & ST_ISM) ) {  // ISM packets that were never absorbed are not binned (dust_transfer.f90:549)
            const double S[4] = {R.S[0], POLA ? R.S[POLA ? 1 : 0] : 0.0, POLA ? R.S[POLA ? 2 : 0] : 0.0, POLA ? R.S[POLA ? 3 : 0] : 0.0};
            capteur<POLA>(M, A.sed, R.lambda, R.u, R.v, R.w, S, (fl & ST_STAR) != 0, (fl & ST_SCATT) != 0);
            c_esc++;
          }
          { const unsigned int ev = (R.pk_cross & 0x7FFFFFFFu) + R.event; ev_max = ev > ev_max ? ev : ev_max; }
          st = S_EMIT;
          finished++;
        }
        // ---- long flights to the FLY ring, finished packets' records to the FREE ring ------------------------
        {
          const bool to_fly = rid >= 0 && st == S_FLIGHT, to_free = rid >= 0 && st == S_EMIT;
          rq_push(Q, rings, RQ_FLY, lane, to_fly, rid);
          rq_push(Q, rings, RQ_FREE, lane, to_free, rid);
          if (to_fly || to_free) { rid = -1; st = S_EMIT; }
        }
      }
      // a serving wave holds no packet in registers: say so, so that the flight registers are not kept alive
      // through the serving phases
      flight_clear(F);
      bag_S1 = bag_S2 = bag_S3 = 0.0; bag_plo = bag_phi = bag_event = 0u; bag_lambda = 1; bag_fl = 0; bag_tau = 0.0f;
    }

    // ---- bookkeeping common to both roles ---------------------------------------------------------
    {
      const int fin = __popcll(__ballot(finished > 0)) + __popcll(__ballot(finished > 1));
      if (fin > 0 && lane == 0) atomicAdd(&Q->n_pending, -fin);
    }
    if (VORO && ((ep + 1) % A.flush_every) == 0) {  // barrier-free partial fold of the deposit cache (see thermal_body_voro)
      const int n_waves = (blockDim.x + 63) >> 6;
      const int slice = (wave + (ep + 1) / A.flush_every) % n_waves;
      const int ns = 1 << cache_log_ns, per = (ns + n_waves - 1) / n_waves;
      const int i0 = slice * per, i1 = (i0 + per < ns) ? i0 + per : ns;
      for (int i = i0 + lane; i < i1; i += 64) {
        const int t = DC.tag[i];
        if (t == 0) continue;
        const unsigned long long bits = atomicExch(reinterpret_cast<unsigned long long*>(&DC.val[i]), 0ull);
        const double e = __longlong_as_double((long long)bits);
        if (e != 0.0) atomic_add_f64(&A.E_abs[t - 1], e);
      }
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

  if (CARRY && suspended) {
    // ---- end of a chunk: what this wave holds goes to carry_out (see "Chunks without tails") -----------------
    const bool held = rid < 0 && st != S_EMIT;   // a packet in registers
    const bool owned = rid >= 0;                 // a record of this lane's
    const unsigned long long left = pk_end - pk_next;  // reserved, not started (lane i takes the items pk_next + i + 64 q)
    const int n_left = (int)((left > (unsigned long long)lane) ? (left - lane + (BIN_WAVE - 1)) / BIN_WAVE : 0ull);
    const int mine = (held || owned ? 1 : 0) + n_left;
    int pre = mine;  // inclusive prefix sum over the lanes
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(pre, off); if (lane >= off) pre += t; }
    const int total = __shfl(pre, 63);
    unsigned int base = 0u;
    if (lane == 0 && total > 0) base = atomicAdd(A.carry_out_n, (unsigned int)total);
    base = __shfl(base, 0);
    unsigned int at = base + (unsigned int)(pre - mine);
    if (at + (unsigned int)mine > A.carry_cap) { *A.err = 16; rq_st(&Q->abort_flag, 1); }  // (never: the host sizes it for the worst case)
    else {
      if (held) {
        Rec<POLA>& R = carry_out[at++];
        R.x = F.x; R.y = F.y; R.z = F.z; R.u = F.u; R.v = F.v; R.w = F.w; R.extr = F.extr; R.S[0] = F.S0;
        if (POLA) { R.S[POLA ? 1 : 0] = bag_S1; R.S[POLA ? 2 : 0] = bag_S2; R.S[POLA ? 3 : 0] = bag_S3; }
        R.ri = F.ri; R.zj = F.zj; R.k = F.k; R.lambda = bag_lambda; R.star_key = F.star_key;
        R.p_lo = bag_plo; R.p_hi = bag_phi; R.event = bag_event; R.pk_cross = F.pk_cross; R.tau_rand = bag_tau;
        R.flags = st | bag_fl;
      } else if (owned) {
        recs[rid].flags = (recs[rid].flags & ~ST_MASK) | st;
        rec_copy(&carry_out[at++], &recs[rid]);
      }
      for (int q = 0; q < n_left; ++q) {
        const unsigned long long item = pk_next + (unsigned long long)lane + (unsigned long long)BIN_WAVE * q;
        if (BIN && item < n_carry) rec_copy(&carry_out[at++], &carry_in[item]);
        else {  // a packet that was never emitted: state S_EMIT, its id in p_lo / p_hi
          Rec<POLA>& Rc = carry_out[at++];
          rec_copy(&Rc, static_cast<const Rec<POLA>*>(nullptr));
          const unsigned long long pid = A.first_packet + (item - n_carry);
          Rc.p_lo = (uint32_t)pid; Rc.p_hi = (uint32_t)(pid >> 32); Rc.flags = S_EMIT;
        }
      }
    }
  }
  __syncthreads();
  if (CARRY && carry_out && __syncthreads_or(suspended ? 1 : 0)) {
    // the packets that wait in the FLY and SRV rings (nobody pops or pushes any more)
    for (int q = RQ_FLY; q <= RQ_SRV; ++q) {
      const unsigned int h = Q->head[q], t = Q->tail[q];
      const unsigned int n = t - h;
      if (threadIdx.x == 0) Q->pad1 = n ? (int)atomicAdd(A.carry_out_n, n) : 0;
      __syncthreads();
      const unsigned int base = (unsigned int)Q->pad1;
      if (base + n > A.carry_cap) { if (threadIdx.x == 0) *A.err = 16; }
      else
        for (unsigned int i = threadIdx.x; i < n; i += blockDim.x)
          rec_copy(&carry_out[base + i], &recs[rings[q * RQ_CAP + ((h + i) & (RQ_CAP - 1))] & 0xFFFFu]);
      __syncthreads();
    }
  }
  lds_flush_sent(T, M, A.n_sent);
  if (BIN) bin_drain(BS, A.bin, A.E_abs);
  if (LDSE) {
    for (int i = threadIdx.x; i < M.n_cells; i += blockDim.x) {
      const double e = E_lds[i];
      if (e != 0.0) atomic_add_f64(&A.E_abs[i], e);
    }
  }
  if (VORO) {  // final fold of the deposit cache
    for (int i = threadIdx.x; i < (1 << cache_log_ns); i += blockDim.x) {
      const double e = DC.val[i];
      if (e != 0.0) atomic_add_f64(&A.E_abs[DC.tag[i] - 1], e);
    }
  }
  unsigned int cs[8] = {c_pack, c_cross, c_flight, c_scatt, c_abs, c_esc, c_kill, c_dark};
#ifdef MCGPU_COUNT_ITERS  // the eight counters carry the schedule's statistics instead
  cs[0] = d_srv_lanes; cs[1] = d_srv_int; cs[6] = d_emit; cs[3] = d_fly_cross;  // lanes, summed over rounds
  cs[2] = d_srv_rounds; cs[5] = d_fly_rounds;                                   // rounds
  cs[4] = d_srv_iters; cs[7] = d_fly_iters;                                     // crossing iterations
#if MCGPU_COUNT_ITERS == 2  // ... or the serving phases': rounds in which a phase ran, lanes it ran for
  cs[0] = d_emit_rounds; cs[1] = d_emit; cs[2] = d_exit_rounds; cs[3] = d_exit; cs[4] = d_int_rounds; cs[5] = d_srv_int;
  cs[6] = d_first_rounds; cs[7] = d_first;
#endif
#endif
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    unsigned long long vsum = cs[q];
    for (int off = 32; off > 0; off >>= 1) vsum += __shfl_down(vsum, off);
    if (lane == 0 && vsum) atomicAdd(&A.counters[q], vsum);
  }
  {
    unsigned int vm = ev_max;
    for (int off = 32; off > 0; off >>= 1) { const unsigned int o = __shfl_down(vm, off); vm = o > vm ? o : vm; }
    if (lane == 0 && vm) atomicMax(&A.counters[10], (unsigned long long)vm);
  }
  if (MRW) {
    unsigned long long v8 = c_walks, v9 = c_steps;
    for (int off = 32; off > 0; off >>= 1) { v8 += __shfl_down(v8, off); v9 += __shfl_down(v9, off); }
    if (lane == 0 && v8) atomicAdd(&A.counters[8], v8);
    if (lane == 0 && v9) atomicAdd(&A.counters[9], v9);
  }
}

#ifndef MCGPU_ROLES_BLOCK
#define MCGPU_ROLES_BLOCK 1024  // threads of a workgroup of this schedule: 128 VGPRs, 4 waves per SIMD
#endif

template <bool L3D, bool POLA, bool DARK, bool LDSE, bool MRW = false>
__global__ void __launch_bounds__(MCGPU_ROLES_BLOCK) k_thermal_roles(const DevModel M, const RunArgs A, int n_rec, int n_srv_pref,
                                                                     int k_short, int fly_iters, int fly_idle, int emit_qmax) {
  extern __shared__ double lds_raw[];
  roles_body<L3D, POLA, DARK, LDSE, MRW>(M, A, lds_raw, n_rec, n_srv_pref, k_short, fly_iters, fly_idle, emit_qmax);
}

// the same, handing its last packets to the tail kernel (mc_tail.hip.h); 2D grids (the 3D kernel below has it built in)
// lvariable_dust in the role schedule (the tables gain the cell axis -- the flights gather 16 bytes per
// cell entered, the interactions read the class's rows from HBM)
template <bool L3D, bool POLA, bool DARK, bool LDSE>
__global__ void __launch_bounds__(MCGPU_ROLES_BLOCK) k_thermal_roles_var(const DevModel M, const RunArgs A, int n_rec, int n_srv_pref,
                                                                         int k_short, int fly_iters, int fly_idle, int emit_qmax) {
  extern __shared__ double lds_raw[];
  roles_body<L3D, POLA, DARK, LDSE, false, false, false, false, true>(M, A, lds_raw, n_rec, n_srv_pref, k_short, fly_iters, fly_idle, emit_qmax);
}

template <bool POLA, bool DARK, bool LDSE, bool MRW>
__global__ void __launch_bounds__(MCGPU_ROLES_BLOCK) k_thermal_roles_tail(const DevModel M, const RunArgs A, int n_rec, int n_srv_pref,
                                                                          int k_short, int fly_iters, int fly_idle, int emit_qmax) {
  extern __shared__ double lds_raw[];
  roles_body<false, POLA, DARK, LDSE, MRW, false, false, true>(M, A, lds_raw, n_rec, n_srv_pref, k_short, fly_iters, fly_idle, emit_qmax);
}

// the same with binned deposits (3D grids: the absorbed-energy array does not fit in LDS)
#ifndef MCGPU_ROLES_BIN_BLOCK
#define MCGPU_ROLES_BIN_BLOCK 768  // 168 VGPRs, 3 waves per SIMD: the 3D crossing + the staging do not fit into 128 registers
#endif
template <bool POLA, bool DARK, bool MRW = false>
__global__ void __launch_bounds__(MCGPU_ROLES_BIN_BLOCK) k_thermal_roles_bin(const DevModel M, const RunArgs A, int n_rec, int n_srv_pref,
                                                                         int k_short, int fly_iters, int fly_idle, int emit_qmax) {
  extern __shared__ double lds_raw[];
  roles_body<true, POLA, DARK, false, MRW, false, true>(M, A, lds_raw, n_rec, n_srv_pref, k_short, fly_iters, fly_idle, emit_qmax);
}

// ... with the flight-parametric crossing in the flying waves (option "crossing" = 1; TAIL: hands its last packets to k_tail)
template <bool POLA, bool TAIL>
__global__ void __launch_bounds__(MCGPU_ROLES_BLOCK) k_thermal_roles_param(const DevModel M, const RunArgs A, int n_rec, int n_srv_pref,
                                                                           int k_short, int fly_iters, int fly_idle, int emit_qmax) {
  extern __shared__ double lds_raw[];
  roles_body<false, POLA, false, true, false, false, false, TAIL, false, true>(M, A, lds_raw, n_rec, n_srv_pref, k_short, fly_iters, fly_idle, emit_qmax);
}

// the role schedule on a Voronoi grid
template <bool POLA, bool MRW = false>
__global__ void __launch_bounds__(MCGPU_ROLES_BLOCK) k_thermal_voro_roles(const DevModel M, const RunArgs A, const VoroGrid G,
                                                                          int cache_log_ns, int n_rec, int n_srv_pref, int k_short,
                                                                          int fly_iters, int fly_idle, int emit_qmax) {
  extern __shared__ double lds_raw[];
  roles_body<true, POLA, false, false, MRW, true>(M, A, lds_raw, n_rec, n_srv_pref, k_short, fly_iters, fly_idle, emit_qmax,
                                                  &G, cache_log_ns);
}

}  // namespace mcgpu
