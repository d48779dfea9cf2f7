// The tail of a launch: ONE PACKET PER WAVE, the other 63 lanes working ahead for it.
//
// A packet is sequential, and the slowest of 1e8 packets of ref4.1 has ~3e4 events (P(N > n) ~ exp(-n / 2300)); at
// 2.2-2.5 us per event for a lone packet in the throughput kernels that is 73 ms at the end of EVERY launch, whatever its
// size, and seconds on a thick disk.  A lone packet's event is ~650 instructions at ~4 cycles
// each plus ~1600 cycles of waiting on dependent table reads: the lane that owns the packet is the only one working.
// So when a workgroup of the role kernel has few packets left it hands them over (RunArgs::carry_out,
// mc_roles.hip.h "Chunks without tails"), and this kernel runs each of them on a whole wave:
//   * the random numbers of the NEXT 64 interactions -- counter-based: they depend on the packet id and the event
//     number only -- are drawn by the 64 lanes at once (lane j: event e + j), together with everything that depends
//     on them alone: -log(1 - rand) of the next flight's optical depth, sin / cos of the two azimuths an interaction may
//     use.  One batch costs what ONE event's draws cost; an event fetches its values with a wave-uniform shuffle;
//   * the three table searches of an interaction (scattering angle: 181 entries, Temp_LTE: n_T, the re-emission CDF:
//     n_lambda) are ONE probe per lane and a ballot instead of 6-8 dependent bisection steps;
//   * the crossing (fly_step_2d / fly_step_3d), the Stokes update and the bookkeeping run wave-uniform, lane 0 deposits.
// Every value is computed by the very expressions of the throughput kernels (only by another lane, or earlier), so a
// packet's history is the same whichever kernel finishes it: the frozen parity tests run through this path.
//
// The last packets on the host (round 6; RunArgs::tail_host_max, host_tail.cpp).  A wave runs a lone packet's event in
// 1.0-1.6 us -- the arithmetic depth of one event on a machine built for throughput --, a host core runs it in 50-100 ns,
// and the longest packet of a launch is one chain of 3e4 (ref4.1) to 5e5 (a thick disk) events: 30 to 800 ms at the end of
// every launch.  So k_tail only THINS the tail out: its packets' remaining lives are exponentially distributed, the
// machine runs thousands of them at once, and once no more than tail_host_max are unfinished every wave writes its
// packet back as a record (at the top of its next interaction: the state a record in S_INTERACT holds) and leaves.
// The library's host side (host_tail.cpp: THIS header compiled for the CPU, one packet per thread) finishes those from
// copies of the tables, the absorbed energy and the counters and hands the sums back.  Same functions, same records,
// same random numbers: a packet's history does not depend on where it ends.
#pragma once
#include "mc_roles.hip.h"

namespace mcgpu {

#ifdef MCGPU_HOST_TAIL
#define TAIL_WL(lane) (-1)     // (the host runs one lane: the walk's searches are the throughput kernels' bisections)
#else
#define TAIL_WL(lane) (lane)
#endif

constexpr int TAIL_N_COUNTERS = 10;  // packets .. mrw_steps (= TAIL_N_COUNTERS of include/mcgpu.h)
constexpr int TAIL_LONGEST = 16;     // counters[16 .. 20]: (events << 32 | count) of the longest packet, by atomicMax: its
                                     // crossings (whole life), and the scatterings, absorptions, walks and walk steps this kernel ran

// smallest k in [lo, hi) with tab[k] >= x, else hi (tab non-decreasing): one probe per lane and pass
template <typename Tp>
__device__ inline int wave_first_ge(const Tp* tab, int lo, int hi, Tp x, int lane) {
#ifdef MCGPU_HOST_TAIL   // (host_tail.cpp: one lane -- the bisection of the throughput kernels; the tables are monotone)
  (void)lane;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (tab[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
#else
  for (int base = lo; base < hi; base += BIN_WAVE) {
    const int k = base + lane;
    const bool hit = (k < hi) && !(tab[k < hi ? k : lo] < x);
    const unsigned long long m = __ballot(hit);
    if (m) return base + (__ffsll((long long)m) - 1);
  }
  return hi;
#endif
}

// what the 64 lanes hold for the interactions [base, base + 64) of the wave's packet
struct TailBatch {
  float g0, g1, g2, g3, g4, g5;   // the interaction's draws (Rng::interaction_event)
  double tau;                     // the optical depth of the flight that follows (dust_transfer.f90:1208-1215)
  double ss, cs, sa, ca;          // sin, cos of pi (2 g3 - 1) (scattering) and of pi (2 g4 - 1) (re-emission)
  uint32_t base;                  // first event of the batch (wave-uniform); 0: nothing drawn yet
};

__device__ inline void tail_draw(TailBatch& B, uint32_t k0, uint32_t k1, uint32_t p_lo, uint32_t p_hi, uint32_t event, int lane) {
  Rng rng;
  rng.k0 = k0; rng.k1 = k1; rng.p_lo = p_lo; rng.p_hi = p_hi; rng.event = event + (uint32_t)lane;
  float g[8];
  rng.interaction_event(g);   // (one block per event; scattering method 1 never comes here)
  B.g0 = g[0]; B.g1 = g[1]; B.g2 = g[2]; B.g3 = g[3]; B.g4 = g[4]; B.g5 = g[5];
  const float rand = g[5];
  B.tau = tau_of_draw(rand);
  sincos_pi(2.0 * (double)g[3] - 1.0, &B.ss, &B.cs);
  sincos_pi(2.0 * (double)g[4] - 1.0, &B.sa, &B.ca);
  B.base = event;
}

// One packet, from the state its record holds to its end.  Wave-uniform control flow: every lane holds the same
// packet state; lane 0 makes the deposits and counts.
// Where a packet's summed deposits go, and what of the wave's own deposits E_abs does not hold yet (host_tail.cpp keeps a
// cache of deposits per thread: threads that add to one shared array line by line serialise each other)
#ifndef MCGPU_TAIL_DEPOSIT
#define MCGPU_TAIL_DEPOSIT(A, ic, v) atomic_add_f64(&(A).E_abs[ic], (v))
#define MCGPU_TAIL_UNFOLDED(ic) 0.0
#define MCGPU_TAIL_LOAD_E(p) (*(p))   // (the cell's absorbed energy: a PLAIN load on the device, see cell_energy below; the
#endif                               // host build reads it as a relaxed atomic -- other threads add to it meanwhile)

#ifndef MCGPU_TAIL_TEST_HOOK   // (tests/emu: hand a packet over after a given number of its events, whatever the others do --
#define MCGPU_TAIL_TEST_HOOK(events_here) false   // one emulated lane runs the packets one after the other)
#endif

// have the tail's unfinished packets become few enough for the host?  (wave-uniform: lane 0's load)
__device__ inline bool tail_hand_over_now(const RunArgs& A, unsigned int n_total) {
  if (A.tail_host_max == 0u) return false;
  unsigned int done = __hip_atomic_load(A.tail_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  done = __shfl(done, 0);
  return n_total - done <= A.tail_host_max;
}

// Returns true when the packet has ended here, false when it was written to A.tail_out for the host.
template <bool L3D, bool POLA, bool DARK, bool MRW>
__device__ __forceinline__ bool tail_packet(const Lds& T, const DevModel& M, const RunArgs& A, const Rec<POLA>& R0, int lane,
                                            unsigned int* cs, unsigned int n_total = 0u) {
  const int n_rad = M.n_rad, nz = M.nz;
  const uint32_t key0 = (uint32_t)A.seed, key1 = (uint32_t)(A.seed >> 32);
  Flight F;
  flight_clear(F);
  F.x = R0.x; F.y = R0.y; F.z = R0.z; F.u = R0.u; F.v = R0.v; F.w = R0.w; F.extr = R0.extr; F.S0 = R0.S[0];
  double S1 = POLA ? R0.S[POLA ? 1 : 0] : 0.0, S2 = POLA ? R0.S[POLA ? 2 : 0] : 0.0, S3 = POLA ? R0.S[POLA ? 3 : 0] : 0.0;
  F.ri = R0.ri; F.zj = R0.zj; F.k = R0.k; F.star_key = R0.star_key; F.pk_cross = R0.pk_cross;
  if (L3D && F.k < 0) index_cell<true>(T, M, F.x, F.y, F.z, F.ri, F.zj, F.k, F.ri);   // (a stop the role kernel left to re-index)
  int lambda = R0.lambda, st = R0.flags & ST_MASK;
  bool flag_star = (R0.flags & ST_STAR) != 0, flag_scatt = (R0.flags & ST_SCATT) != 0, flag_ism = (R0.flags & ST_ISM) != 0;
  int n_int = MRW ? ((R0.flags >> ST_NINT_SHIFT) & 7) : 0;
  const uint32_t p_lo = R0.p_lo, p_hi = R0.p_hi;
  uint32_t event = R0.event;
  double tau_next = 0.0;   // the optical depth of the next flight
  {                        // (a record in state NEWFLIGHT carries the draw, not the depth)
    const float rand = R0.tau_rand;
    tau_next = tau_of_draw(rand);
  }
  unsigned int c_cross = 0, c_flight = 0, c_scatt = 0, c_abs = 0, c_esc = 0, c_kill = 0, c_dark = 0, c_walks = 0, c_steps = 0;
  if (st == S_EMIT) {
    // a work item the role kernel had reserved but not started: the packet is emitted here (mc_photon_loop body,
    // dust_transfer.f90:529-541, as in roles_body)
    Rng rng;
    rng.init(A.seed, ((unsigned long long)p_hi << 32) | p_lo);
    float f[12];
    rng.emission_event(f);
    lambda = select_wl_em(T, M, f[0]);
    if (lane == 0) unsafeAtomicAdd(&A.n_sent[lambda - 1], 1.0);
    cs[0] += 1u;
    bool lintersect;
    int ri = 0, zj = 1, k = 1;
    CylEmitOps<L3D> ops{T, M, ri, zj, k};
    const int rc = emit_packet(M, f, lambda, T.fstar[lambda - 1], M.frac_E_disk[lambda - 1],
                               M.prob_E_cell ? M.prob_E_cell + (size_t)(M.n_cells + 1) * (lambda - 1) : nullptr,
                               ops, F.x, F.y, F.z, F.u, F.v, F.w, flag_star, flag_ism, lintersect);
    if (rc) { *A.err = rc; return true; }
    flag_scatt = false;
    F.S0 = 1.0; S1 = S2 = S3 = 0.0;
    F.ri = ri; F.zj = zj; F.k = k; F.star_key = -1; F.pk_cross = 0u; F.extr = 0.0;
    n_int = 0;
    event = rng.event;
    const float rand = f[8];
    tau_next = tau_of_draw(rand);
    st = lintersect ? S_NEWFLIGHT : S_EXITED;
  }
  F.ic = -1;   // (no cell's kappa_factor at hand yet: flight_constants<..., REUSE>)
  if (st == S_FLIGHT) flight_constants<L3D, false, true>(T, M, F, lambda);
  TailBatch B;
  B.base = 0u;

  // Deposits: a trapped packet deposits into the same cell event after event, and an atomic drops the cell's line from
  // L2 -- the very line the next absorption's Temp_LTE reads.  The wave therefore sums its deposits locally while the
  // packet stays in a cell and makes ONE atomic when it moves on (or ends).
  int dep_cell = -1;
  double dep_sum = 0.0;
  auto flush = [&]() {
    if (dep_cell >= 0 && dep_sum != 0.0 && lane == 0) MCGPU_TAIL_DEPOSIT(A, dep_cell, dep_sum);
    dep_cell = -1; dep_sum = 0.0;
  };
  auto add_energy = [&](int ic, double v) {
    if (ic != dep_cell) { flush(); dep_cell = ic; }
    dep_sum += v;
  };
  // The cell's absorbed energy for Temp_LTE: what E_abs held when the packet entered the cell plus the wave's own
  // unflushed deposits.  A PLAIN load, once per cell: the launch's bulk is complete when this kernel runs (the role
  // kernels have folded their private grids, the log of a binned run has been folded), so E_abs only still changes by
  // the tail packets' own deposits -- 1e-4 of the total -- and a value some microseconds old is as good an estimate as
  // the reference's per-thread partial sum.  (The agent-scope load of the throughput kernels goes to the memory side
  // -- the line was last written by an atomic --: ~2 us, more than the rest of the event.)
  int e_cell = -1, v_cell = -1;
  double e_val = 0.0, v_val = 1.0;
  auto cell_volume = [&](int ic) {   // (a trapped packet is absorbed in one cell again and again: one load per cell)
    if (ic != v_cell) { v_val = M.volume[ic]; v_cell = ic; }
    return v_val;
  };
  auto cell_energy = [&](int ic) {
    if (A.frozen) return A.E_prior[ic];
    if (ic != e_cell) { e_val = MCGPU_TAIL_LOAD_E(&A.E_abs[ic]) + MCGPU_TAIL_UNFOLDED(ic); e_cell = ic; }
    return (e_val + (ic == dep_cell ? dep_sum : 0.0)) * A.qscale;
  };

  // Stokes Q, U, V lazily (POLA): a scattering leaves I unchanged (update_Stokes renormalises it, scattering.f90:1294) and
  // an absorption sets Q = U = V = 0 (dust_transfer.f90:1369), so the scatterings BEFORE a packet's last absorption never
  // reach the SED -- and a trapped packet has thousands of them, each a rotation, two square roots and three divisions
  // (290 of an event's ~700 dependent vector instructions).  Lane q keeps the q-th scattering since the last absorption
  // (directions before and after, angle bin, draw, wavelength); an absorption forgets them; the packet's end -- or the
  // 65th pending scattering -- has every lane compute ITS scattering's rotation and Mueller ratios at once
  // (stokes_rotation, mueller_pos) and the wave apply them in order (stokes_apply: the very expressions of
  // update_stokes).  Q, U, V at the end are the throughput kernels' bit for bit; I differs from theirs by the rounding of
  // the forgotten renormalisations (1e-16 per scattering).
  int n_pend = 0, pe_it = 1, pe_lam = 1;
  float pe_r2 = 0.0f;
  double pe_u0 = 0.0, pe_v0 = 0.0, pe_w0 = 1.0, pe_u1 = 0.0, pe_v1 = 0.0, pe_w1 = 1.0;
  auto stokes_flush = [&]() {
    if (!POLA || n_pend == 0) return;
    double cw = 1.0, sw = 0.0, M12 = 0.0, M22 = 1.0, M33 = 1.0, M34 = 0.0, M44 = 1.0;
    if (lane < n_pend) {
      stokes_rotation(pe_u0, pe_v0, pe_w0, pe_u1, pe_v1, pe_w1, cw, sw);
      mueller_pos(M, pe_lam, pe_it, pe_r2, -1, M12, M22, M33, M34, M44);
    }
    double S[4] = {F.S0, S1, S2, S3};
    for (int q = 0; q < n_pend; ++q)
      stokes_apply(S, __shfl(cw, q), __shfl(sw, q), __shfl(M12, q), __shfl(M22, q), __shfl(M33, q), __shfl(M34, q), __shfl(M44, q));
    F.S0 = S[0]; S1 = S[1]; S2 = S[2]; S3 = S[3];
    n_pend = 0;
  };

  for (;;) {
    if (st == S_INTERACT) {
      // ---- the interaction's draws, from the batch the wave drew ahead ------------------------------------------
      if (B.base == 0u || event - B.base >= (uint32_t)BIN_WAVE) {
        if (tail_hand_over_now(A, n_total) || MCGPU_TAIL_TEST_HOOK(event - R0.event)) {   // (once per batch of 64 events) -> the host finishes this packet
          stokes_flush();
          flush();
          if (lane == 0) {
            const unsigned int at = atomicAdd(A.tail_out_n, 1u);
            if (at >= A.tail_host_max) *A.err = 17;   // (never: no more than tail_host_max packets are unfinished)
            else {
              Rec<POLA>& R = reinterpret_cast<Rec<POLA>*>(A.tail_out)[at];
              rec_copy(&R, static_cast<const Rec<POLA>*>(nullptr));
              R.x = F.x; R.y = F.y; R.z = F.z; R.u = F.u; R.v = F.v; R.w = F.w; R.extr = 0.0; R.S[0] = F.S0;
              if (POLA) { R.S[POLA ? 1 : 0] = S1; R.S[POLA ? 2 : 0] = S2; R.S[POLA ? 3 : 0] = S3; }
              R.ri = F.ri; R.zj = F.zj; R.k = F.k; R.lambda = lambda; R.star_key = -1;
              R.p_lo = p_lo; R.p_hi = p_hi; R.event = event; R.pk_cross = F.pk_cross; R.tau_rand = 0.0f;
              R.flags = S_INTERACT | (flag_star ? ST_STAR : 0) | (flag_scatt ? ST_SCATT : 0) | (flag_ism ? ST_ISM : 0) |
                        (MRW ? (n_int << ST_NINT_SHIFT) : 0);
            }
          }
          cs[1] += c_cross; cs[2] += c_flight; cs[3] += c_scatt; cs[4] += c_abs; cs[5] += c_esc; cs[6] += c_kill; cs[7] += c_dark;
          cs[8] += c_walks; cs[9] += c_steps;
          return false;
        }
        tail_draw(B, key0, key1, p_lo, p_hi, event, lane);
      }
      const int q = (int)(event - B.base);
      const float g0 = __shfl(B.g0, q), g1 = __shfl(B.g1, q), g2 = __shfl(B.g2, q), g3 = __shfl(B.g3, q);
      tau_next = __shfl(B.tau, q);
      event += 1u;
      // ---- interact_direction (mc_device.hip.h), its searches one probe per lane ------------------------------
      const bool scat = g0 < T.albedo[lambda - 1];  // dust_transfer.f90:1284
      const int lambda_in = lambda;
      int itheta = 1;
      double cospsi, sphi, cphi;
      int abs_Ti = 0;            // (an absorption's temperature bracket, for the walk that may follow it)
      double abs_frac = 0.0;
      if (scat) {
        flag_scatt = true;
        c_scatt++;
        if (M.aniso_method == 1) {  // angle_diff_theta_pos (scattering.f90:1433-1475)
          const size_t col = M.p_lambda_fixed ? (size_t)0 : (size_t)(lambda - 1);
          const float* prob = T.prob + (size_t)(M.nang + 1) * col;
          itheta = wave_first_ge(prob, 1, M.nang, g1, lane);
          const double c0 = T.cost[itheta - 1], c1 = T.cost[itheta];
          cospsi = c0 + (double)g2 * (c1 - c0);
        } else {  // hg (scattering.f90:1354-1383)
          const float gg = T.g[lambda - 1];
          const double rand_dp = fmin((double)g1, 1.0 - 1e-6);
          if (fabsf(gg) > 1.17549435e-38f) {
            const double ga = (double)gg, gb = ga * ga;
            const double qq = (1.0 - gb) / (1.0 - ga + 2.0 * ga * rand_dp);
            cospsi = (1.0 + gb - qq * qq) / (2.0 * ga);
          } else {
            cospsi = 2.0 * rand_dp - 1.0;
          }
          itheta = (int)floor(acos(cospsi) * 180.0 / PI) + 1;
          if (itheta > M.nang) itheta = M.nang;
        }
        if (M.lisotropic) { itheta = 1; cospsi = 2.0 * (double)g1 - 1.0; }
        sphi = __shfl(B.ss, q); cphi = __shfl(B.cs, q);
      } else {
        c_abs++;
        flag_star = false;
        flag_scatt = false;
        // im_reemission_LTE (thermal_emission.f90:710-771): Temp_LTE, then the wavelength
        const int ic = cell_index<L3D>(n_rad, nz, F.ri, F.zj, F.k);
        const double Qheat = cell_energy(ic) * M.L_packet_th / cell_volume(ic);
        int Ti = 2;
        double frac_T2 = 0.0;
        if (!(Qheat < TINY_DP)) {
          const double log_Qheat = log_pos(Qheat);
          if (!(log_Qheat < T.lq[0])) {
            Ti = wave_first_ge(T.lq, 1, M.n_T - 1, log_Qheat, lane) + 1;  // first Ti in [2, n_T] with lq(Ti) >= log Qheat
            frac_T2 = (log_Qheat - T.lq[Ti - 2]) / (T.lq[Ti - 1] - T.lq[Ti - 2]);
          }
        }
        abs_Ti = Ti; abs_frac = frac_T2;
#ifdef MCGPU_HOST_TAIL
        lambda = reemission_wavelength(T, M, Ti, frac_T2, g2);   // (one lane: the throughput kernels' bisection)
#else
        {  // reemission_wavelength: the first l in [1, n_lambda) whose interpolated CDF reaches the draw, else n_lambda
          const double frac_T1 = 1.0 - frac_T2;
          const double* cdf1 = T.cdf + (size_t)M.n_lambda * (Ti - 2);
          const double* cdf2 = T.cdf + (size_t)M.n_lambda * (Ti - 1);
          int found = M.n_lambda;
          for (int base = 1; base < M.n_lambda; base += BIN_WAVE) {
            const int l = base + lane;
            const int ls = l < M.n_lambda ? l : 1;
            const double proba = frac_T1 * cdf1[ls - 1] + frac_T2 * cdf2[ls - 1];
            const unsigned long long m = __ballot((l < M.n_lambda) && !((double)g2 > proba));
            if (m) { found = base + (__ffsll((long long)m) - 1); break; }
          }
          lambda = found;
        }
#endif
        cospsi = 2.0 * (double)g3 - 1.0;
        sphi = __shfl(B.sa, q); cphi = __shfl(B.ca, q);
      }
      double u1, v1, w1;
      cdapres_sc(cospsi, sphi, cphi, scat ? F.u : 0.0, scat ? F.v : 0.0, scat ? F.w : 1.0, u1, v1, w1);
      if (!flag_scatt) flag_ism = false;  // absorbed and re-emitted by the dust (:1367)
      if (POLA) {   // (interact_stokes, lazily: see stokes_flush)
        if (!scat) { n_pend = 0; S1 = 0.0; S2 = 0.0; S3 = 0.0; }
        else if (M.aniso_method == 1) {
          if (n_pend == BIN_WAVE) stokes_flush();
          if (lane == n_pend) {
            pe_u0 = F.u; pe_v0 = F.v; pe_w0 = F.w; pe_u1 = u1; pe_v1 = v1; pe_w1 = w1;
            pe_it = itheta; pe_lam = lambda_in; pe_r2 = g2;
          }
          n_pend++;
        }
      }
      F.u = u1; F.v = v1; F.w = w1;
      if (MRW) {  // (dust_transfer.f90:1244-1249, 1222-1239; see roles_body)
        n_int = (F.pk_cross & 0x80000000u) ? 0 : (n_int < 7 ? n_int + 1 : 7);
        F.pk_cross &= 0x7FFFFFFFu;
        if (!flag_scatt && !flag_star && n_int > M.mrw_n_inter) {
          const int ic = cell_index<L3D>(n_rad, nz, F.ri, F.zj, F.k);
          double x = F.x, y = F.y, z = F.z, u = F.u, v = F.v, w = F.w;
          int lam2 = lambda;
          const bool done = mrw_walk(T, M, key0, key1, p_lo, p_hi, event, F.ri, F.zj, ic, F.S0, x, y, z, u, v, w, lam2,
                                     [&]() { return cell_energy(ic); },
                                     [&](double e) { add_energy(ic, e); }, c_walks, c_steps, L3D ? F.k : 1, TAIL_WL(lane),   // (lane: wave-wide searches)
                                     abs_Ti, abs_frac);
          if (done) { F.x = x; F.y = y; F.z = z; F.u = u; F.v = v; F.w = w; lambda = lam2; }
        }
      }
      st = S_NEWFLIGHT;
    }
    if (st == S_NEWFLIGHT) {  // (dust_transfer.f90:1208-1215; optical_depth.f90:68)
      F.extr = tau_next;
      const int i_star = intersect_stars(M, F.x, F.y, F.z, F.u, F.v, F.w);
      int key = -1;
      if (i_star > 0) {
        const int* sc = &M.star_cell[4 * (i_star - 1)];
        key = sc[0] + (n_rad + 2) * ((sc[1] + nz + 1) + (2 * nz + 3) * (sc[2] - 1));
      }
      F.star_key = key;
      c_flight++;
      flight_constants<L3D, false, true>(T, M, F, lambda);
      st = S_FLIGHT;
    }
    if (st == S_FLIGHT) {
      F.st = S_FLIGHT;
      int killed = 0;
      while (F.st == S_FLIGHT) {
        int dep_ic = -1;
        double dep_v = 0.0;
        if (L3D) killed += fly_step_3d<DARK, false, true, false, false, MRW>(T, M, A, nullptr, F, c_cross, c_kill, c_dark, dep_ic, dep_v);
        else killed += fly_step_2d<DARK, false, MRW, true>(T, M, A, nullptr, F, c_cross, c_kill, c_dark, &dep_ic, &dep_v);
        if (dep_ic >= 0) add_energy(dep_ic, dep_v);
      }
      st = F.st;
      if (killed) break;  // (the star's cell, or a runaway packet: finished)
    }
    if (st == S_EXITED) {  // capteur (output.f90:294-597)
      stokes_flush();
      if (!flag_ism) {
        if (lane == 0) {
          const double S[4] = {F.S0, S1, S2, S3};
          capteur<POLA>(M, A.sed, lambda, F.u, F.v, F.w, S, flag_star, flag_scatt);
        }
        c_esc++;
      }
      break;
    }
    if (st == S_EMIT) break;  // (killed at the star)
  }
  flush();
  {
    const unsigned int ev = (F.pk_cross & 0x7FFFFFFFu) + event;
    if (ev > cs[TAIL_N_COUNTERS]) {   // the longest packet of this wave so far: its own counts travel with its events
      cs[TAIL_N_COUNTERS] = ev;
      if (lane == 0 && ev > 10000u) {
        const unsigned long long hi = (unsigned long long)ev << 32;
        atomicMax(&A.counters[TAIL_LONGEST + 0], hi | (unsigned long long)(F.pk_cross & 0x7FFFFFFFu));
        atomicMax(&A.counters[TAIL_LONGEST + 1], hi | (unsigned long long)c_scatt);
        atomicMax(&A.counters[TAIL_LONGEST + 2], hi | (unsigned long long)c_abs);
        atomicMax(&A.counters[TAIL_LONGEST + 3], hi | (unsigned long long)c_walks);
        atomicMax(&A.counters[TAIL_LONGEST + 4], hi | (unsigned long long)c_steps);
      }
    }
  }
  cs[1] += c_cross; cs[2] += c_flight; cs[3] += c_scatt; cs[4] += c_abs; cs[5] += c_esc; cs[6] += c_kill; cs[7] += c_dark;
  cs[8] += c_walks; cs[9] += c_steps;
  return true;
}

#ifndef MCGPU_TAIL_BLOCK
#define MCGPU_TAIL_BLOCK 256
#endif
#ifndef MCGPU_TAIL_MIN_WAVES   // waves per SIMD the register budget is cut for (k_tail is throughput-bound while it holds more
#define MCGPU_TAIL_MIN_WAVES 1 // packets than waves reside; 1: whatever the packet's state needs, 190-260 VGPRs -> 2 waves)
#endif

// The packets the role kernel handed over: carry[0 .. *carry_n) (records), one per wave at a time, taken from a global
// counter.  Launched behind the role kernel (and the fold of its log) on the same stream.
template <bool L3D, bool POLA, bool DARK, bool MRW>
__global__ void __launch_bounds__(MCGPU_TAIL_BLOCK, MCGPU_TAIL_MIN_WAVES) k_tail(const DevModel M, const RunArgs A, const void* carry, const unsigned int* carry_n,
                                                           unsigned int* next) {
  extern __shared__ double lds_raw[];
  const Lds T = lds_carve(lds_raw, M);
  lds_stage(T, M);
  __syncthreads();
  const int lane = threadIdx.x & (BIN_WAVE - 1);
  const unsigned int n = *carry_n;
  const Rec<POLA>* recs = reinterpret_cast<const Rec<POLA>*>(carry);
  unsigned int cs[TAIL_N_COUNTERS + 1];   // (the last entry: the longest packet's events, a maximum -> slot 10)
  for (int q = 0; q <= TAIL_N_COUNTERS; ++q) cs[q] = 0u;
  for (;;) {
    unsigned int i = 0u;
    if (lane == 0) i = atomicAdd(next, 1u);
    i = __shfl(i, 0);
    if (i >= n) break;
    if (tail_hand_over_now(A, n)) {   // few enough are left: this one goes to the host as it is
      if (lane == 0) {
        const unsigned int at = atomicAdd(A.tail_out_n, 1u);
        if (at >= A.tail_host_max) *A.err = 17;
        else rec_copy(&reinterpret_cast<Rec<POLA>*>(A.tail_out)[at], &recs[i]);
      }
      continue;
    }
    const Rec<POLA> R = recs[i];
    const bool ended = tail_packet<L3D, POLA, DARK, MRW>(T, M, A, R, lane, cs, n);
    if (ended && A.tail_host_max && lane == 0) atomicAdd(A.tail_done, 1u);
  }
  if (lane == 0)
    for (int q = 0; q < TAIL_N_COUNTERS; ++q)
      if (cs[q]) atomicAdd(&A.counters[q], (unsigned long long)cs[q]);
  if (lane == 0 && cs[TAIL_N_COUNTERS]) atomicMax(&A.counters[10], (unsigned long long)cs[TAIL_N_COUNTERS]);
}

}  // namespace mcgpu
