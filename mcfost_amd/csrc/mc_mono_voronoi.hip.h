// SED mode on a Voronoi grid: the monochromatic packet loop of mc_mono.hip.h (forced scattering, exact
// per-stream stopping by scout + commit passes, ray-tracing method 1 deposits) with the grid operators
// of mc_voronoi.hip.h.  The mesh is 3D, so xI_scatt has one azimuth / elevation sub-bin per cell
// (dust_ray_tracing.f90:91-98: n_az_rt = n_theta_rt = 1).
#pragma once
#include "mc_mono.hip.h"
#include "mc_voronoi.hip.h"

namespace mcgpu {

template <bool POLA, bool SCOUT, bool F32 = false>
__device__ __forceinline__ void mono_body_voro(const DevModel& M, const MonoArgs& A, const VoroGrid& G,
                                               double* lds_base) {
  const Lds T = lds_carve(lds_base, M, true);
  lds_stage_mono(T, M, A.p_lambda);
  const MonoLds ML = mono_lds_setup<POLA>(M, A, lds_base, true);
  const int lane = threadIdx.x & 63;
  const int lambda = A.lambda;

  int st = S_EMIT;
  double x = 0, y = 0, z = 0, u = 0, v = 0, w = 1, extr = 0;
  int icell = 0, prev_cell = 0, star_icell = 0;
  bool flag_star = false, flag_scatt = false, flag_ism = false;
  double S[4] = {1.0, 0.0, 0.0, 0.0};
  Rng rng;
  rng.init(0, 0);
  unsigned int c_cross = 0, c_flight = 0, c_scatt = 0, c_abs = 0, c_esc = 0, c_kill = 0, c_pack = 0;
  unsigned int pk_cross = 0;
  unsigned long long pk_next = 0, pk_end = 0, my_item = 0;
  float tau_rand = 0.0f;
  const bool var = M.n_classes != 0;   // lvariable_dust: the tables of the cell's class, as in mono_body
  const int na1 = M.nang + 1;

  for (;;) {
    if (st == S_EXITED) {
      if (!flag_ism) {
        const int capt = capteur<POLA, true>(M, SCOUT ? nullptr : A.sed, lambda, u, v, w, S, flag_star, flag_scatt);
        if (SCOUT) { if (capt == A.capt_sup) A.hits[my_item] = 1; }
        else if (A.hit_count && capt == A.capt_sup) {
          unsigned long long ch, sq;
          mono_item<false>(A, my_item, ch, sq);
          atomicAdd(&A.hit_count[ch], 1ull);
        }
        if (capt > 0) c_esc++;
      }
      st = S_EMIT;
    }
    {
      const bool need = (st == S_EMIT);
      const unsigned long long mask = __ballot(need);
      if (mask) {
        if (pk_next >= pk_end) {
          const int leader = __ffsll((long long)mask) - 1;
          unsigned long long base = 0;
          if (lane == leader) base = atomicAdd(A.next_item, (unsigned long long)PK_BATCH);
          base = __shfl(base, leader);
          pk_next = base < A.n_items ? base : A.n_items;
          pk_end = (base + PK_BATCH < A.n_items) ? base + PK_BATCH : A.n_items;
          if (pk_end < pk_next) pk_end = pk_next;
        }
        const unsigned long long avail = pk_end - pk_next;
        const unsigned long long rank = (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
        const unsigned long long cnt = (unsigned long long)__popcll(mask);
        const unsigned long long my = pk_next + rank;
        const bool served = need && (rank < avail);
        if (need && !served && pk_next >= A.n_items) st = S_DONE;
        pk_next += (cnt < avail) ? cnt : avail;
        if (served) {
          my_item = my;
          unsigned long long chunk, seq;
          mono_item<SCOUT>(A, my, chunk, seq);
          rng.init(A.seed, ((chunk + (unsigned long long)A.first_chunk) << 40) | seq);
          c_pack++;
          pk_cross = 0;
          float f[12];
          rng.emission_event(f);
          tau_rand = f[8];
          bool lintersect;
          flag_scatt = false;
          S[0] = 1.0; S[1] = 0.0; S[2] = 0.0; S[3] = 0.0;
          VoroEmitOps ops{G, M, icell};
          const int rc = emit_packet(M, f, lambda, A.frac_E_stars, A.frac_E_disk, A.prob_E_cell, ops, x, y, z, u, v, w,
                                     flag_star, flag_ism, lintersect);
          if (rc) {
            *A.err = rc;
            st = S_DONE;
          }
          if (st != S_DONE) st = lintersect ? S_NEWFLIGHT : S_EXITED;
        }
      }
    }

    if (st == S_INTERACT) {  // forced scattering (dust_transfer.f90:1263-1278); no dark zone on this grid
      float g[8];
      rng.interaction_event(g, M.m1 != 0);
      tau_rand = g[5];
      const int cls = var ? M.cell_class[icell - 1] : -1;
      const Lds Tc = var ? class_tables(T, M, cls) : T;
      if (mono_attenuate<POLA>(Tc, lambda, S)) {
        c_abs++;
        st = S_EMIT;
      } else {
        double u1, v1, w1;
        int lam = lambda;
        const float* prob_c = (var && M.v_scatt) ? M.v_prob + ((size_t)cls * M.n_lambda + (A.p_lambda - 1)) * na1 : nullptr;
        interact<POLA>(Tc, M, g, lam, u, v, w, u1, v1, w1, S, flag_star, flag_scatt, c_scatt, c_abs,
                       []() { return 0.0; }, M.volume, true, prob_c, 0, (var && M.v_scatt) ? cls : -1);  // T.prob = column p_lambda
        u = u1; v = v1; w = w1;
        st = S_NEWFLIGHT;
      }
    }

    if (st == S_NEWFLIGHT) {
      const float rand = tau_rand;
      extr = tau_of_draw(rand);
      if (!SCOUT && A.rt1) angles_scatt_rt1<POLA>(M, A, ML.R, u, v, w, (F32 && M.n_classes == 0) ? ML.mu : nullptr, S);  // optical_depth.f90:65
      const int i_star = intersect_stars(M, x, y, z, u, v, w);
      star_icell = (i_star > 0) ? M.star_cell[4 * (i_star - 1)] : 0;
      c_flight++;
      prev_cell = 0;
      st = S_FLIGHT;
    }

    if (__ballot(st != S_DONE) == 0ull) break;

#pragma unroll 1
    for (int it = 0; it < A.inner_iters; ++it) {
      if (A.min_active > 0 && it > 0) {
        const int flying = __popcll(__ballot(st == S_FLIGHT)), alive = __popcll(__ballot(st != S_DONE));
        if (flying * 64 < A.min_active * alive) break;
      }
      RtDeposit dep;
      dep.on = false; dep.icell = 1; dep.phik = 1; dep.psup = 1; dep.l = 0.0;
      if (st == S_FLIGHT) {
        if (icell < 0) {
          st = S_EXITED;
        } else if (star_icell > 0 && icell == star_icell) {
          c_kill++;
          st = S_EMIT;
        } else {
          const VoroCell C = G.cell[icell - 1];
          const double opacity = (var ? M.v_kappa[(size_t)M.cell_class[icell - 1] * M.n_lambda + (lambda - 1)] : T.kappa[lambda - 1]) * C.kf;
          double x1, y1, z1, l, l_contrib, l_void;
          int next;
          voro_cross_cell(G, M, C, x, y, z, u, v, w, icell, prev_cell, x1, y1, z1, next, l, l_contrib, l_void);
          c_cross++;
          const double tau = l_contrib * opacity;
          const bool stop = tau > extr;
          const double lc = stop ? l_contrib * (extr / tau) : l_contrib;
          if (!SCOUT && A.rt1) { dep.on = true; dep.icell = icell; dep.l = lc; }  // save_radiation_field(l_contrib)
          if (stop) {
            const double ls = l_void + lc;
            x = nd_add(x, nd_mul(ls, u));
            y = nd_add(y, nd_mul(ls, v));
            z = nd_add(z, nd_mul(ls, w));
            st = S_INTERACT;
          } else {
            extr = extr - tau;
            x = x1; y = y1; z = z1;
            prev_cell = icell;
            icell = next;
          }
          if (++pk_cross > 200000000u) {  // a packet that never leaves: flag it, drop it
            *A.err = 13;
            st = S_EMIT;
          }
        }
      }
      if (!SCOUT && A.rt1 && __ballot(dep.on) != 0ull)
      {
#ifndef MCGPU_LANE_EMULATION
        if constexpr (F32) deposit_rt1_wave_f32<POLA>(M, A, ML.R, ML.mu, dep, S, flag_star, ML.tile, ML.tile_addr, ML.tile_mask);
        else
#endif
        deposit_rt1_wave<POLA>(M, A, ML.R, ML.mu, dep, S, flag_star, ML.tile, ML.tile_addr, ML.tile_mask);
      }
    }
  }

  if (!SCOUT) {
    unsigned int cs[8] = {c_pack, c_cross, c_flight, c_scatt, c_abs, c_esc, c_kill, 0u};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      unsigned long long vsum = cs[q];
      for (int off = 32; off > 0; off >>= 1) vsum += __shfl_down(vsum, off);
      if (lane == 0 && vsum) atomicAdd(&A.counters[q], vsum);
      // n_phot_envoyes(lambda) (dust_transfer.f90:536): every packet of this launch has the same wavelength, so the
      // wave adds its packet count once (one FP64 atomic per packet on ONE address would serialise the whole chip)
      if (q == 0 && lane == 0 && vsum) unsafeAtomicAdd(&A.n_sent[A.lambda - 1], (double)vsum);
    }
  }
}

template <bool POLA, bool SCOUT, bool F32 = false>
__global__ void __launch_bounds__(512) k_mono_voro(const DevModel M, const MonoArgs A, const VoroGrid G) {
  extern __shared__ double lds_raw[];
  mono_body_voro<POLA, SCOUT, F32>(M, A, G, lds_raw);
}

}  // namespace mcgpu
