// mc_rounds.hip.h -- the packet loop as two alternating kernels over a packet pool in HBM.
//
// Why: inside one persistent kernel (mc_device.hip.h, k_thermal*) the rare, register-hungry
// phases (emission, scattering / re-emission with their FP64 transcendentals) set the register
// allocation (213-256 VGPR, 2 waves per SIMD) and run with a third of the lanes active, while
// the cell-crossing loop on its own needs 64 VGPR and runs 2-4x faster (tools/flight_bench).
// Here the two halves are separate kernels:
//
//   k_fly    persistent wavefronts, one pool slot per lane and pass: loads the packet, crosses
//            cells until it stops / leaves / is killed (or the pass is cut when too few lanes of
//            the wave still fly), stores it back, appends slots that need service to a list.
//            LDS holds only the radial/vertical grid vectors, kappa, kappa_abs and (2D) the
//            workgroup's private absorbed-energy grid, folded into HBM at the end of the launch.
//   k_serve  one lane per listed slot (compact: every lane has work): capteur for packets that
//            left, emission of the next packet id, scattering or absorption + re-emission, and
//            the set-up of the next flight.
//
// The host alternates k_serve / k_fly until no slot is flying or listed.  Physics, random
// streams and results are the same as the single-kernel engine: a packet's history depends only
// on (seed, packet id).  The packet state is a structure of arrays in HBM, indexed by slot, so
// every load and store of a wave is contiguous.
#pragma once
#include "mc_device.hip.h"

namespace mcgpu {


struct RoundArgs {
  int* list;               // slots that need service
  unsigned int* list_n;    // number of entries (k_fly appends, k_serve consumes)
  unsigned int* flying_n;  // slots still flying after k_fly
  int n_passes;            // slots per lane in k_fly
};

// LDS of k_fly: [E_lds (LDSE) | r_lim_2 | zmax | ch | rzn | tan_phi | kappa | kabs]
__host__ __device__ inline size_t lds_fly_doubles(const DevModel& M) {
  return (size_t)(M.n_rad + 1) + 3 * (size_t)M.n_rad + M.n_az + 2 * (size_t)M.n_lambda;
}

__device__ inline Lds lds_carve_fly(double* p, const DevModel& M) {
  Lds T;
  T.r_lim_2 = p; p += M.n_rad + 1;
  T.zmax = p; p += M.n_rad;
  T.ch = p; p += M.n_rad;
  T.rzn = p; p += M.n_rad;
  T.tan_phi = p; p += M.n_az;
  T.kappa = p; p += M.n_lambda;
  T.kabs = p;
  T.lq = T.cum = T.fstar = T.cdf = T.cost = nullptr;
  T.albedo = T.prob = T.g = nullptr;
  return T;
}

template <bool L3D, bool POLA, bool DARK, bool LDSE>
__device__ __forceinline__ void fly_body(const DevModel& M, const RunArgs& A, const Pool& P, const RoundArgs& R,
                                         double* lds_base) {
  double* const E_lds = lds_base;
  const Lds T = lds_carve_fly(lds_base + (LDSE ? M.n_cells : 0), M);
  stage(T.r_lim_2, M.r_lim_2, (size_t)M.n_rad + 1);
  stage(T.zmax, M.zmax, (size_t)M.n_rad);
  stage(T.ch, M.ch, (size_t)M.n_rad);
  for (int i = threadIdx.x; i < M.n_rad; i += blockDim.x) T.rzn[i] = (double)M.nz / M.zmax[i];
  stage(T.tan_phi, M.tan_phi_lim, (size_t)M.n_az);
  stage(T.kappa, M.kappa, (size_t)M.n_lambda);
  stage(T.kabs, M.kappa_abs, (size_t)M.n_lambda);
  if (LDSE)
    for (int i = threadIdx.x; i < M.n_cells; i += blockDim.x) E_lds[i] = 0.0;
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int n_rad = M.n_rad, nz = M.nz;
  const int n_lanes = gridDim.x * blockDim.x, gid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned int c_cross = 0, c_dark = 0, n_fly_left = 0;

#pragma unroll 1
  for (int pass = 0; pass < R.n_passes; ++pass) {
    const int slot = pass * n_lanes + gid;
    int stw = (slot < P.n_slots) ? P.st[slot] : S_DONE;
    int st = stw & ST_MASK;
    const bool loaded = (st == S_FLIGHT);
    double x = 0, y = 0, z = 0, u = 0, v = 0, w = 1, extr = 0, inv_a = 1, inv_w = 1, kf = 0, S0 = 1.0;
    int ri = 0, zj = 1, k = 1, lambda = 1, star_key = -1;
    if (loaded) {
      x = P.x[slot]; y = P.y[slot]; z = P.z[slot];
      u = P.u[slot]; v = P.v[slot]; w = P.w[slot];
      extr = P.extr[slot];
      ri = P.ri[slot]; zj = P.zj[slot]; k = P.k[slot];
      lambda = P.lambda[slot]; star_key = P.star_key[slot];
      if (POLA) S0 = P.S[slot];
      const double a = u * u + v * v;  // cylindrical_grid.f90:941-952
      inv_a = (a > TINY_REAL) ? 1.0 / a : HUGE_REAL;
      inv_w = (fabs(w) > TINY_REAL) ? 1.0 / w : copysign(HUGE_DP, w);
      kf = is_real_cell<L3D>(n_rad, nz, ri, zj) ? M.kappa_factor[cell_index<L3D>(n_rad, nz, ri, zj, k)] : 0.0;
    }
    const int n_start = __popcll(__ballot(loaded));
    bool dirflip = false;
    // ---- cell crossings (physical_length, optical_depth.f90:77-178) ---------------------
#pragma unroll 1
    for (int it = 0; it < A.inner_iters && n_start > 0; ++it) {
      if (it > 0 && __popcll(__ballot(st == S_FLIGHT)) * 64 < A.min_active * n_start) break;
      if (st == S_FLIGHT) {
        const int azj = zj < 0 ? -zj : zj;
        const bool out = (ri == n_rad + 1) || ((azj == nz + 1) && (fabs(z) > M.zmaxmax));
        bool killed = false;
        if (star_key >= 0) killed = (ri + (n_rad + 2) * ((zj + nz + 1) + (2 * nz + 3) * (k - 1))) == star_key;
        if (out) {
          st = S_EXITED;
        } else if (killed) {
          st = S_KILLED;
        } else {
          const bool real_cell = is_real_cell<L3D>(n_rad, nz, ri, zj);
          const int ic = real_cell ? cell_index<L3D>(n_rad, nz, ri, zj, k) : 0;
          const double opacity = real_cell ? T.kappa[lambda - 1] * kf : 0.0;
          {
            double x1, y1, z1, l;
            int ri1, zj1, k1;
            MCGPU_CROSS<L3D>(T, M, x, y, z, u, v, w, inv_a, inv_w, ri, zj, k, x1, y1, z1, ri1, zj1, k1, l);
            c_cross++;
            const double tau = l * opacity;
            if (tau > extr) {
              const double lc = l * (extr / tau);
              if (real_cell) deposit<LDSE>(A.E_abs, E_lds, ic, T.kabs[lambda - 1] * lc * S0);
              x = x + lc * u;
              y = y + lc * v;
              z = z + lc * w;
              if (L3D) index_cell<L3D>(T, M, x, y, z, ri, zj, k);  // optical_depth.f90:162-165
              st = S_INTERACT;
            } else {
              extr = extr - tau;
              if (real_cell) deposit<LDSE>(A.E_abs, E_lds, ic, T.kabs[lambda - 1] * l * S0);
              const bool real1 = is_real_cell<L3D>(n_rad, nz, ri1, zj1);
              const int ic1 = real1 ? cell_index<L3D>(n_rad, nz, ri1, zj1, k1) : 0;
              if (DARK && real1 && M.dark[ic1]) {
                // dark-zone mirror (optical_depth.f90:104-112), decided as soon as the next cell
                // is known: back to the entry point of the cell just crossed, direction
                // reversed, then an interaction there (a real cell never exits nor holds the
                // star, so the reference's exit / star tests of the next turn cannot fire first)
                u = -u; v = -v; w = -w;
                c_dark++;
                dirflip = true;
                st = S_INTERACT;
              } else {
                x = x1; y = y1; z = z1;
                ri = ri1; zj = zj1; k = k1;
                kf = real1 ? M.kappa_factor[ic1] : 0.0;
              }
            }
          }
        }
      }
    }
    if (loaded) {
      P.x[slot] = x; P.y[slot] = y; P.z[slot] = z;
      P.extr[slot] = extr;
      P.ri[slot] = ri; P.zj[slot] = zj; P.k[slot] = k;
      if (DARK && dirflip) { P.u[slot] = u; P.v[slot] = v; P.w[slot] = w; }
      P.st[slot] = (stw & ~ST_MASK) | st;
    }
    // slots that need service (also those that were already waiting): one list append per wave
    const bool need = (st != S_FLIGHT) && (st != S_DONE);
    const unsigned long long mask = __ballot(need);
    if (mask) {
      const int leader = __ffsll((long long)mask) - 1;
      unsigned int base = 0;
      if (lane == leader) base = atomicAdd(R.list_n, (unsigned int)__popcll(mask));
      base = __shfl(base, leader);
      if (need) R.list[base + __popcll(mask & ((1ull << lane) - 1ull))] = slot;
    }
    if (st == S_FLIGHT) n_fly_left++;
  }

  if (LDSE) {  // fold the workgroup's private grid into HBM
    __syncthreads();
    for (int i = threadIdx.x; i < M.n_cells; i += blockDim.x) {
      const double e = E_lds[i];
      if (e != 0.0) atomic_add_f64(&A.E_abs[i], e);
    }
  }
  unsigned int cs[3] = {c_cross, c_dark, n_fly_left};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    unsigned long long vsum = cs[q];
    for (int off = 32; off > 0; off >>= 1) vsum += __shfl_down(vsum, off);
    if (lane == 0 && vsum) {
      if (q == 0) atomicAdd(&A.counters[1], vsum);
      else if (q == 1) atomicAdd(&A.counters[7], vsum);
      else atomicAdd(R.flying_n, (unsigned int)vsum);
    }
  }
}

template <bool L3D, bool POLA, bool DARK>
__global__ void __launch_bounds__(1024) k_fly_lds(const DevModel M, const RunArgs A, const Pool P, const RoundArgs R) {
  extern __shared__ double lds_raw[];
  fly_body<L3D, POLA, DARK, true>(M, A, P, R, lds_raw);
}
template <bool L3D, bool POLA, bool DARK>
__global__ void __launch_bounds__(512) k_fly_hbm(const DevModel M, const RunArgs A, const Pool P, const RoundArgs R) {
  extern __shared__ double lds_raw[];
  fly_body<L3D, POLA, DARK, false>(M, A, P, R, lds_raw);
}

// ---------------------------------------------------------------------------------------------
// k_serve: everything that happens to a packet between two flights
// ---------------------------------------------------------------------------------------------
template <bool L3D, bool POLA>
__global__ void __launch_bounds__(256) k_serve(const DevModel M, const RunArgs A, const Pool P, const RoundArgs R) {
  extern __shared__ double lds_raw[];
  const Lds T = lds_carve(lds_raw, M);
  lds_stage(T, M);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int n_rad = M.n_rad, nz = M.nz;
  const unsigned int n_list = *R.list_n;
  unsigned int c_pack = 0, c_flight = 0, c_scatt = 0, c_abs = 0, c_esc = 0, c_kill = 0;
  const unsigned int n_round = (n_list + 63u) & ~63u;  // whole waves enter the loop body together

#pragma unroll 1
  for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {
    const bool valid = i < n_list;
    const int slot = valid ? R.list[i] : 0;
    int stw = valid ? P.st[slot] : S_DONE;
    int st = stw & ST_MASK;
    bool flag_star = (stw & ST_STAR) != 0, flag_scatt = (stw & ST_SCATT) != 0, flag_ism = (stw & ST_ISM) != 0;
    double x = 0, y = 0, z = 0, u = 0, v = 0, w = 1, extr = 0;
    double S[4] = {1.0, 0.0, 0.0, 0.0};
    int ri = 0, zj = 1, k = 1, lambda = 1, star_key = -1;
    Rng rng;
    rng.init(A.seed, 0);
    float tau_rand = 0.0f;
    if (valid && st != S_EMIT) {
      x = P.x[slot]; y = P.y[slot]; z = P.z[slot];
      u = P.u[slot]; v = P.v[slot]; w = P.w[slot];
      ri = P.ri[slot]; zj = P.zj[slot]; k = P.k[slot];
      lambda = P.lambda[slot];
      rng.p_lo = P.p_lo[slot]; rng.p_hi = P.p_hi[slot]; rng.event = P.event[slot];
      if (POLA) {
        S[0] = P.S[slot]; S[1] = P.S[P.n_slots + slot];
        S[2] = P.S[2 * (size_t)P.n_slots + slot]; S[3] = P.S[3 * (size_t)P.n_slots + slot];
      }
    }

#pragma unroll 1
    for (int rep = 0; rep < 4; ++rep) {  // > 1 turn only when a fresh packet misses the grid
      // ---- packets that left the grid or hit a star ---------------------------------------
      if (st == S_EXITED) {
        if (!flag_ism) {
          capteur<POLA>(M, A.sed, lambda, u, v, w, S, flag_star, flag_scatt);
          c_esc++;
        }
        st = S_EMIT;
      }
      if (st == S_KILLED) {
        c_kill++;
        st = S_EMIT;
      }
      // ---- EMIT: next packet id, emission (dust_transfer.f90:529-541, 1047-1151) ----------
      {
        const bool need = (st == S_EMIT);
        const unsigned long long mask = __ballot(need);
        if (mask) {
          const int leader = __ffsll((long long)mask) - 1;
          unsigned long long base = 0;
          if (lane == leader) base = atomicAdd(A.next_packet, (unsigned long long)__popcll(mask));
          base = __shfl(base, leader);
          const unsigned long long my = base + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
          if (need && my >= A.n_packets) st = S_DONE;
          if (need && my < A.n_packets) {
            rng.init(A.seed, A.first_packet + my);
            c_pack++;
            float f[12];
            rng.emission_event(f);
            tau_rand = f[8];
            lambda = select_wl_em(T, M, f[0]);
            atomic_add_f64(&A.n_sent[lambda - 1], 1.0);
            bool lintersect;
            flag_scatt = false;
            S[0] = 1.0; S[1] = 0.0; S[2] = 0.0; S[3] = 0.0;
            st = S_NEWFLIGHT;
            CylEmitOps<L3D> ops{T, M, ri, zj, k};
            const int rc = emit_packet(M, f, lambda, T.fstar[lambda - 1], M.frac_E_disk[lambda - 1],
                                       M.prob_E_cell ? M.prob_E_cell + (size_t)(M.n_cells + 1) * (lambda - 1) : nullptr,
                                       ops, x, y, z, u, v, w, flag_star, flag_ism, lintersect);
            if (rc) {
              *A.err = rc;
              st = S_DONE;
            }
            if (st != S_DONE && !lintersect) st = S_EXITED;  // never entered the grid (:549-550)
          }
        }
      }
      if (__ballot(st == S_EXITED) == 0ull) break;
    }

    // ---- INTERACT: scatter or absorb + re-emit (dust_transfer.f90:1260-1402) ----------------
    if (st == S_INTERACT) {
      float g[8];
      rng.interaction_event(g);
      tau_rand = g[5];
      double u1, v1, w1;
      const int ic = cell_index<L3D>(n_rad, nz, ri, zj, k);
      // every k_fly launch folds its deposits into HBM, so the running sum is the energy absorbed so
      // far by all packets of this GPU (x n_replicas: thermal_emission.f90:670)
      interact<POLA>(T, M, g, lambda, u, v, w, u1, v1, w1, S, flag_star, flag_scatt, c_scatt, c_abs, [&]() {
        return A.frozen ? A.E_prior[ic]
                        : __hip_atomic_load(&A.E_abs[ic], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * A.qscale;
      }, M.volume + ic);
      if (!flag_scatt) flag_ism = false;
      u = u1; v = v1; w = w1;
      st = S_NEWFLIGHT;
    }

    // ---- NEWFLIGHT: optical depth to the next event, star on the way ------------------------
    if (st == S_NEWFLIGHT) {
      const float rand = tau_rand;  // dust_transfer.f90:1208-1215 (tau in FP64)
      extr = (rand > 1.0e-6f) ? -log(1.0 - (double)rand) : (double)rand;
      const int i_star = intersect_stars(M, x, y, z, u, v, w);  // optical_depth.f90:68
      star_key = -1;
      if (i_star > 0) {
        const int* sc = &M.star_cell[4 * (i_star - 1)];
        star_key = sc[0] + (n_rad + 2) * ((sc[1] + nz + 1) + (2 * nz + 3) * (sc[2] - 1));
      }
      c_flight++;
      st = S_FLIGHT;
    }

    if (valid) {
      if (st == S_FLIGHT) {
        P.x[slot] = x; P.y[slot] = y; P.z[slot] = z;
        P.u[slot] = u; P.v[slot] = v; P.w[slot] = w;
        P.extr[slot] = extr;
        P.ri[slot] = ri; P.zj[slot] = zj; P.k[slot] = k;
        P.lambda[slot] = lambda; P.star_key[slot] = star_key;
        P.p_lo[slot] = rng.p_lo; P.p_hi[slot] = rng.p_hi; P.event[slot] = rng.event;
        if (POLA) {
          P.S[slot] = S[0]; P.S[P.n_slots + slot] = S[1];
          P.S[2 * (size_t)P.n_slots + slot] = S[2]; P.S[3 * (size_t)P.n_slots + slot] = S[3];
        }
      }
      P.st[slot] = st | (flag_star ? ST_STAR : 0) | (flag_scatt ? ST_SCATT : 0) | (flag_ism ? ST_ISM : 0);
    }
  }

  unsigned int cs[8] = {c_pack, 0u, c_flight, c_scatt, c_abs, c_esc, c_kill, 0u};
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    unsigned long long vsum = cs[q];
    for (int off = 32; off > 0; off >>= 1) vsum += __shfl_down(vsum, off);
    if (lane == 0 && vsum) atomicAdd(&A.counters[q], vsum);
  }
}

// lists the slots that hold a packet in flight (hand-over to the single-kernel finisher)
__global__ void k_collect_flying(const Pool P, int* list, unsigned int* n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool fly = (i < P.n_slots) && ((P.st[i] & ST_MASK) == S_FLIGHT);
  const unsigned long long mask = __ballot(fly);
  if (mask) {
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)mask) - 1;
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(n, (unsigned int)__popcll(mask));
    base = __shfl(base, leader);
    if (fly) list[base + __popcll(mask & ((1ull << lane) - 1ull))] = i;
  }
}

// all slots start empty and listed
__global__ void k_pool_init(const Pool P, int* list) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < P.n_slots) {
    P.st[i] = S_EMIT;
    list[i] = i;
  }
}

}  // namespace mcgpu
