// SED-mode packet loop: one wavelength of run_sed_mc
// (dust_transfer.f90:828-1042), i.e. mc_photon_loop with lmono and not lmono0 (:439-572):
//   * fixed wavelength, emission split star / disk by frac_E_stars(lambda), prob_E_cell(:,lambda)
//     (repartition_energie, thermal_emission.f90:1771-1949, stays on the host);
//   * forced scattering: Stokes *= albedo at every interaction, packet dropped below
//     tiny_real*1e6 or in a dark-zone cell (dust_transfer.f90:1263-1278);
//   * ray-tracing method 1 deposits: per flight the scattering angle towards every observer
//     (angles_scatt_rt1, dust_ray_tracing.f90:409-476), per cell crossing
//     xI_scatt(phik,psup,:,iRT,icell) += l * (RPO.M.ROP.Stokes) (calc_xI_scatt[_pola] :480-632
//     from save_radiation_field, radiation_field.f90:63-89);
//   * capteur into the SED arrays (output.f90:294-397, 572-592).
//
// The reference runs n_photons_loop sequential streams, each until n_photons2 packets landed in
// inclination bin capt_sup (or n_phot_lim were sent).  Here a SCOUT pass transports batches of
// every stream's packets without deposits and records which of them land in capt_sup; a scan
// gives each stream's exact stopping index K; the COMMIT pass then runs exactly the packets
// s < K of every stream with deposits.  Counter-based random numbers (stream c, sequence s ->
// id (c << 40) | s) make the second pass replay the first exactly.
//
// xI_scatt is accumulated in FP64 (global_atomic_add_f64) in a device layout with the flux type
// fastest and padded to one 64-byte line, [icell][psup][phik][iRT][8]: the <= 5 deposits per observer of
// one crossing are ONE L2 line operation when issued by 8 neighbouring lanes of one instruction
// (deposit_rt1_wave).  mcgpu_fetch_xI transposes to the reference's xI_scatt(phik,psup,type,iRT,icell)
// and rounds to default real.
#pragma once
#include "mc_device.hip.h"
#include "mc_xi32.hip.h"

namespace mcgpu {

struct MonoArgs {
  uint64_t seed;
  int lambda, p_lambda, capt_sup, rt1;
  double frac_E_stars, frac_E_disk;
  const double* prob_E_cell;  // (0:n_cells) at this wavelength, or null
  // work items
  unsigned long long n_items;
  const unsigned long long* item_base;  // COMMIT: [n_chunks+1] prefix sums of K
  int n_chunks, first_chunk;
  const int* active;                    // SCOUT: streams still running
  const unsigned long long* seq0;       // SCOUT: [n_chunks] first sequence number of this batch
  unsigned long long batch;             // SCOUT: packets per stream in this batch
  unsigned char* hits;                  // SCOUT: [n_active * batch] 1 = binned in capt_sup
  unsigned long long* hit_count;        // COMMIT (may be null): [n_chunks] packets of the launch binned in capt_sup
  // ray tracing method 1
  int RT_n_incl, nRT;                   // nRT = RT_n_incl * RT_n_az
  const double* rt_u;                   // [nRT] tab_u_rt(ibin,iaz), q = ibin-1 + RT_n_incl*(iaz-1)
  const double* rt_v;
  const double* rt_w;                   // [RT_n_incl]
  int n_az_rt, n_theta_rt, N_type_flux, contrib;
  const float* s11;                     // tab_s11_pos(0:nang, p_lambda)
  double* xI;                           // device layout [n_cells][n_theta_rt][n_az_rt][nRT][XI_LINE] of doubles, or with
  int xI_f32;                           // xI_f32 (mcgpu_set_xI_precision(4)): the PACKED default-real layout (Xi32Lay, above):
  Xi32Lay xi;                           // [sub-bin][xi.binf default reals]
  // ray tracing method 2 (2D): the specific intensity per cell and direction bin
  int rt2, n_theta_I, n_phi_I;
  double* I_spec;                       // device layout [n_cells][n_phi_I][n_theta_I][XI_LINE]
  double* I_spec_star;                  // [n_cells]: unscattered starlight
  // accumulators
  double* sed;
  double* n_sent;
  unsigned long long* counters;
  unsigned long long* next_item;
  int* err;
  int inner_iters, min_active;
  int flags;  // diagnostics: bit 0 = compute the deposits but skip the atomics, bit 1 = only the I deposit
  // the commit pass's deposits as a LOG (k_mono<..., LOG>; "The deposits as a log" below, mc_xilog.hip.h folds it)
  unsigned int* log_keys;               // [log_cap] sub-bin index ((icell-1) n_theta_rt + psup-1) n_az_rt + phik-1, bit 31: flag_star
  unsigned long long* log_vals;         // [log_cap] flight id | path length (default real) << 32
  float* log_rows;                      // [rows_cap][nRT x (4 with Stokes tracking, else 1)] the flights' deposit weights
  unsigned long long* log_ctl;          // [0] records reserved, [1] flights reserved (in blocks, by the waves)
  unsigned long long log_cap, rows_cap;
  unsigned int log_sentinel;            // key of an unused entry: sorts behind every sub-bin
  unsigned long long item_lo;           // COMMIT: the launch runs the work items [item_lo, item_lo + n_items)
  int kf_lds;                           // COMMIT: kappa_factor(1:n_cells) is staged in the workgroup's LDS (mono_lds_bytes)
  int rowf;                             // COMMIT: > 0: the per-lane results are weight rows of this many default reals
};

// per-lane results of angles_scatt_rt1, kept in LDS as [q][thread]
struct RtScratch {
  int* itheta;
  double* cosw;
  double* sinw;
  const double* rot;   // per observer (cost, sint, sing) of rotation(., -u_obs, -v_obs, -w_obs): they depend on the observer only
};

// slim = the workgroup stages only the tables the SED mode reads (lds_carve(..., mono = true)).  (The phase-function
// column is selected by index, interact(..., lds_col): choosing between two LDS pointers there crashed hipcc 7.2.)
// log: the kernel that writes its deposits to the log keeps no per-lane results and no tiles (the flight's weights go
// straight to its row in HBM)
// kf_cells > 0: kappa_factor of that many cells rides along (MonoArgs::kf_lds): a global load in the crossing loop waits
// for EVERY vector-memory operation of the wave issued before it -- loads and the fire-and-forget atomics share one
// in-order counter on this target and return out of order with respect to each other, so the compiler's wait is
// vmcnt(0) -- i.e. for the previous crossing's deposits to come back from the memory side (microseconds under load):
// with the one per-crossing load served from LDS the commit pass's waves never wait for their own atomics.
// rowf > 0 (the commit pass with default-real records and one dust class on cylindrical / spherical grids): the per-lane
// results are the flight's deposit weights as a row of rowf default reals in the sub-bin's order (xi32_row_floats) instead
// of (cosw, sinw, itheta), and the tiles carry no slot masks.
__host__ __device__ inline size_t mono_lds_bytes(const DevModel& M, int nRT, int threads, bool pola, bool slim, bool log = false,
                                                 int kf_cells = 0, int rowf = 0) {
  size_t b = (lds_bytes(M, slim) + 7) / 8 * 8;
  b += (size_t)6 * (M.nang + 1) * sizeof(float);                     // the Mueller columns of p_lambda
  b = (b + 7) / 8 * 8;
  const int nsc = log ? 0 : nRT, tsc = log ? 0 : threads;
  if (rowf > 0 && !log) {
    b = (b + 15) / 16 * 16;
    b += (size_t)rowf * threads * sizeof(float);                     // the flights' weight rows, [chunk of 4][thread]
    b += (size_t)tsc * (8 * sizeof(double) + sizeof(unsigned long long));  // deposit tiles: 16 default reals + address
  } else {
    b += (size_t)nsc * threads * (pola ? 2 * sizeof(double) : 0);      // cosw, sinw
    b += (size_t)tsc * (8 * sizeof(double) + sizeof(unsigned long long));  // deposit tiles: record + address
    b += (size_t)nsc * threads * sizeof(int);                          // itheta
    b += (size_t)tsc * sizeof(unsigned int);                           // deposit tiles: slot mask
  }
  b = (b + 7) / 8 * 8;
  b += (size_t)3 * nRT * sizeof(double);                             // the observers' rotation constants
  b += (size_t)kf_cells * sizeof(double);                            // kappa_factor
  return b;
}

// angles_scatt_rt1 (dust_ray_tracing.f90:409-476) for this lane's direction
// w_mu != nullptr (the commit pass with default-real records and one dust class): the flight's DEPOSIT WEIGHTS are stored
// instead of the angles -- between two interactions the packet's Stokes vector, its direction and hence the product
// RPO . Mueller . ROP . Stokes for every observer are constant, a crossing only multiplies them by its path length
// (calc_xI_scatt_pola, dust_ray_tracing.f90:533-632, evaluated once per flight instead of once per crossing).  Four
// default-real weights take the 16 bytes of (cosw, sinw); without Stokes tracking the one weight takes itheta's 4.
// row != nullptr (the commit pass that LOGS its deposits): the weights go to the flight's row in HBM, [nRT][4] (Stokes
// tracking) or [nRT] default reals, and nothing is kept in LDS.
// (one observer q of the loop; slot = the thread of the workgroup whose per-lane results in LDS receive it: the lane's
// own, or -- angles_scatt_rt1_wave below -- the lane another lane computes for)
// rowimg != nullptr (MonoArgs::rowf > 0): the weights go to the lane's ROW in LDS, in the order of the sub-bin
// (xi32_row_floats; `star`: the packet's origin decides which of the two origins' places holds the flux where they are
// interleaved), so that a crossing stages a line of its deposits as four 16-byte chunks times the path length.
#ifndef MCGPU_LANE_EMULATION
__device__ __forceinline__ void row_put(float* rowimg, int p, int slot, float v) {
  typedef __attribute__((address_space(3))) float lds_f32_t;
  ((lds_f32_t*)rowimg)[(((size_t)(p >> 2) * blockDim.x + slot) << 2) + (p & 3)] = v;
}
#endif
template <bool POLA>
__device__ inline void angles_scatt_rt1_one(const DevModel& M, const MonoArgs& A, const RtScratch& R, int q, int slot, double u,
                                             double v, double w, const float* w_mu, const double* S, float* row,
                                             float* rowimg = nullptr, bool star = false) {
  const double ur = A.rt_u[q], vr = A.rt_v[q], wr = A.rt_w[q % A.RT_n_incl];
  const float cos_scatt = (float)nd_add(nd_add(nd_mul(ur, u), nd_mul(vr, v)), nd_mul(wr, w));
  // k = nint(acos(cos_scatt) * nang / pi) in default real (:430-434).  The default-real arccosine of the runtime decides
  // the bin unless the quotient lies within delta = 1.2e-6 nang of a bin edge (2.2e-4 at 180 bins) -- the roundings of
  // both evaluations together (acosf to 2 ulp of pi, two default-real operations at ~nang, the reference's correctly
  // rounded acos and default-real product) stay below 0.5e-6 nang -- where the reference's expression itself runs
  // (4e-4 of the calls, and NaN): the FP64 acos was 75 of this observer's instructions.
  int k;
  const float qf = acosf(cos_scatt) * ((float)M.nang * 0.318309886183790672f) + 0.5f;
  const float qfl = floorf(qf);
  const float dq = 1.2e-6f * (float)M.nang;
  if (qf - qfl > dq && qf - qfl < 1.0f - dq) k = (int)qfl;
  else {
    const float ac = (float)acos((double)cos_scatt);  // the correctly rounded default-real acos
    if (ac != ac) k = 1;
    else k = (int)llrint(floor(nf_mul(ac, (float)M.nang) / PI + 0.5));  // nint()
  }
  if (k > M.nang) k = M.nang;
  if (k < 1) k = 1;
  if (!row && !rowimg) R.itheta[q * blockDim.x + slot] = k;
#ifndef MCGPU_LANE_EMULATION   // (the lane emulation has no default-real commit pass)
  if (!POLA && w_mu) {
    const float wI = (float)(S[0] * (double)w_mu[k]);
    if (row) row[q] = wI;
    else if (rowimg) {
      const Xi32Lay& X = A.xi;
      if (!X.sum_I) row_put(rowimg, q * X.sA, slot, wI);
      else if (!X.split) { row_put(rowimg, q * X.sA, slot, star ? wI : 0.0f); row_put(rowimg, q * X.sA + 1, slot, star ? 0.0f : wI); }
      else row_put(rowimg, q, slot, wI);
    } else R.itheta[q * blockDim.x + slot] = __float_as_int(wI);
  }
#endif
  if (POLA) {
    // rotation(u, v, w, -ur, -vr, -wr, ...) with the observer's constants from LDS (only y' and z' are needed)
    const double cost = R.rot[3 * q], sint = R.rot[3 * q + 1], sing = R.rot[3 * q + 2];
    const double prod = cost * u + sint * v;
    const double v1pj = cost * v - sint * u;
    const double v1pk = sing * w - (-wr) * prod;
    double xnyp = sqrt(v1pk * v1pk + v1pj * v1pj), costhet;
    if (xnyp < 1e-10) { xnyp = 0.0; costhet = 1.0; }
    else costhet = -1.0 * v1pj / xnyp;
    // theta = acos(costhet) (pi -> 0), omega = 2 (theta + pi / 2), negated below the plane; cos and sin of omega
    // (dust_ray_tracing.f90:452-470) without the acos and the sincos -- 180 of this observer's ~450 instructions:
    // cos(2 theta + pi) = 1 - 2 c^2, sin(2 theta + pi) = -2 c sqrt(1 - c^2); the composition's own rounding apart
    // (1e-16, like update_stokes' rotation since round 4)
    double cosw = 1.0 - 2.0 * costhet * costhet;
    double sinw = -2.0 * costhet * sqrt(fmax(1.0 - costhet * costhet, 0.0));
    if (v1pk < 0.0) sinw = -sinw;
    if (fabs(cosw) < 1e-06) cosw = 0.0;
    if (fabs(sinw) < 1e-06) sinw = 0.0;
    if (!row && !rowimg) {
      R.cosw[q * blockDim.x + slot] = cosw;
      R.sinw[q * blockDim.x + slot] = sinw;
    }
#ifndef MCGPU_LANE_EMULATION
    if (w_mu) {   // (the expressions of deposit_rt1_wave, without the path length)
      const int na1 = M.nang + 1;
      const float s11 = w_mu[k];
      const float s12 = -s11 * w_mu[na1 + k], s22 = s11 * w_mu[2 * na1 + k], s33 = -s11 * w_mu[3 * na1 + k];
      const float s34 = -s11 * w_mu[4 * na1 + k], s44 = -s11 * w_mu[5 * na1 + k];
      const double C1 = S[0], C4 = S[3];
      const double C2 = cosw * S[1] + (-sinw) * S[2];
      const double C3 = sinw * S[1] + cosw * S[2];
      const double D1 = (double)s11 * C1 + (double)s12 * C2;
      const double D2 = (double)s12 * C1 + (double)s22 * C2;
      const double D3 = (double)s33 * C3 + (double)(-s34) * C4;
      const double D4 = (double)s34 * C3 + (double)s44 * C4;
      if (row) {
        reinterpret_cast<float4*>(row)[q] = make_float4((float)D1, (float)((-cosw) * D2 + (-sinw) * D3), (float)((-sinw) * D2 + cosw * D3), (float)D4);
      } else if (rowimg) {
        const Xi32Lay& X = A.xi;
        const float f0 = (float)D1, f1 = (float)((-cosw) * D2 + (-sinw) * D3), f2 = (float)((-sinw) * D2 + cosw * D3), f3 = (float)D4;
        if (!X.sum_I) {
          const int b = q * X.sA;
          row_put(rowimg, b, slot, f0); row_put(rowimg, b + 1, slot, f1); row_put(rowimg, b + 2, slot, f2); row_put(rowimg, b + 3, slot, f3);
        } else {
          const int b = q * X.sA;       // (interleaved: sA = 5; split: sA = 3)
          row_put(rowimg, b, slot, f1); row_put(rowimg, b + 1, slot, f2); row_put(rowimg, b + 2, slot, f3);
          if (X.split) row_put(rowimg, A.nRT * 3 + q, slot, f0);
          else { row_put(rowimg, b + 3, slot, star ? f0 : 0.0f); row_put(rowimg, b + 4, slot, star ? 0.0f : f0); }
        }
      } else {
        float2* wc = reinterpret_cast<float2*>(R.cosw) + (q * blockDim.x + slot);
        float2* ws = reinterpret_cast<float2*>(R.sinw) + (q * blockDim.x + slot);
        *wc = make_float2((float)D1, (float)((-cosw) * D2 + (-sinw) * D3));
        *ws = make_float2((float)((-sinw) * D2 + cosw * D3), (float)D4);
      }
    }
#endif
  }
}

template <bool POLA>
__device__ inline void angles_scatt_rt1(const DevModel& M, const MonoArgs& A, const RtScratch& R, double u,
                                        double v, double w, const float* w_mu = nullptr, const double* S = nullptr, float* row = nullptr,
                                        float* rowimg = nullptr, bool star = false) {
  for (int q = 0; q < A.nRT; ++q) angles_scatt_rt1_one<POLA>(M, A, R, q, (int)threadIdx.x, u, v, w, w_mu, S, row, rowimg, star);
}

// One pending deposit of a lane: where (cell, azimuth / elevation sub-bin) and how long the path was.
struct RtDeposit {
  bool on;
  int icell, phik, psup;
  double l;
};

// sub-bin of a path (radiation_field.f90:64-83)
template <bool L3D>
__device__ inline void rt1_subbin(const MonoArgs& A, double x0, double y0, double z0, double x1, double y1,
                                  double z1, int& phik, int& psup) {
  phik = 1; psup = 1;
  if (!L3D) {
    const double xm = 0.5 * (x0 + x1), ym = 0.5 * (y0 + y1), zm = 0.5 * (z0 + z1);
    const double phi_pos = atan2(xm, ym);
    phik = (int)floor(modulo_d(phi_pos, 2 * PI) / (2 * PI) * (double)A.n_az_rt) + 1;
    if (phik > A.n_az_rt) phik = A.n_az_rt;
    psup = (zm > 0.0) ? 1 : 2;
  }
}

// the same from plain arguments (the ray tracer has no MonoArgs)
__device__ inline void rt1_subbin_of(int n_az_rt, bool l3D, double x0, double y0, double z0, double x1, double y1,
                                     double z1, int& phik, int& psup) {
  phik = 1; psup = 1;
  if (!l3D) {
    const double xm = 0.5 * (x0 + x1), ym = 0.5 * (y0 + y1), zm = 0.5 * (z0 + z1);
    const double phi_pos = atan2(xm, ym);
    phik = (int)floor(modulo_d(phi_pos, 2 * PI) / (2 * PI) * (double)n_az_rt) + 1;
    if (phik > n_az_rt) phik = n_az_rt;
    psup = (zm > 0.0) ? 1 : 2;
  }
}

constexpr int XI_LINE = 8;  // doubles per (cell, sub-bin, observer) record of the device layout: one 64-byte line

// save_radiation_field, lscatt_ray_tracing2 branch (radiation_field.f90:91-129; 2D only): the direction bin of a path --
// the azimuth of the packet's direction relative to the azimuth of the path's midpoint, and cos(theta) mirrored below
// the midplane
__device__ inline void rt2_bins(const MonoArgs& A, double x0, double y0, double z0, double x1, double y1, double z1,
                                double u, double v, double w, int& theta_I, int& phi_I) {
  const double xm = 0.5 * (x0 + x1), ym = 0.5 * (y0 + y1), zm = 0.5 * (z0 + z1);
  const double phi_pos = atan2(xm, ym);
  const double phi_vol = atan2(-u, -v) + 2 * PI;  // two_pi ensures phi_vol > phi_pos
  phi_I = (int)floor(modulo_d(phi_vol - phi_pos, 2 * PI) / (2 * PI) * (double)A.n_phi_I) + 1;
  if (phi_I > A.n_phi_I) phi_I = 1;
  if (zm > 0.0) theta_I = (int)floor(0.5 * (w + 1.0) * (double)A.n_theta_I) + 1;
  else theta_I = (int)floor(0.5 * (-w + 1.0) * (double)A.n_theta_I) + 1;
  if (theta_I > A.n_theta_I) theta_I = A.n_theta_I;
}

// I_spec(1:n_Stokes, theta_I, phi_I, icell) += l * Stokes (and the copy of I in the slot of its origin with
// lsepar_contrib), or I_spec_star(icell) += l * Stokes(1) for starlight that has not interacted yet: one record of one
// 64-byte line per crossing -- against one per observer with method 1
// (called by the whole wavefront: `on` = this lane deposits; the records go out through the wave's tile like method 1's)
template <bool POLA>
__device__ inline void deposit_rt2_wave(const MonoArgs& A, bool on, int icell, int theta_I, int phi_I, double l,
                                        const double S[4], bool flag_star, bool direct, double* tile,
                                        unsigned long long* tile_addr, unsigned int* tile_mask);

// save_radiation_field, lscatt_ray_tracing1 branch (radiation_field.f90:63-89) with calc_xI_scatt
// (dust_ray_tracing.f90:480-529) / calc_xI_scatt_pola (:533-632), for the whole wavefront.
// mu = the six Mueller columns [s11 | s12/s11 | s22/s11 | s33/s11 | s34/s11 | s44/s11] of p_lambda in LDS.
//
// FP64 atomics to scattered addresses are bound by L2 line operations (2.4e10 /s on MI355X, measured:
// tools/atomic_line_bench.hip), and a line operation costs the same whether one or eight lanes of the
// instruction hit that line.  A lane's <= 5 values for one observer share one 64-byte record, so the wave
// transposes them through a per-wave LDS tile: lane j stores its record, then in round r the 8 lanes of
// group g = lane/8 add the 8 slots of lane 8r+g's record -- one instruction, 8 records, 8 line
// operations instead of 40.
#ifndef MCGPU_LANE_EMULATION
// The wave's tile, explicitly in LDS, and the records' addresses, explicitly global.  Round 4 found what the SED mode's
// commit pass waited for (`wait_frac` 0.56 at 42 % of the atomic-line rate): the tile was read and written through
// `volatile` generic pointers -- LLVM's address-space inference leaves volatile accesses alone -- so every tile access
// was a FLAT load / store with `sc0 sc1` and an `s_waitcnt vmcnt(0)` behind it, and the atomics (addresses that came out
// of the tile as integers: generic again) were flat atomics: each tile access waited for the previous atomic to COMPLETE
// (~2-3 us under load).  Now: ds_read / ds_write, global atomics, the order between the lanes' writes and reads kept by
// the wave barrier (one wave executes its LDS instructions in order) and a compiler fence; the reads of a round are
// issued together, ahead of its atomics.
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) unsigned int lds_u32;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(1))) double glb_f64;
typedef __attribute__((address_space(1))) float glb_f32;
__device__ __forceinline__ void tile_sync() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}
constexpr int TILE_UNROLL = 8;   // rounds whose tile reads are in flight together (2 / 4 / 8 / 16 measured: profiles/r06_sed_tile_unroll_ab.log)

// The records of one instruction's lanes (<= 4 Stokes values + the copy of I in the slot of its origin, all in one
// 64-byte line per lane) go out through the wave's LDS tile: the lanes that deposit stage their record in consecutive
// places, then K lanes serve each record -- floor(64 / K) records per atomic instruction instead of one value of 64.
__device__ inline void wave_deposit_records(int lane, int K, unsigned int mask, int cslot, double* rec, double v0, double v1,
                                            double v2, double v3, double* tile_g, unsigned long long* tile_addr_g,
                                            unsigned int* tile_mask_g, bool contrib) {
  lds_f64* const tile = (lds_f64*)tile_g;
  lds_u64* const tile_addr = (lds_u64*)tile_addr_g;
  lds_u32* const tile_mask = (lds_u32*)tile_mask_g;
  const int NR = 64 / K;
  const int rl = lane / K, j = lane - rl * K;
  const bool lane_used = rl < NR;
  const bool is_contrib = contrib && j == K - 1;
  const unsigned long long any = __ballot(mask != 0);
  const int n_act = __popcll(any);
  if (mask) {
    const int place = __popcll(any & ((1ull << lane) - 1ull));
    lds_f64* my = tile + place * XI_LINE;
    my[0] = v0; my[1] = v1; my[2] = v2; my[3] = v3;
    tile_addr[place] = (unsigned long long)rec;
    tile_mask[place] = mask | ((unsigned int)cslot << 8);
  }
  tile_sync();
  for (int r0 = 0; r0 < n_act; r0 += TILE_UNROLL * NR) {
    unsigned int mw[TILE_UNROLL];
    unsigned long long ad[TILE_UNROLL];
    double val[TILE_UNROLL];
#pragma unroll
    for (int t = 0; t < TILE_UNROLL; ++t) {
      const int src = r0 + t * NR + rl;
      const bool ok = lane_used && src < n_act;
      const int sc = ok ? src : 0;
      const unsigned int m_t = tile_mask[sc];
      mw[t] = ok ? m_t : 0u;
      ad[t] = tile_addr[sc];
      val[t] = tile[sc * XI_LINE + (is_contrib ? 0 : j)];
    }
#pragma unroll
    for (int t = 0; t < TILE_UNROLL; ++t) {
      const int slot = is_contrib ? (int)(mw[t] >> 8) : j;
      if ((mw[t] >> slot) & 1u) atomic_add_f64((double*)((glb_f64*)ad[t] + slot), val[t]);
    }
  }
  tile_sync();
}
#endif

#ifndef MCGPU_LANE_EMULATION
// angles_scatt_rt1 for the lanes of a wave that start a flight (`need`), computed by ALL its lanes (round 6).  The loop
// over the observers costs ~450 instructions each and runs, in the single-role loop of mono_body, for the third of the
// lanes that have just interacted while the others wait: here the (lane, observer) pairs are spread over the 64 lanes --
// the starting lanes put their direction and Stokes vector into the wave's deposit tile (free between deposits), every
// lane takes pairs i = lane, lane + 64, ... and writes the result into the per-lane LDS results of the lane it computed
// for.  Same expressions on the same values (angles_scatt_rt1_one): the results do not depend on who computes them.
// w_mu as in angles_scatt_rt1 (the commit pass with default-real records and one dust class).
template <bool POLA>
__device__ inline void angles_scatt_rt1_wave(const DevModel& M, const MonoArgs& A, const RtScratch& R, bool need, double u, double v,
                                             double w, const float* w_mu, const double S[4], double* tile_g, float* rowimg = nullptr,
                                             bool star = false) {
  const int lane = threadIdx.x & 63;
  const unsigned long long m = __ballot(need);
  if (m == 0ull) return;
  lds_f64* const t = (lds_f64*)tile_g;       // 64 places x XI_LINE doubles
  if (need) {
    lds_f64* my = t + __popcll(m & ((1ull << lane) - 1ull)) * XI_LINE;
    my[0] = u; my[1] = v; my[2] = w; my[3] = S[0]; my[4] = POLA ? S[1] : 0.0; my[5] = POLA ? S[2] : 0.0; my[6] = POLA ? S[3] : 0.0;
    my[7] = (double)((int)threadIdx.x | (star ? 0x10000 : 0));   // (whose results these are, and the packet's origin)
  }
  tile_sync();
  const int n_items = __popcll(m) * A.nRT;
  for (int i = lane; i < n_items; i += 64) {
    const int j = i / A.nRT, q = i - j * A.nRT;
    const lds_f64* p = t + j * XI_LINE;
    const double Sj[4] = {p[3], p[4], p[5], p[6]};
    const int who = (int)p[7];
    angles_scatt_rt1_one<POLA>(M, A, R, q, who & 0xFFFF, p[0], p[1], p[2], w_mu, Sj, nullptr, rowimg, (who & 0x10000) != 0);
  }
  tile_sync();
}
#endif

template <bool POLA>
__device__ inline void deposit_rt2_wave(const MonoArgs& A, bool on, int icell, int theta_I, int phi_I, double l,
                                        const double S[4], bool flag_star, bool direct, double* tile,
                                        unsigned long long* tile_addr, unsigned int* tile_mask) {
  if (on && direct) atomic_add_f64(&A.I_spec_star[icell - 1], l * S[0]);
  const bool rec_on = on && !direct;
  double* rec = A.I_spec + ((((size_t)(rec_on ? icell - 1 : 0) * A.n_phi_I + (phi_I - 1)) * A.n_theta_I) + (theta_I - 1)) * XI_LINE;
  const int cslot = A.contrib ? (POLA ? 4 : 1) + (flag_star ? 1 : 3) : 0;  // n_Stokes + 2 / + 4, 1-based
  const unsigned int mask = rec_on ? ((POLA ? 0xFu : 1u) | (A.contrib ? 1u << cslot : 0u)) : 0u;
#ifdef MCGPU_LANE_EMULATION
  (void)tile; (void)tile_addr; (void)tile_mask;
  if (rec_on) {
    atomic_add_f64(rec, l * S[0]);
    if (POLA) { atomic_add_f64(rec + 1, l * S[1]); atomic_add_f64(rec + 2, l * S[2]); atomic_add_f64(rec + 3, l * S[3]); }
    if (A.contrib) atomic_add_f64(rec + cslot, l * S[0]);
  }
#else
  if (__ballot(mask != 0) == 0ull) return;
  wave_deposit_records(threadIdx.x & 63, (POLA ? 4 : 1) + (A.contrib ? 1 : 0), mask, cslot, rec, l * S[0], POLA ? l * S[1] : 0.0,
                       POLA ? l * S[2] : 0.0, POLA ? l * S[3] : 0.0, tile, tile_addr, tile_mask, A.contrib != 0);
#endif
}

// lvariable_dust (M.n_classes): the columns are those of the cell's class, tab_s11_pos(it, p_icell, p_lambda) etc.
// (dust_ray_tracing.f90:503-512), gathered from the per-class tables in HBM instead of the LDS copy.
__device__ inline size_t mono_class_col(const DevModel& M, const MonoArgs& A, int icell) {
  return M.n_classes ? ((size_t)M.cell_class[icell - 1] * M.n_lambda + (A.p_lambda - 1)) * (size_t)(M.nang + 1) : 0;
}
#ifdef MCGPU_LANE_EMULATION
#define MONO_MU(tbl, vtab) (var ? (vtab)[vcol + it] : mu[(tbl) * na1 + it])
#else
// (the class tables are global, the columns of p_lambda sit in LDS: two loads in two address spaces, selected by VALUE --
// a select of the pointers made every read a flat load, whose `vmcnt` wait also waits for the atomics in flight)
#define MONO_MU(tbl, vtab) (var ? (vtab)[vcol + it] : ((const lds_f32*)mu)[(tbl) * na1 + it])
#endif

template <bool POLA>
__device__ inline void deposit_rt1_wave(const DevModel& M, const MonoArgs& A, const RtScratch& R, const float* mu,
                                        const RtDeposit& D, const double S[4], bool flag_star, double* tile,
                                        unsigned long long* tile_addr, unsigned int* tile_mask) {
  const int na1 = M.nang + 1;
  const int lane = threadIdx.x & 63;
  const bool var = M.n_classes != 0;
  const size_t vcol = D.on ? mono_class_col(M, A, D.icell) : 0;
#ifdef MCGPU_LANE_EMULATION
  (void)tile; (void)tile_addr; (void)tile_mask;
#else
  // the values a record receives: I (and Q, U, V with Stokes tracking) plus, with lsepar_contrib, the copy of I in the
  // slot of its origin.  An atomic instruction costs the wave about the same whatever its lanes do (measured: half the
  // records per instruction = 1.7x the time), so the 64 lanes of one instruction serve floor(64 / K) records with K
  // lanes each: 12 records instead of 8 with Stokes tracking and contributions, 32 without Stokes tracking.
  const int K = (POLA ? 4 : 1) + (A.contrib ? 1 : 0);
#endif
  for (int q = 0; q < A.nRT; ++q) {
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    unsigned int mask = 0;
    int cslot = 0;
    double* rec = nullptr;
    if (D.on) {
      rec = A.xI + ((((size_t)(D.icell - 1) * A.n_theta_rt + (D.psup - 1)) * A.n_az_rt + (D.phik - 1)) * A.nRT + q) * XI_LINE;
      const int it = R.itheta[q * blockDim.x + threadIdx.x];
      const float s11 = MONO_MU(0, M.v_s11);
      if (!POLA) {
        v0 = D.l * S[0] * (double)s11;
        mask = 1u;
        if (A.contrib) { cslot = flag_star ? 2 : 4; mask |= 1u << cslot; }  // n_Stokes + 2 / + 4, n_Stokes = 1
      } else {
        const float s12 = -s11 * MONO_MU(1, M.v_s12), s22 = s11 * MONO_MU(2, M.v_s22), s33 = -s11 * MONO_MU(3, M.v_s33);
        const float s34 = -s11 * MONO_MU(4, M.v_s34), s44 = -s11 * MONO_MU(5, M.v_s44);
        const double cosw = R.cosw[q * blockDim.x + threadIdx.x], sinw = R.sinw[q * blockDim.x + threadIdx.x];
        const double C1 = S[0], C4 = S[3];
        const double C2 = cosw * S[1] + (-sinw) * S[2];
        const double C3 = sinw * S[1] + cosw * S[2];
        const double D1 = (double)s11 * C1 + (double)s12 * C2;
        const double D2 = (double)s12 * C1 + (double)s22 * C2;
        const double D3 = (double)s33 * C3 + (double)(-s34) * C4;
        const double D4 = (double)s34 * C3 + (double)s44 * C4;
        v0 = D.l * D1;
        v1 = D.l * ((-cosw) * D2 + (-sinw) * D3);
        v2 = D.l * ((-sinw) * D2 + cosw * D3);
        v3 = D.l * D4;
        mask = 0xFu;
        if (A.contrib) { cslot = flag_star ? 5 : 7; mask |= 1u << cslot; }
      }
      if (MCGPU_DIAG(A.flags, 1)) mask = 0;            // diagnostics: compute, do not deposit
      else if (MCGPU_DIAG(A.flags, 2)) mask &= 1u;     // diagnostics: only the I deposit
    }
#ifdef MCGPU_LANE_EMULATION
    if (A.xI_f32 && D.on) {  // (the CPU emulation has one lane and no tile: the default-real layout, value by value)
      float* bin32 = reinterpret_cast<float*>(A.xI) +
          (((size_t)(D.icell - 1) * A.n_theta_rt + (D.psup - 1)) * A.n_az_rt + (D.phik - 1)) * A.xi.binf;
      const int nS = POLA ? 4 : 1;
      const float v[4] = {(float)v0, (float)v1, (float)v2, (float)v3};
      for (int t = 0; t < nS; ++t) {
        const int o = xi32_offset(A.xi, q, t, nS);       // (-2: I where it is the sum of the origins -- not stored)
        if (o >= 0 && ((mask >> t) & 1u)) atomicAdd(bin32 + o, v[t]);
      }
      if (cslot && ((mask >> cslot) & 1u)) atomicAdd(bin32 + xi32_offset(A.xi, q, cslot, nS), (float)v0);
    } else {
      if (mask & 1u) atomic_add_f64(rec, v0);
      if (POLA && (mask & 2u)) atomic_add_f64(rec + 1, v1);
      if (POLA && (mask & 4u)) atomic_add_f64(rec + 2, v2);
      if (POLA && (mask & 8u)) atomic_add_f64(rec + 3, v3);
      if (cslot && ((mask >> cslot) & 1u)) atomic_add_f64(rec + cslot, v0);
    }
#else
    wave_deposit_records(lane, K, mask, cslot, rec, v0, v1, v2, v3, tile, tile_addr, tile_mask, A.contrib != 0);
#endif
  }
}

#ifndef MCGPU_LANE_EMULATION
// The same with default-real records (mcgpu_set_xI_precision(4), the type of the reference's own array) in the PACKED
// layout (Xi32Lay): the sub-bin's observers side by side, so the wave goes through the sub-bin LINE BY LINE -- per line the
// lanes whose deposits reach it stage the 16 values of their crossing that fall on it (the tile: 64 places x 16 default
// reals), then 16 lanes serve each staged place: four places, four line operations per atomic instruction, every value of
// a line added by ONE instruction.  In the split arrangement a line that holds only one origin's values is staged by the
// packets of that origin alone.  Lines per crossing at ten observers with Stokes tracking and contributions: 3 (round 4's
// pairs of padded records: 5; the interleaved records with I: 4).  var: the weights depend on the cell's dust class and
// are computed per crossing; otherwise they are the flight's (angles_scatt_rt1 left them in LDS) times the path length.
template <bool POLA>
__device__ inline void deposit_rt1_wave_f32(const DevModel& M, const MonoArgs& A, const RtScratch& R, const float* mu,
                                            const RtDeposit& D, const double S[4], bool flag_star, double* tile,
                                            unsigned long long* tile_addr, unsigned int* tile_mask) {
  const int na1 = M.nang + 1;
  const int lane = threadIdx.x & 63;
  const bool var = M.n_classes != 0;
  const size_t vcol = D.on ? mono_class_col(M, A, D.icell) : 0;
  const Xi32Lay X = A.xi;
  const int n_lines = X.binf >> 4, n_stokes = A.nRT * X.nA;
  const bool on = D.on && !MCGPU_DIAG(A.flags, 1);
  if (__ballot(on) == 0ull) return;
  const unsigned long long lt = (1ull << lane) - 1ull;
  lds_f32* const tile32 = (lds_f32*)tile;            // (explicit address spaces, tile_sync: see wave_deposit_records)
  lds_u64* const taddr = (lds_u64*)tile_addr;
  (void)tile_mask;
  const size_t bin = on ? ((size_t)(D.icell - 1) * A.n_theta_rt + (D.psup - 1)) * A.n_az_rt + (D.phik - 1) : 0;
  float* const bin32 = reinterpret_cast<float*>(A.xI) + bin * (size_t)X.binf;
  const int sr = lane >> 4, sf = lane & 15;   // serving: place 4 g + sr of a group, value sf of the line
  const float lf = (float)D.l;
  const bool quv = POLA && !MCGPU_DIAG(A.flags, 2), origins = X.oS >= 0 && !MCGPU_DIAG(A.flags, 2);
  // observer q's deposits I, Q, U, V of this crossing
  auto values = [&](int q, float& v0, float& v1, float& v2, float& v3) {
    v1 = 0.0f; v2 = 0.0f; v3 = 0.0f;
    if (!var) {   // the flight's weights (angles_scatt_rt1) times this crossing's path length
      if (!POLA) v0 = lf * __int_as_float(R.itheta[q * blockDim.x + threadIdx.x]);
      else {
        const float2 wc = reinterpret_cast<const float2*>(R.cosw)[q * blockDim.x + threadIdx.x];
        const float2 ws = reinterpret_cast<const float2*>(R.sinw)[q * blockDim.x + threadIdx.x];
        v0 = lf * wc.x; v1 = lf * wc.y; v2 = lf * ws.x; v3 = lf * ws.y;
      }
    } else {
      const int it = R.itheta[q * blockDim.x + threadIdx.x];
      const float s11 = MONO_MU(0, M.v_s11);
      if (!POLA) v0 = (float)(D.l * S[0] * (double)s11);
      else {
        const float s12 = -s11 * MONO_MU(1, M.v_s12), s22 = s11 * MONO_MU(2, M.v_s22), s33 = -s11 * MONO_MU(3, M.v_s33);
        const float s34 = -s11 * MONO_MU(4, M.v_s34), s44 = -s11 * MONO_MU(5, M.v_s44);
        const double cosw = R.cosw[q * blockDim.x + threadIdx.x], sinw = R.sinw[q * blockDim.x + threadIdx.x];
        const double C1 = S[0], C4 = S[3];
        const double C2 = cosw * S[1] + (-sinw) * S[2];
        const double C3 = sinw * S[1] + cosw * S[2];
        const double D1 = (double)s11 * C1 + (double)s12 * C2;
        const double D2 = (double)s12 * C1 + (double)s22 * C2;
        const double D3 = (double)s33 * C3 + (double)(-s34) * C4;
        const double D4 = (double)s34 * C3 + (double)s44 * C4;
        v0 = (float)(D.l * D1);
        v1 = (float)(D.l * ((-cosw) * D2 + (-sinw) * D3));
        v2 = (float)(D.l * ((-sinw) * D2 + cosw * D3));
        v3 = (float)(D.l * D4);
      }
    }
  };
  for (int line = 0; line < n_lines; ++line) {
    const int g0 = line << 4;
    // who reaches this line: in the split arrangement the lines behind the Stokes values hold one origin each
    bool mine = on;
    if (X.split) {
      if (g0 >= X.oT) mine = on && !flag_star;
      else if (g0 >= n_stokes) mine = on && flag_star;
    }
    const unsigned long long any = __ballot(mine);
    const int n_act = __popcll(any);
    if (n_act == 0) continue;
    if (mine) {
      typedef float f32x4_t __attribute__((ext_vector_type(4)));
      typedef __attribute__((address_space(3))) f32x4_t lds_f32x4;
      const f32x4_t z4 = {0.0f, 0.0f, 0.0f, 0.0f};
      const int place = __popcll(any & lt);
      taddr[place] = (unsigned long long)(bin32 + g0);
      lds_f32* const my = tile32 + place * 16;
      *(lds_f32x4*)(my) = z4; *(lds_f32x4*)(my + 4) = z4; *(lds_f32x4*)(my + 8) = z4; *(lds_f32x4*)(my + 12) = z4;
      if (!X.split || g0 < n_stokes) {   // the observers whose (Stokes) values fall on this line (g: position in the line)
        const int q_lo = g0 / X.sA;
        int q_hi = (g0 + 15) / X.sA;
        if (q_hi > A.nRT - 1) q_hi = A.nRT - 1;
        for (int q = q_lo; q <= q_hi; ++q) {
          float v0, v1, v2, v3;
          values(q, v0, v1, v2, v3);
          int g = q * X.sA - g0;
          if (!X.sum_I) { if (g >= 0 && g < 16) my[g] = v0; ++g; }
          if (POLA) {
            if (quv) {
              if (g >= 0 && g < 16) my[g] = v1;
              if (g + 1 >= 0 && g + 1 < 16) my[g + 1] = v2;
              if (g + 2 >= 0 && g + 2 < 16) my[g + 2] = v3;
            }
            g += 3;
          }
          if (origins && !X.split) {   // interleaved: the flux in the place of its origin, star then thermal
            const int gc = g + (flag_star ? 0 : 1);
            if (gc >= 0 && gc < 16) my[gc] = v0;
          }
        }
      }
      if (origins && X.split) {        // split: the origin's values that fall on this line
        const int o = (flag_star ? X.oS : X.oT) - g0;   // place of observer 0's value relative to the line
        int q_lo = -o, q_hi = 15 - o;
        if (q_lo < 0) q_lo = 0;
        if (q_hi > A.nRT - 1) q_hi = A.nRT - 1;
        for (int q = q_lo; q <= q_hi; ++q) {
          float v0, v1, v2, v3;
          values(q, v0, v1, v2, v3);
          my[o + q] = v0;
        }
      }
    }
    tile_sync();
    for (int p0 = 0; p0 < n_act; p0 += 4 * TILE_UNROLL) {
      unsigned long long ad[TILE_UNROLL];
      float val[TILE_UNROLL];
#pragma unroll
      for (int t = 0; t < TILE_UNROLL; ++t) {
        const int src = p0 + 4 * t + sr;
        const bool ok = src < n_act;
        const int sc = ok ? src : 0;
        ad[t] = taddr[sc];
        const float v = tile32[sc * 16 + sf];
        val[t] = ok ? v : 0.0f;
      }
#pragma unroll
      for (int t = 0; t < TILE_UNROLL; ++t)
        if (val[t] != 0.0f) atomicAdd((float*)((glb_f32*)ad[t] + sf), val[t]);
    }
    tile_sync();
  }
}
#endif

#ifndef MCGPU_LANE_EMULATION
// The same deposits from the flight's weight ROW (MonoArgs::rowf > 0: default-real records, one dust class; round 6).
// Measured with the atomic instructions compiled out (profiles/r06_sed_deposit_cost.log): staging and serving the lines
// cost three times the transport itself and the atomics only 5-10 % on top -- the pass was bound by the instructions of
// deposit_rt1_wave_f32's staging (per observer two LDS reads, four products and five placed stores).  Here the weights
// already lie in the sub-bin's order (angles_scatt_rt1_one, row_put), so a line of a crossing's deposits is FOUR 16-byte
// chunks of the row times the path length, stored as four 16-byte chunks of the tile; the values of the stellar origin a
// thermal packet shares a line with are cleared, and its own origin's values -- the same numbers -- are placed on the
// thermal lines one by one.  The serving half is deposit_rt1_wave_f32's (a value of zero is added like any other: it
// shares its line with values that are not).
__device__ inline void deposit_rt1_wave_row(const MonoArgs& A, const float* row_g, const RtDeposit& D, bool flag_star, double* tile,
                                            unsigned long long* tile_addr) {
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) f32x4_t lds_f32x4;
  const int lane = threadIdx.x & 63;
  const Xi32Lay X = A.xi;
  const int n_lines = X.binf >> 4, n_stokes = A.nRT * X.nA, n_chunks = A.rowf >> 2;
  const bool on = D.on && !MCGPU_DIAG(A.flags, 1);
  if (__ballot(on) == 0ull) return;
  const unsigned long long lt = (1ull << lane) - 1ull;
  lds_f32* const tile32 = (lds_f32*)tile;            // (explicit address spaces, tile_sync: see wave_deposit_records)
  lds_u64* const taddr = (lds_u64*)tile_addr;
  const lds_f32x4* const row4 = (const lds_f32x4*)row_g + threadIdx.x;   // chunk c of this lane: row4[c * blockDim.x]
  const lds_f32* const row1 = (const lds_f32*)row_g;
  const size_t bin = on ? ((size_t)(D.icell - 1) * A.n_theta_rt + (D.psup - 1)) * A.n_az_rt + (D.phik - 1) : 0;
  float* const bin32 = reinterpret_cast<float*>(A.xI) + bin * (size_t)X.binf;
  const int sr = lane >> 4, sf = lane & 15;   // serving: place 4 g + sr of a group, value sf of the line
  const float lf = (float)D.l;
  for (int line = 0; line < n_lines; ++line) {
    const int g0 = line << 4;
    // who reaches this line: in the split arrangement the lines behind the Stokes values hold one origin each
    bool mine = on;
    const bool t_line = X.split && g0 >= X.oT;
    if (X.split) {
      if (t_line) mine = on && !flag_star;
      else if (g0 >= n_stokes) mine = on && flag_star;
    }
    const unsigned long long any = __ballot(mine);
    const int n_act = __popcll(any);
    if (n_act == 0) continue;
    if (mine) {
      const int place = __popcll(any & lt);
      taddr[place] = (unsigned long long)(bin32 + g0);
      lds_f32x4* const my4 = (lds_f32x4*)(tile32 + place * 16);
      if (!t_line) {
        const int c0 = g0 >> 2;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          f32x4_t v = {0.0f, 0.0f, 0.0f, 0.0f};
          if (c0 + c < n_chunks) {
            v = row4[(size_t)(c0 + c) * blockDim.x];
            v *= lf;
            if (X.split) {   // (the stellar origin's values on a line a thermal packet reaches for its Stokes values)
              const int p0 = g0 + 4 * c;
              if (p0 + 3 >= n_stokes && !flag_star) {
                if (p0 >= n_stokes) v.x = 0.0f;
                if (p0 + 1 >= n_stokes) v.y = 0.0f;
                if (p0 + 2 >= n_stokes) v.z = 0.0f;
                if (p0 + 3 >= n_stokes) v.w = 0.0f;
              }
            }
          }
          my4[c] = v;
        }
      } else {   // a thermal line: observer q's flux at oT + q, the row holds it at n_stokes + q
        const int q0 = g0 - X.oT;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          f32x4_t v = {0.0f, 0.0f, 0.0f, 0.0f};
          const int q = q0 + 4 * c;
          if (q < A.nRT) {
            auto w = [&](int qq) {
              const int p = n_stokes + qq;
              return qq < A.nRT ? lf * row1[(((size_t)(p >> 2) * blockDim.x + threadIdx.x) << 2) + (p & 3)] : 0.0f;
            };
            v.x = w(q); v.y = w(q + 1); v.z = w(q + 2); v.w = w(q + 3);
          }
          my4[c] = v;
        }
      }
    }
    tile_sync();
    for (int p0 = 0; p0 < n_act; p0 += 4 * TILE_UNROLL) {
      unsigned long long ad[TILE_UNROLL];
      float val[TILE_UNROLL];
      bool ok[TILE_UNROLL];
#pragma unroll
      for (int t = 0; t < TILE_UNROLL; ++t) {
        const int src = p0 + 4 * t + sr;
        ok[t] = src < n_act;
        const int sc = ok[t] ? src : 0;
        ad[t] = taddr[sc];
        val[t] = tile32[sc * 16 + sf];
      }
#pragma unroll
      for (int t = 0; t < TILE_UNROLL; ++t)
        if (ok[t] && !MCGPU_DIAG(A.flags, 4)) atomicAdd((float*)((glb_f32*)ad[t] + sf), val[t]);
    }
    tile_sync();
  }
}
#endif

// LDS of a SED-mode workgroup after the shared tables: the Mueller columns of p_lambda, the per-lane
// results of angles_scatt_rt1 and the per-wave deposit tiles (mono_lds_bytes is the matching size).
struct MonoLds {
  float* mu;
  RtScratch R;
  double* tile;
  unsigned long long* tile_addr;
  unsigned int* tile_mask;
  const double* kf;     // kappa_factor in LDS (MonoArgs::kf_lds), else null
  float* row;           // the weight rows (MonoArgs::rowf > 0), [chunk][thread][4], else null
};

template <bool POLA>
__device__ inline MonoLds mono_lds_setup(const DevModel& M, const MonoArgs& A, double* lds_base, bool slim, bool log = false) {
  MonoLds L;
  const int na1 = M.nang + 1;
  const size_t nsc = log ? 0 : (size_t)A.nRT, tsc = log ? 0 : (size_t)blockDim.x;   // (mono_lds_bytes)
  L.mu = reinterpret_cast<float*>(lds_base + (lds_bytes(M, slim) + 7) / 8);
  double* p = lds_base + (lds_bytes(M, slim) + 7) / 8 + ((size_t)6 * na1 * sizeof(float) + 7) / 8;
  L.row = nullptr;
  size_t used;   // bytes from lds_base to the end of the per-lane arrays
  if (A.rowf > 0 && !log) {
    p = lds_base + ((size_t)(p - lds_base) * 8 + 15) / 16 * 2;   // (16-byte chunks)
    L.row = reinterpret_cast<float*>(p);
    L.R.cosw = nullptr; L.R.sinw = nullptr; L.R.itheta = nullptr; L.tile_mask = nullptr;
    {  // (the row's padding stays zero for the whole launch: only the weights' places are ever written)
      float* r = L.row;
      for (size_t i = threadIdx.x; i < (size_t)A.rowf * blockDim.x; i += blockDim.x) r[i] = 0.0f;
    }
    p += (size_t)A.rowf * blockDim.x / 2;
    L.tile = p + (size_t)(threadIdx.x >> 6) * 64 * XI_LINE;
    p += tsc * XI_LINE;
    L.tile_addr = reinterpret_cast<unsigned long long*>(p) + (size_t)(threadIdx.x >> 6) * 64;
    p += tsc;
    used = (size_t)(p - lds_base) * 8;
  } else {
  L.R.cosw = p;
  L.R.sinw = p + (POLA ? nsc * blockDim.x : 0);
  p += (POLA ? (size_t)2 * nsc * blockDim.x : 0);
  L.tile = p + (log ? 0 : (size_t)(threadIdx.x >> 6) * 64 * XI_LINE);  // this wave's 64 x 8 doubles
  p += tsc * XI_LINE;
  L.tile_addr = reinterpret_cast<unsigned long long*>(p) + (log ? 0 : (size_t)(threadIdx.x >> 6) * 64);
  p += tsc;
  L.R.itheta = reinterpret_cast<int*>(p);
  L.tile_mask = reinterpret_cast<unsigned int*>(L.R.itheta + nsc * blockDim.x) + (log ? 0 : (size_t)(threadIdx.x >> 6) * 64);
  used = (size_t)((L.R.itheta + nsc * blockDim.x) - reinterpret_cast<int*>(lds_base)) * sizeof(int) + tsc * sizeof(unsigned int);
  }
  {  // rotation()'s cost, sint, sing for the axis (-u_obs, -v_obs, -w_obs): the same expressions, once per observer
    double* rot = lds_base + (used + 7) / 8;
    for (int q = threadIdx.x; q < A.nRT; q += blockDim.x) {
      const double u1 = -A.rt_u[q], v1 = -A.rt_v[q], w1 = -A.rt_w[q % A.RT_n_incl];
      double cost, sint, sing;
      if (w1 > 0.999999999) { cost = 1.0; sint = 0.0; sing = 0.0; }
      else if (fabs(u1) < TINY_REAL) { cost = 0.0; sint = 1.0; sing = sqrt(1.0 - w1 * w1); }
      else { const double h = sqrt(u1 * u1 + v1 * v1); cost = u1 / h; sint = v1 / h; sing = sqrt(1.0 - w1 * w1); }
      rot[3 * q] = cost; rot[3 * q + 1] = sint; rot[3 * q + 2] = sing;
    }
    L.R.rot = rot;
    L.kf = nullptr;
    if (A.kf_lds) {   // kappa_factor behind the rotation constants
      double* kf = rot + (size_t)3 * A.nRT;
      for (int i = threadIdx.x; i < M.n_cells; i += blockDim.x) kf[i] = M.kappa_factor[i];
      L.kf = kf;
    }
  }
  // the Mueller columns of p_lambda
  const size_t col = (size_t)na1 * (A.p_lambda - 1);
  for (int i = threadIdx.x; i < na1; i += blockDim.x) {
    L.mu[i] = A.s11 ? A.s11[i] : 0.0f;
    if (POLA) {
      L.mu[na1 + i] = M.s12[col + i]; L.mu[2 * na1 + i] = M.s22[col + i]; L.mu[3 * na1 + i] = M.s33[col + i];
      L.mu[4 * na1 + i] = M.s34[col + i]; L.mu[5 * na1 + i] = M.s44[col + i];
    }
  }
  __syncthreads();
  return L;
}

// work item -> (stream, sequence number) (see the header comment)
template <bool SCOUT>
__device__ inline void mono_item(const MonoArgs& A, unsigned long long my, unsigned long long& chunk,
                                 unsigned long long& seq) {
  if (SCOUT) {
    const unsigned long long a = my / A.batch;
    chunk = (unsigned long long)A.active[a];
    seq = A.seq0[chunk] + (my - a * A.batch);
  } else {
    int lo = 0, hi = A.n_chunks;  // item_base[lo] <= my < item_base[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (A.item_base[mid] <= my) lo = mid; else hi = mid;
    }
    chunk = (unsigned long long)lo;
    seq = my - A.item_base[lo] + (A.seq0 ? A.seq0[lo] : 0ull);  // (seq0: where this launch's range of the stream starts)
  }
}

// forced scattering (dust_transfer.f90:1263-1278): the packet's weight after the albedo, or dead
template <bool POLA>
__device__ inline bool mono_attenuate(const Lds& T, int lambda, double S[4]) {  // (T: the tables of the cell's class)
  const double alb = (double)T.albedo[lambda - 1];
  S[0] *= alb;
  if (POLA) { S[1] *= alb; S[2] *= alb; S[3] *= alb; }
  return S[0] < (double)(FLT_TINY_X1E6);
}

// kappa(p_icell, lambda) * kappa_factor(icell) (optical_depth.f90:100-102) of a real cell
// (kf_lds: kappa_factor staged in LDS, or null -- see mono_lds_bytes; the two loads in their own address spaces)
template <bool L3D>
__device__ inline double mono_opacity(const Lds& T, const DevModel& M, int lambda, int ri, int zj, int k, const double* kf_lds = nullptr) {
  if (!is_real_cell<L3D>(M.n_rad, M.nz, ri, zj)) return 0.0;
  const int ic = cell_index<L3D>(M.n_rad, M.nz, ri, zj, k);
  const double kap = M.n_classes ? M.v_kappa[(size_t)M.cell_class[ic] * M.n_lambda + (lambda - 1)] : T.kappa[lambda - 1];
#ifndef MCGPU_LANE_EMULATION
  if (kf_lds) return kap * ((const __attribute__((address_space(3))) double*)kf_lds)[ic];
#else
  if (kf_lds) return kap * kf_lds[ic];
#endif
  return kap * M.kappa_factor[ic];
}

#ifndef MCGPU_LANE_EMULATION
// ---- The deposits as a log (round 6) ---------------------------------------------------------------------------------
// The commit pass's xI_scatt atomics -- nRT / 2 line operations per crossing against a chip-wide ceiling of 2.4e10 a
// second -- were 99 % of BASELINE config 2's wall time.  k_mono<..., LOG> makes none: a crossing appends ONE 12-byte record
// (sub-bin | flag_star, flight id, path length) and a flight, once, the row of its nRT x 4 default-real deposit weights
// (what angles_scatt_rt1 computes per flight anyway); mc_xilog.hip.h sorts the records by sub-bin and sums each sub-bin's
// consecutive records in registers -- 2.4x the atomics' rate standalone (tools/xi_fold_bench.hip), and the transport
// kernel loses its per-lane LDS scratch, its staging tiles and its waits.  A wave reserves log space in blocks (one
// global atomic per 2048 records / 256 flights, not per crossing) and fills what it leaves unused with a key that sorts last.
constexpr unsigned int XLOG_REC_BLOCK = 2048;
constexpr unsigned int XLOG_FL_BLOCK = 256;
struct XiLogCursor { unsigned long long rec_next, rec_end, fl_next, fl_end; };   // (wave-uniform)

__device__ inline void xlog_pad(const MonoArgs& A, unsigned long long from, unsigned long long to, int lane) {
  for (unsigned long long i = from + (unsigned long long)lane; i < to; i += 64ull)
    if (i < A.log_cap) A.log_keys[i] = A.log_sentinel;
}

// a flight id for every lane with `want` (call with the whole wave)
__device__ inline unsigned long long xlog_flights(const MonoArgs& A, XiLogCursor& C, bool want, int lane) {
  const unsigned long long m = __ballot(want);
  if (!m) return 0ull;
  const unsigned long long n = (unsigned long long)__popcll(m);
  if (C.fl_next + n > C.fl_end) {
    unsigned long long base = 0ull;
    if (lane == 0) base = atomicAdd(&A.log_ctl[1], (unsigned long long)XLOG_FL_BLOCK);
    base = __shfl(base, 0);
    C.fl_next = base; C.fl_end = base + XLOG_FL_BLOCK;
    if (C.fl_end > A.rows_cap && lane == 0) *A.err = 18;   // (the host sizes a launch for its flights: see xi_log_commit)
  }
  const unsigned long long id = C.fl_next + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
  C.fl_next += n;
  return id;
}

// one record per lane with `on` (call with the whole wave)
__device__ inline void xlog_append(const MonoArgs& A, XiLogCursor& C, bool on, unsigned int key, unsigned int fid, float l, int lane) {
  const unsigned long long m = __ballot(on);
  if (!m) return;
  const unsigned long long n = (unsigned long long)__popcll(m);
  if (C.rec_next + n > C.rec_end) {
    xlog_pad(A, C.rec_next, C.rec_end, lane);
    unsigned long long base = 0ull;
    if (lane == 0) base = atomicAdd(&A.log_ctl[0], (unsigned long long)XLOG_REC_BLOCK);
    base = __shfl(base, 0);
    C.rec_next = base; C.rec_end = base + XLOG_REC_BLOCK;
    if (C.rec_end > A.log_cap && lane == 0) *A.err = 18;
  }
  if (on) {
    const unsigned long long at = C.rec_next + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
    if (at < A.log_cap) {
      A.log_keys[at] = key;
      A.log_vals[at] = (unsigned long long)fid | ((unsigned long long)__float_as_uint(l) << 32);
    }
  }
  C.rec_next += n;
}
#endif

// SCOUT: no deposits, no SED; records hits.  Otherwise the COMMIT pass.
// SPH: the grid operators of spherical_grid.f90 (as thermal_body has them)
// LOG (with F32, one dust class): the deposits go to the log instead of xI_scatt (see "The deposits as a log")
template <bool L3D, bool POLA, bool DARK, bool SCOUT, bool F32 = false, bool SPH = false, bool LOG = false>
__device__ __forceinline__ void mono_body(const DevModel& M, const MonoArgs& A, double* lds_base) {
  const Lds T = lds_carve(lds_base, M, true);
  lds_stage_mono(T, M, A.p_lambda);
  const int na1 = M.nang + 1;
  const MonoLds ML = mono_lds_setup<POLA>(M, A, lds_base, true, LOG || SCOUT);   // (a scout pass makes no deposits: no per-lane results, no tiles)
  const float* mu = ML.mu;
  const RtScratch& R = ML.R;
  double* const tile = ML.tile;
  unsigned long long* const tile_addr = ML.tile_addr;
  unsigned int* const tile_mask = ML.tile_mask;


  const int lane = threadIdx.x & 63;
  const int n_rad = M.n_rad, nz = M.nz;
  const int lambda = A.lambda;
  int st = S_EMIT;
  double x = 0, y = 0, z = 0, u = 0, v = 0, w = 1, extr = 0, inv_a = 0, inv_w = 0;
  double xo = 0, yo = 0, zo = 0;
  int ri = 0, zj = 1, k = 1, ri_o = 0, zj_o = 1, k_o = 1;
  int star_key = -1;
  bool flag_star = false, flag_scatt = false, flag_ism = false;
  double S[4] = {1.0, 0.0, 0.0, 0.0};
  Rng rng;
  rng.init(0, 0);
  unsigned int c_cross = 0, c_flight = 0, c_scatt = 0, c_abs = 0, c_esc = 0, c_kill = 0, c_dark = 0, c_pack = 0;
  unsigned int pk_cross = 0;
  unsigned long long pk_next = 0, pk_end = 0, my_item = 0;
  float tau_rand = 0.0f;
  double kf = 0.0;  // opacity of the packet's cell (mono_opacity)
  const bool var = M.n_classes != 0;
#ifndef MCGPU_LANE_EMULATION
  XiLogCursor LC = {0ull, 0ull, 0ull, 0ull};
  unsigned int my_fid = 0u;
  const size_t row_floats = (size_t)A.nRT * (POLA ? 4 : 1);
#endif

  for (;;) {
    if (st == S_EXITED) {  // capteur (dust_transfer.f90:549-552); forced scattering never clears flag_ISM
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
          // work item -> (stream, sequence number)
          my_item = my + (SCOUT ? 0ull : A.item_lo);
          unsigned long long chunk, seq;
          mono_item<SCOUT>(A, my_item, chunk, seq);
          rng.init(A.seed, ((chunk + (unsigned long long)A.first_chunk) << 40) | seq);
          c_pack++;
          pk_cross = 0;
          float f[12];
          rng.emission_event(f);  // f[0]: the wavelength draw of the thermal step, unused (lmono, :535)
          tau_rand = f[8];
          bool lintersect;
          flag_scatt = false;
          S[0] = 1.0; S[1] = 0.0; S[2] = 0.0; S[3] = 0.0;
          int rc;
          if (SPH) {
            SphEmitOps<L3D> ops{T, M, ri, zj, k};
            rc = emit_packet(M, f, lambda, A.frac_E_stars, A.frac_E_disk, A.prob_E_cell, ops, x, y, z, u, v, w,
                             flag_star, flag_ism, lintersect);
          } else {
            CylEmitOps<L3D> ops{T, M, ri, zj, k};
            rc = emit_packet(M, f, lambda, A.frac_E_stars, A.frac_E_disk, A.prob_E_cell, ops, x, y, z, u, v, w,
                             flag_star, flag_ism, lintersect);
          }
          if (rc) {
            *A.err = rc;
            st = S_DONE;
          }
          if (st != S_DONE) st = lintersect ? S_NEWFLIGHT : S_EXITED;
        }
      }
    }

    if (st == S_INTERACT) {  // forced scattering (dust_transfer.f90:1263-1278)
      float g[8];
      rng.interaction_event(g, M.m1 != 0);
      tau_rand = g[5];
      bool dead = false;
      if (DARK) dead = M.dark[cell_index<L3D>(n_rad, nz, ri, zj, k)] != 0;
      // lvariable_dust: the albedo and the scattering tables of the cell's class (column p_lambda of its cumulative table)
      const int cls = var ? M.cell_class[cell_index<L3D>(n_rad, nz, ri, zj, k)] : -1;
      const Lds Tc = var ? class_tables(T, M, cls) : T;
      if (!dead) dead = mono_attenuate<POLA>(Tc, lambda, S);
      if (dead) {
        c_abs++;
        st = S_EMIT;  // lpacket_alive = .false.: not binned
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

#ifndef MCGPU_LANE_EMULATION
    unsigned long long fid_new = 0ull;
    if (LOG && !SCOUT) fid_new = xlog_flights(A, LC, st == S_NEWFLIGHT && A.rt1, lane);
#endif
#ifndef MCGPU_LANE_EMULATION
    // (the commit pass with default-real records and one dust class: the new flights' observer weights by the whole wave)
    constexpr bool kAnglesByWave = F32 && !SCOUT && !LOG;
    if (kAnglesByWave && A.rt1 && !var) angles_scatt_rt1_wave<POLA>(M, A, R, st == S_NEWFLIGHT, u, v, w, mu, S, tile, ML.row, flag_star);
#else
    constexpr bool kAnglesByWave = false;
#endif
    if (st == S_NEWFLIGHT) {
      const float rand = tau_rand;
      extr = tau_of_draw(rand);
      const double a = u * u + v * v;
      inv_a = (a > TINY_REAL) ? 1.0 / a : HUGE_REAL;
      inv_w = (fabs(w) > TINY_REAL) ? 1.0 / w : copysign(HUGE_DP, w);
#ifndef MCGPU_LANE_EMULATION
      if (LOG && !SCOUT && A.rt1) {   // the flight's deposit weights -> its row of the log
        my_fid = (unsigned int)fid_new;
        angles_scatt_rt1<POLA>(M, A, R, u, v, w, mu, S, A.log_rows + (fid_new < A.rows_cap ? fid_new : 0ull) * row_floats);
      } else
#endif
      if (!SCOUT && A.rt1 && !(kAnglesByWave && !var))
        angles_scatt_rt1<POLA>(M, A, R, u, v, w, (F32 && !var) ? mu : nullptr, S, nullptr, ML.row, flag_star);  // optical_depth.f90:65
      const int i_star = intersect_stars(M, x, y, z, u, v, w);
      star_key = -1;
      if (i_star > 0) {
        const int* sc = &M.star_cell[4 * (i_star - 1)];
        star_key = sc[0] + (n_rad + 2) * ((sc[1] + nz + 1) + (2 * nz + 3) * (sc[2] - 1));
      }
      c_flight++;
      ri_o = 0; zj_o = 0; k_o = 0;
      xo = x; yo = y; zo = z;
      kf = mono_opacity<L3D>(T, M, lambda, ri, zj, k, ML.kf);
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
        const int azj = zj < 0 ? -zj : zj;
        const bool out = (ri == n_rad + 1) || (!SPH && (azj == nz + 1) && (fabs(z) > M.zmaxmax));   // (test_exit_grid_sph: the outer radius only)
        bool killed = false;
        if (star_key >= 0) {
          const int key = ri + (n_rad + 2) * ((zj + nz + 1) + (2 * nz + 3) * (k - 1));
          killed = (key == star_key);
        }
        if (out) {
          st = S_EXITED;
        } else if (killed) {
          c_kill++;
          st = S_EMIT;
        } else {
          const bool real_cell = is_real_cell<L3D>(n_rad, nz, ri, zj);
          double opacity = 0.0;
          int ic = 0;
          bool mirrored = false;
          if (real_cell) {
            ic = cell_index<L3D>(n_rad, nz, ri, zj, k);
            opacity = kf;
            if (DARK) {
              if (M.dark[ic]) {  // optical_depth.f90:104-112
                u = -u; v = -v; w = -w;
                x = xo; y = yo; z = zo;
                ri = ri_o; zj = zj_o; k = k_o;
                c_dark++;
                mirrored = true;
                st = S_INTERACT;
              }
            }
          }
          if (!mirrored) {
            double x1, y1, z1, l;
            int ri1, zj1, k1;
            if (SPH) cross_cell_sph<L3D>(T, M, x, y, z, u, v, w, ri, zj, k, x1, y1, z1, ri1, zj1, k1, l);
            else MCGPU_CROSS<L3D>(T, M, x, y, z, u, v, w, inv_a, inv_w, ri, zj, k, x1, y1, z1, ri1, zj1, k1, l);
            c_cross++;
            const double tau = l * opacity;
            if (tau > extr) {
              const double lc = l * (extr / tau);
              if (!SCOUT && A.rt1 && real_cell) {
                dep.on = true; dep.icell = ic + 1; dep.l = lc;
                rt1_subbin<L3D>(A, x, y, z, x1, y1, z1, dep.phik, dep.psup);
              }
              if (!L3D && !SCOUT && A.rt2 && real_cell) {  // (the midpoint is the crossing's, optical_depth.f90:149-150)
                dep.on = true; dep.icell = ic + 1; dep.l = lc;
                rt2_bins(A, x, y, z, x1, y1, z1, u, v, w, dep.psup, dep.phik);  // (psup, phik: theta_I, phi_I)
              }
              x = x + lc * u;
              y = y + lc * v;
              z = z + lc * w;
              if (L3D && !SPH) index_cell<L3D>(T, M, x, y, z, ri, zj, k, ri);   // (optical_depth.f90:162-165: lcylindrical only)
              st = S_INTERACT;
            } else {
              extr = extr - tau;
              if (!SCOUT && A.rt1 && real_cell) {
                dep.on = true; dep.icell = ic + 1; dep.l = l;
                rt1_subbin<L3D>(A, x, y, z, x1, y1, z1, dep.phik, dep.psup);
              }
              if (!L3D && !SCOUT && A.rt2 && real_cell) {
                dep.on = true; dep.icell = ic + 1; dep.l = l;
                rt2_bins(A, x, y, z, x1, y1, z1, u, v, w, dep.psup, dep.phik);
              }
              if (DARK) { xo = x; yo = y; zo = z; ri_o = ri; zj_o = zj; k_o = k; }
              x = x1; y = y1; z = z1;
              ri = ri1; zj = zj1; k = k1;
              kf = mono_opacity<L3D>(T, M, lambda, ri, zj, k, ML.kf);
            }
            if (++pk_cross > 200000000u) {  // a packet that never leaves: flag it, drop it
              *A.err = 13;
              st = S_EMIT;
            }
          }
        }
      }
      // the deposits of this crossing, by the whole wavefront (S and flag_star are still the flight's:
      // an interaction changes them only in the next outer phase)
      if (!L3D && !SCOUT && A.rt2 && __ballot(dep.on) != 0ull)
        deposit_rt2_wave<POLA>(A, dep.on, dep.icell, dep.psup, dep.phik, dep.l, S, flag_star, flag_star && !flag_scatt, tile,
                               tile_addr, tile_mask);
      if (!SCOUT && A.rt1 && __ballot(dep.on) != 0ull)
      {
#ifndef MCGPU_LANE_EMULATION
        if constexpr (LOG) {
          const unsigned int bin = (unsigned int)((((size_t)(dep.icell - 1) * A.n_theta_rt + (dep.psup - 1)) * A.n_az_rt) + (dep.phik - 1));
          xlog_append(A, LC, dep.on, bin | (flag_star ? 0x80000000u : 0u), my_fid, (float)dep.l, lane);
        } else if constexpr (F32) {
          if (ML.row) deposit_rt1_wave_row(A, ML.row, dep, flag_star, tile, tile_addr);
          else deposit_rt1_wave_f32<POLA>(M, A, R, mu, dep, S, flag_star, tile, tile_addr, tile_mask);
        } else
#endif
        deposit_rt1_wave<POLA>(M, A, R, mu, dep, S, flag_star, tile, tile_addr, tile_mask);
      }
    }
  }
#ifndef MCGPU_LANE_EMULATION
  if (LOG && !SCOUT) xlog_pad(A, LC.rec_next, LC.rec_end, lane);   // (what the wave reserved and did not use sorts behind the records)
#endif

  if (!SCOUT) {
    unsigned int cs[8] = {c_pack, c_cross, c_flight, c_scatt, c_abs, c_esc, c_kill, c_dark};
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

template <bool L3D, bool POLA, bool DARK, bool SCOUT, bool F32 = false, bool LOG = false>
__global__ void __launch_bounds__(512) k_mono(const DevModel M, const MonoArgs A) {
  extern __shared__ double lds_raw[];
  mono_body<L3D, POLA, DARK, SCOUT, F32, false, LOG>(M, A, lds_raw);
}

// ... on a spherical grid (no dark zone there)
template <bool L3D, bool POLA, bool SCOUT, bool F32 = false>
__global__ void __launch_bounds__(512) k_mono_sph(const DevModel M, const MonoArgs A) {
  extern __shared__ double lds_raw[];
  mono_body<L3D, POLA, false, SCOUT, F32, true>(M, A, lds_raw);
}

// Stopping index of every active stream from this batch's hit flags: one wave per stream.
//   need[c]  packets still to be binned in capt_sup (updated)
//   sent[c]  packets sent so far = seq0 (updated: += batch, or the exact stop)
//   lim      n_phot_lim as an integer cap
// done[c] = 1 when the stream stopped inside this batch (or hit the cap).
static __global__ void k_mono_scan(const int* active, int n_active, unsigned long long batch, const unsigned char* hits,
                            unsigned long long* need, unsigned long long* sent, unsigned long long lim,
                            int* done) {
  const int a = blockIdx.x;
  if (a >= n_active) return;
  const int c = active[a];
  const int lane = threadIdx.x;
  const unsigned char* h = hits + (size_t)a * batch;
  unsigned long long want = need[c], base = sent[c];
  unsigned long long stop = batch;  // packets of this batch that count
  bool fin = false;
  // the cap: the stream may send at most lim packets in total
  unsigned long long usable = batch;
  if (base + batch >= lim) { usable = lim > base ? lim - base : 0; }
  for (unsigned long long s0 = 0; s0 < usable && !fin; s0 += 64) {
    const unsigned long long s = s0 + lane;
    const bool hit = (s < usable) && h[s];
    const unsigned long long m = __ballot(hit);
    const unsigned long long n = (unsigned long long)__popcll(m);
    if (n >= want) {  // the want-th hit of this group ends the stream
      unsigned long long mm = m;
      for (unsigned long long q = 1; q < want; ++q) mm &= mm - 1;  // drop the first want-1 hits
      const int pos = __ffsll((long long)mm) - 1;
      stop = s0 + (unsigned long long)pos + 1;
      want = 0;
      fin = true;
    } else {
      want -= n;
    }
  }
  if (!fin && usable < batch) { stop = usable; fin = true; }  // n_phot_lim reached
  if (lane == 0) {
    need[c] = want;
    sent[c] = base + (fin ? stop : batch);
    done[c] = fin ? 1 : 0;
  }
}

// device layout [icell][psup][phik][iRT][8] -> the reference's xI_scatt(phik,psup,type,iRT,icell),
// in FP64 and/or default real (what the reference's array holds).  One thread per output element.
static __global__ void k_xI_fetch(const double* xI, float* out32, double* out64, int n_az, int n_theta, int n_type, int nRT,
                           size_t n, int f32, Xi32Lay xi, int nS) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  size_t r = i;
  const int phik = (int)(r % n_az); r /= n_az;
  const int psup = (int)(r % n_theta); r /= n_theta;
  const int type = (int)(r % n_type); r /= n_type;
  const int q = (int)(r % nRT); r /= nRT;  // r = icell - 1
  const size_t bin = (r * n_theta + psup) * n_az + phik;
  double v;
  if (f32) {   // (the packed default-real layout, xi32_*: a type no deposit reaches reads as 0)
    v = xi32_value(reinterpret_cast<const float*>(xI) + bin * xi.binf, xi, q, type, nS);
  } else {
    v = xI[(bin * nRT + q) * XI_LINE + type];
  }
  if (out64) out64[i] = v;
  if (out32) out32[i] = (float)v;
}

// the reference's xI_scatt(phik,psup,type,iRT,icell) -> device layout (mcgpu_set_xI).  One thread per element.  (Default
// real: the types the Monte Carlo never deposits -- direct light, n_Stokes + 1 and + 3 -- have no place and are dropped, and
// with lsepar_contrib neither has I: it is read back as the sum of the two origins, which is what the Monte Carlo makes it.)
static __global__ void k_xI_put(double* xI, const double* in64, int n_az, int n_theta, int n_type, int nRT, size_t n, int f32,
                         Xi32Lay xi, int nS) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  size_t r = i;
  const int phik = (int)(r % n_az); r /= n_az;
  const int psup = (int)(r % n_theta); r /= n_theta;
  const int type = (int)(r % n_type); r /= n_type;
  const int q = (int)(r % nRT); r /= nRT;
  const size_t bin = (r * n_theta + psup) * n_az + phik;
  if (f32) {
    const int o = xi32_offset(xi, q, type, nS);
    if (o >= 0) reinterpret_cast<float*>(xI)[bin * xi.binf + o] = (float)in64[i];
  } else {
    xI[(bin * nRT + q) * XI_LINE + type] = in64[i];
  }
}

}  // namespace mcgpu
