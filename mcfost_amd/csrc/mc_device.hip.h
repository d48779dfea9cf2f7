// mc_device.hip.h -- device side of the MI355X Monte Carlo packet engine.
//
// One photon packet per lane, persistent wavefronts.  This is a from-scratch
// CDNA4 design of the path the reference runs as an OpenMP loop
// (dust_transfer.f90:439-572); it is NOT a translation of the Fortran:
//   * the packet is a register-resident state machine (EMIT -> NEWFLIGHT ->
//     FLIGHT -> INTERACT -> ...); a lane whose packet leaves the grid pulls the
//     next packet id from a wave-aggregated global counter;
//   * the cell is carried as (ri, zj, phik) integers; the reference's
//     cell_map / cell_map_i/j/k gathers are replaced by closed forms;
//   * z_lim is evaluated from one per-radius cell height held in LDS;
//   * all wavelength tables (opacities, emission CDFs, Lucy/B&W tables) are
//     staged in LDS once per workgroup; kappa_factor and the absorbed-energy
//     grid stay in HBM/L2, deposits are native FP64 atomics;
//   * random numbers are counter-based (Philox4x32-10) in registers, keyed by
//     (seed, packet id): results do not depend on which lane runs a packet.
// The arithmetic of every operator follows the reference expression by
// expression (file:line cited at each routine) so that it can be checked
// against the CPU oracle.
#pragma once
#ifndef MCGPU_LANE_EMULATION  // tests/emu compiles this header for one emulated lane on the CPU
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#ifndef MCGPU_CROSS
#define MCGPU_CROSS cross_cell_lean  // (tests/emu can substitute the branch-for-branch form of tests/emu/cross_cell_literal.h)
#endif
#ifndef MCGPU_LDS_BLOCK
#define MCGPU_LDS_BLOCK 512  // threads of an LDS-deposit workgroup (one per CU on the 2D BASELINE grid)
#endif

// Diagnostic switches of the kernels (skip deposits, record wave timelines: tools/*.py) exist in
// -DMCGPU_TUNING builds only; the shipped library compiles them out.
#ifdef MCGPU_TUNING
#define MCGPU_DIAG(flags, bit) (((flags) & (bit)) != 0)
#else
#define MCGPU_DIAG(flags, bit) false
#endif

namespace mcgpu {

constexpr double PI = 3.141592653589793238462643383279502884197;
constexpr double GRID_PREC = 1.0e-14;              // cylindrical_grid.f90:16
constexpr double TINY_REAL = 1.17549435082228750797e-38;  // tiny(0.0)
constexpr double HUGE_REAL = 3.40282346638528859812e+38;  // huge(1.0)
constexpr double HUGE_DP = 1.79769313486231570815e+308;
constexpr double TINY_DP = 2.22507385850720138309e-308;
constexpr float FLT_TINY = 1.17549435082228750797e-38f;
constexpr float FLT_HUGE = 3.40282346638528859812e+38f;
constexpr float FLT_TINY_X1E6 = 1.17549435082228750797e-38f * 1.0e6f;  // tiny_real_x1e6 (constants.f90:156)

// Products and sums that must NOT be contracted into FMA (hipcc's default -ffp-contract=
// fast-honor-pragmas fuses a*b+c everywhere else, and HIP's __fmul_rn/nd_add are plain
// operators that get fused too): used where the reference's unfused default-real / FP64
// arithmetic decides a discrete outcome, so that the CPU oracle reproduces it bit for bit.
#ifndef MCGPU_LANE_EMULATION
__device__ __forceinline__ float nf_mul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float nf_add(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float nf_sub(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ double nd_mul(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ double nd_add(double a, double b) {
#pragma clang fp contract(off)
  return a + b;
}
// sqrt(x) for x = 0 or x far above the subnormals (x >= 2^-767): the instruction sequence hipcc expands sqrt(double)
// into -- v_rsq_f64 and two coupled Newton steps -- without the rescaling of tiny arguments and the infinity test
// that sequence carries for the general case (7 of its 24 vector instructions).  The same bits as sqrt() on that
// domain; the crossing's discriminants (AU^2) are never anywhere near 1e-231.
__device__ __forceinline__ double sqrt_nonneg(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return (x == 0.0) ? 0.0 : g;
}
#endif
// ... without the last correction step: within a few units in the last place (the flight-parametric crossing, which
// makes no claim on the reference's last bits)
__device__ __forceinline__ double sqrt_fast_nonneg(double x) {
#ifdef MCGPU_LANE_EMULATION
  return sqrt(x);
#else
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return (x == 0.0) ? 0.0 : g;
#endif
}

// log(x) for positive, finite, normal x: frexp, s = (m - 1) / (m + 1) with m in [sqrt(1/2), sqrt(2)), the series
// 2 s (1 + s^2 / 3 + ... + s^18 / 19) and e ln 2 -- 4e-16 of libm's log (2 ulp; checked over 5e6 arguments on the host),
// 35 vector instructions against the 98 of the library routine, which carries a double-double tail for its last bit.
// The packet loop takes two logarithms per interaction: the optical depth of the next flight, -log(1 - rand)
// (dust_transfer.f90:1208-1215 -- in default real there) and Temp_LTE's log(Qheat) (thermal_emission.f90:684).
__device__ __forceinline__ double log_pos(double x) {
#ifdef MCGPU_LANE_EMULATION
  return log(x);
#else
  double m = __builtin_amdgcn_frexp_mant(x);   // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool lo = m < 0.70710678118654752;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  // 1 / (m + 1), m + 1 in [1.7, 2.42): the hardware's estimate and two Newton steps
  const double d = m + 1.0;
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
  r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
  const double s = (m - 1.0) * r, z = s * s;
  double p = 1.0 / 19.0;
  p = __builtin_fma(p, z, 1.0 / 17.0);
  p = __builtin_fma(p, z, 1.0 / 15.0);
  p = __builtin_fma(p, z, 1.0 / 13.0);
  p = __builtin_fma(p, z, 1.0 / 11.0);
  p = __builtin_fma(p, z, 1.0 / 9.0);
  p = __builtin_fma(p, z, 1.0 / 7.0);
  p = __builtin_fma(p, z, 1.0 / 5.0);
  p = __builtin_fma(p, z, 1.0 / 3.0);
  p = __builtin_fma(p, z, 1.0);
  return __builtin_fma((double)e, 0.6931471805599453, (s + s) * p);
#endif
}
// the optical depth a flight is given (dust_transfer.f90:1208-1215)
__device__ __forceinline__ double tau_of_draw(float rand) {
  return (rand > 1.0e-6f) ? -log_pos(1.0 - (double)rand) : (double)rand;
}

struct DevModel {
  // grid
  int n_rad, nz, n_az, l3D, n_cells;
  const double* r_lim_2;   // [n_rad+1]
  const double* zmax;      // [n_rad]
  const double* ch;        // [n_rad] cell height = z_lim(i,2)
  const double* tan_phi_lim;  // [n_az]
  double zmaxmax, Rmax2;
  // spherical grid (spherical_grid.f90; grid_sph != 0): the polar walls and r^3 (cylindrical_grid.f90:28-31)
  int grid_sph;
  const double* tan_theta_lim;  // [nz+1]
  const double* theta_lim;      // [nz+1]
  const double* r_lim_3;        // [n_rad+1]
  const double* volume;    // [n_cells]
  // stars: x,y,z,r and (ri,zj,k,out_model)
  int n_stars;
  const double* star_xyzr;  // [4*n_stars]
  const int* star_cell;     // [4*n_stars]
  // opacity
  int n_lambda;
  const double* kappa;
  const double* kappa_abs;
  const float* albedo;
  const double* kappa_factor;  // [n_cells]
  const unsigned char* dark;   // [n_cells] or null
  // scattering
  int nang, aniso_method, lisotropic, p_lambda_fixed;
  const float* prob_s11;  // (0:nang, n_lambda)
  const float* s12;
  const float* s22;
  const float* s33;
  const float* s34;
  const float* s44;
  const float* tab_g;
  const double* cos_tab;  // [nang+1] cos(k*pi/nang), tabulated by the host (scattering.f90:1470-1471)
  // thermal
  int n_T;
  const double* log_Qcool;  // [n_T]
  const double* cdf;        // (n_lambda, n_T)
  const double* spec_cum;   // [n_lambda+1]
  const double* frac_E_stars;
  const double* frac_E_disk;
  const double* CDF_E_star;   // (n_lambda, 0:n_stars)
  const double* prob_E_cell;  // (0:n_cells, n_lambda) or null
  double L_packet_th;
  // sed
  int N_thet, N_phi, sym_c, sym_a;
  // 3D midplane crossings land at sign(grid_prec, w) (see include/mcgpu.h)
  int midplane_snap;
  // interstellar radiation field: emitting sphere (stars.f90:27-28); R_ISM = 0: no ISM emission
  double R_ISM, centre_ISM[3];
  // modified random walk (mcgpu_set_mrw; MRW.f90, dust_transfer.f90:1222-1239): mrw = 0: off
  int mrw, mrw_n_zeta, mrw_n_inter;
  float mrw_gamma;
  const double* mrw_zeta;  // [mrw_n_zeta] zeta(y_i), y_i = i/(n-1)
  const double* mrw_chi;   // [n_T] mean transport extinction at tab_Temp, reference cell
  const double* mrw_kdep;  // [n_T] mean absorption opacity of the walk's deposits
  const double *sin_phi, *cos_phi;  // [n_az] sin / cos of the azimuthal walls (3D; cylindrical_grid.f90:586-599), for the walk
  const double* mrw_ext;   // [n_T] extrapolation length of the sphere radius, reference cell
  const int* mrw_guide;    // [MRW_GUIDE + 1] mrw_guide[b] = the last index i with zeta[i] <= b / MRW_GUIDE: where the search of
                           // mrw_sample_y starts for a draw in bucket b (14 dependent loads -> ~4)
  const double* mrw_exit_cdf;  // [n_T][n_lambda] (per class) cumulative spectrum a walk's last step leaves its sphere with;
                               // nullptr: the emission spectrum kdB_dT_CDF (mcgpu_set_mrw_exit_spectrum)
  const double* r_lim;     // [n_rad+1] (distance_to_closest_wall_cyl)
  // lvariable_dust (mcgpu_set_variable_dust; mem.f90:213-244, p_n_cells > 1): per-class tables in HBM, class-major
  int n_classes;             // 0: one class, the tables above (and their LDS copies)
  const int* cell_class;     // [n_cells] p_icell - 1
  const double* v_kappa;     // [n_classes][n_lambda]
  const double* v_kabs;      // [n_classes][n_lambda]
  const float* v_albedo;     // [n_classes][n_lambda]
  const double* v_lq;        // [n_classes][n_T]
  const double* v_cdf;       // [n_classes][n_T][n_lambda]
  const double2* v_kk;       // [n_cells + 1][n_lambda] (kappa(p_icell, l) kappa_factor(icell), kappa_abs_LTE(p_icell, l)); the last row 0
  // ... and, when the host passed them (v_scatt != 0), the scattering tables per class
  int v_scatt;
  const float* v_prob;       // [n_classes][p_lambda_fixed ? 1 : n_lambda][nang+1]
  const float* v_g;          // [n_classes][n_lambda]
  const float* v_s11;        // [n_classes][n_lambda][nang+1]: tab_s11_pos (the ray tracer's phase function; mcgpu_opacity builds it)
  const float* v_s12;        // [n_classes][n_lambda][nang+1], likewise v_s22 ... v_s44
  const float* v_s22;
  const float* v_s33;
  const float* v_s34;
  const float* v_s44;
  // scattering method 1 (lscattering_method1, dust_transfer.f90:1288-1316; mcgpu_set_scattering_method1): the grain that
  // scatters is drawn from the cell's population, then its own phase function and Mueller matrix are used
  int m1, m1_ng;
  const float* m1_Csca;      // C_sca(n_grains, n_lambda)
  const double* m1_nk;       // n_grains(k)
  const double* m1_dens;     // dust_density_o_n_grains(n_grains, class)
  const double* m1_ksca;     // nullptr (low_mem_scattering: the walk below), or ksca_CDF [class][lambda][0:n_grains]
                             // (dust_prop.f90:24, built by k_ksca_cdf / mcgpu_build_ksca_CDF): select_grainsize_high_mem
  const float* m1_prob;      // [n_lambda][n_grains][nang+1] cumulative phase function of the grain
  const float* m1_g;         // tab_g(n_grains, n_lambda)
  const float *m1_s11, *m1_s12, *m1_s22, *m1_s33, *m1_s34, *m1_s44;  // tab_s1x(0:nang, n_grains, n_lambda)
};

// packet state word of the queue records: state | flags
constexpr int ST_MASK = 15, ST_STAR = 16, ST_SCATT = 32, ST_ISM = 64;
constexpr int ST_NINT_SHIFT = 8;  // bits 8..10: MRW's count of interactions in a row inside one cell (0..7)

// HBM side of the binned-deposit log (mc_binned.hip.h); n_buckets = 0: not in use
struct BinLog {
  unsigned int* keys;          // [blocks][64] cell indices (0-based)
  double* vals;                // [blocks][64]
  unsigned int* count;         // [n_buckets][n_parts] blocks written by workgroup `part` since the last fold
  const unsigned int* off;     // [n_buckets] first block of the bucket's region
  const unsigned int* cap;     // [n_buckets] blocks of ONE workgroup's part of the region (n_parts of them in a row)
  unsigned long long* stats;   // [0] blocks that overflowed their part, [1] records added at the end of a launch
  int n_buckets, shift;        // bucket = cell >> shift
  int n_parts;                 // workgroups of the packet kernel
};

struct RunArgs {
  uint64_t seed, first_packet, n_packets;
  double qscale;            // n_replicas
  int frozen;
  const double* E_prior;    // [n_cells] (frozen mode)
  double* E_abs;            // [n_cells]
  double* sed;              // [9 * n_lambda*N_thet*N_phi]
  double* n_sent;           // [n_lambda]
  unsigned long long* counters;  // [MCGPU_N_COUNTERS]
  unsigned long long* next_packet;  // work counter
  int* err;
  int inner_iters;  // crossings attempted between two interaction phases
  int flush_every;  // LDS-deposit kernels: outer iterations per fold of the private grid
  int min_active;   // leave the crossing loop early once fewer lanes than this are in flight
  int flags;        // diagnostics, -DMCGPU_TUNING builds only (MCGPU_DIAG)
  // optional radiation-field accumulators of save_radiation_field (radiation_field.f90:54-55): null = off
  unsigned long long* xN_abs;  // [n_cells] path segments per cell (xN_abs(icell,1,id), lmcfost_lib)
  double* xJ_abs;        // (n_cells, n_lambda) sum of l * Stokes(1) (lxJ_abs_step1)
  // binned deposits (mc_binned.hip.h): the launch is one CHUNK of a run whose deposits reach E_abs when the chunk's
  // log is folded; n_folded = packets whose deposits E_abs holds when this chunk starts (see bin_energy_scale)
  BinLog bin;
  double n_folded;
  // chunked runs of the role kernel (mc_roles.hip.h, "Chunks without tails"): packets a chunk leaves unfinished are
  // written to carry_out as records and are the first work items of the next chunk (carry_in); null = run to the end
  const void* carry_in;             // [*carry_in_n] records (Rec<POLA>)
  const unsigned int* carry_in_n;
  void* carry_out;                  // [carry_cap] records; null: this chunk finishes every packet
  unsigned int* carry_out_n;
  unsigned int carry_cap;
  int tail_threshold;               // <= 0: hand over as soon as the work counter has run out (a chunk of a binned run);
                                    // > 0: once the workgroup has no more than that many packets left (-> k_tail)
  // k_tail's own hand-over (mc_tail.hip.h "The last packets on the host", host_tail.cpp; option "tail_where" = 2): once no
  // more than tail_host_max of its packets are unfinished, every wave writes the packet it runs (and the ones not yet
  // started) to tail_out as records and leaves; 0: k_tail finishes every packet
  unsigned int tail_host_max;
  unsigned int* tail_done;          // packets k_tail has finished
  void* tail_out;                   // [tail_host_max] records (Rec<POLA>)
  unsigned int* tail_out_n;
};

// ---------------------------------------------------------------------------
// Philox4x32-10
// ---------------------------------------------------------------------------
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Per-packet random stream, drawn per EVENT so that a wavefront computes the Philox blocks in
// lock-step (no per-draw refill branches) and the only state is the event counter:
//   event 0 (emission + first flight): blocks 0,1,2 -> f[0..11]
//       f0 wavelength, f1 star/disk choice, star: f2 select_star, f3..f6 uniform sphere,
//       disk: f2 select_cellule, f3..f5 pos_em_cell, f6,f7 isotropic direction, f8 first tau
//   event e >= 1 (e-th interaction + next flight): ONE block, 3+(e-1) -> five 24-bit uniforms (the upper 24 bits of the
//       four words, and the three lower bytes of words 0, 1, 2): g0 scatter/absorb, g1 rand, g2 rand2, scatter: g3 azimuth,
//       absorb: g3, g4 (= g1, which a re-emission does not use otherwise) isotropic direction, g5 tau of the next flight.
//       Round 4: two blocks per event before -- 61 vector instructions each; scattering method 1 (six draws per
//       scattering) keeps the two blocks 3+2(e-1), 4+2(e-1) -> g[0..7].
struct Rng {
  uint32_t k0, k1;      // seed (wave-uniform)
  uint32_t p_lo, p_hi;  // packet id
  uint32_t event;       // next event index
  __device__ inline void init(uint64_t seed, uint64_t packet) {
    k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32);
    p_lo = (uint32_t)packet; p_hi = (uint32_t)(packet >> 32);
    event = 0;
  }
  // uniform default-real in [0,1), 24 bits
  static __device__ inline float real(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }
  __device__ inline void block(uint32_t b, float* f) const {
    uint32_t o[4];
    philox4x32_10(b, 0u, p_lo, p_hi, k0, k1, o);
    f[0] = real(o[0]); f[1] = real(o[1]); f[2] = real(o[2]); f[3] = real(o[3]);
  }
  __device__ inline void emission_event(float f[12]) {
    block(0u, f); block(1u, f + 4); block(2u, f + 8);
    event = 1;
  }
  __device__ inline void interaction_event(float g[8], bool two_blocks = false) {
    if (two_blocks) {   // (wave-uniform: scattering method 1)
      const uint32_t first = 3u + 2u * (event - 1u);
      block(first, g); block(first + 1u, g + 4);
    } else {
      uint32_t o[4];
      philox4x32_10(3u + (event - 1u), 0u, p_lo, p_hi, k0, k1, o);
      g[0] = real(o[0]); g[1] = real(o[1]); g[2] = real(o[2]); g[3] = real(o[3]);
      g[4] = g[1];
      g[5] = (float)(((o[0] & 0xFFu) << 16) | ((o[1] & 0xFFu) << 8) | (o[2] & 0xFFu)) * (1.0f / 16777216.0f);
      g[6] = g[7] = 0.0f;
    }
    event += 1;
  }
};

// ---------------------------------------------------------------------------
// LDS-resident tables
// ---------------------------------------------------------------------------
// What the 2D crossing (fly_step_2d, mc_roles.hip.h) reads of a radial index ri = 0 .. n_rad + 2, in ONE 40-byte row so
// that a crossing forms one LDS address per cell instead of clamping row indices into four tables: the wall radii
// already multiplied by the reference's correction factors (r_lim_2(ri-1) * correct_moins, r_lim_2(ri) * correct_plus:
// cylindrical_grid.f90:978, 982 -- the products are the reference's own, rounded once, whoever forms them), the
// cell height and surface of the column, and nz / zmax for the zj of an end point.  ri = 0 is the central hole
// (rl_in = r_lim_2(0) unscaled, :963; rzn = 0: zj = 1 there, :1117), ri > n_rad repeats the last column (read by lanes
// whose results are discarded).
struct RowT {
  double rl_in, rl_out, ch, zmax, rzn;
};

struct Lds {
  RowT* row;        // n_rad+3 (thermal kernels only)
  double* r_lim_2;  // n_rad+1
  double* zmax;     // n_rad
  double* ch;       // n_rad
  double* rzn;      // n_rad: nz / zmax, for the fast path of the zj recomputation
  double* tan_phi;  // n_az
  double* kappa;    // n_lambda
  double* kabs;     // n_lambda
  double* lq;       // n_T
  double* cum;      // n_lambda+1
  double* fstar;    // n_lambda
  double* cdf;      // n_lambda*n_T
  double* cost;     // nang+1
  float* albedo;    // n_lambda
  float* prob;      // (nang+1) * (p_lambda_fixed ? 1 : n_lambda)
  float* g;         // n_lambda
  unsigned int* nsent;  // n_lambda: packets this workgroup emitted per wavelength (thermal kernels; see lds_count_sent)
};

// mono = the SED-mode kernels (mc_mono*.hip.h): no re-emission, one phase-function column -- they leave out the
// thermal tables (log_Qcool, the emission spectrum, frac_E_stars, the re-emission CDF: 41 KB on ref4.1) and stage
// only column p_lambda of prob_s11, which leaves the LDS to the per-lane ray-tracing scratch and to more waves.
__host__ __device__ inline size_t lds_doubles(const DevModel& M, bool mono = false) {
  const size_t geo = (size_t)(M.n_rad + 1) + M.n_rad + M.n_rad + M.n_rad + M.n_az + M.n_lambda + M.n_lambda + (M.nang + 1);
  return mono ? geo : geo + M.n_T + (M.n_lambda + 1) + M.n_lambda + (size_t)M.n_lambda * M.n_T + (M.n_rad > 0 ? 5 * (size_t)(M.n_rad + 3) : 0);
}
__host__ __device__ inline size_t lds_floats(const DevModel& M, bool mono = false) {
  return (size_t)M.n_lambda + (size_t)(M.nang + 1) * ((mono || M.p_lambda_fixed) ? 1 : M.n_lambda) + M.n_lambda +
         (mono ? 0 : (size_t)M.n_lambda);  // (the last term: the n_sent histogram, 32-bit words)
}
__host__ __device__ inline size_t lds_bytes(const DevModel& M, bool mono = false) {
  return lds_doubles(M, mono) * sizeof(double) + lds_floats(M, mono) * sizeof(float);
}

__device__ inline Lds lds_carve(double* base, const DevModel& M, bool mono = false) {
  Lds T;
  double* p = base;
  T.r_lim_2 = p; p += M.n_rad + 1;
  T.zmax = p; p += M.n_rad;
  T.ch = p; p += M.n_rad;
  T.rzn = p; p += M.n_rad;
  T.tan_phi = p; p += M.n_az;
  T.kappa = p; p += M.n_lambda;
  T.kabs = p; p += M.n_lambda;
  if (mono) {
    T.lq = p; T.cum = p; T.fstar = p; T.cdf = p;  // (not staged, never read by those kernels)
    T.row = nullptr;
  } else {
    T.row = reinterpret_cast<RowT*>(p); p += (M.n_rad > 0 ? 5 * (size_t)(M.n_rad + 3) : 0);
    T.lq = p; p += M.n_T;
    T.cum = p; p += M.n_lambda + 1;
    T.fstar = p; p += M.n_lambda;
    T.cdf = p; p += (size_t)M.n_lambda * M.n_T;
  }
  T.cost = p; p += M.nang + 1;
  float* f = reinterpret_cast<float*>(p);
  T.albedo = f; f += M.n_lambda;
  T.prob = f; f += (size_t)(M.nang + 1) * ((mono || M.p_lambda_fixed) ? 1 : M.n_lambda);
  T.g = f; f += M.n_lambda;
  T.nsent = reinterpret_cast<unsigned int*>(f);
  return T;
}

template <typename Tp>
__device__ inline void stage(Tp* dst, const Tp* src, size_t n) {
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

__device__ inline void lds_stage(const Lds& T, const DevModel& M) {
  stage(T.r_lim_2, M.r_lim_2, (size_t)M.n_rad + 1);
  stage(T.zmax, M.zmax, (size_t)M.n_rad);
  stage(T.ch, M.ch, (size_t)M.n_rad);
  for (int i = threadIdx.x; i < M.n_rad; i += blockDim.x) T.rzn[i] = (double)M.nz / M.zmax[i];
  for (int i = threadIdx.x; i < (M.n_rad > 0 ? M.n_rad + 3 : 0); i += blockDim.x) {   // (Voronoi grids: n_rad = 0, no such tables)
    const int rin = i == 0 ? 0 : (i - 1 < M.n_rad ? i - 1 : M.n_rad - 1);
    const int rout = i == 0 ? 0 : (i < M.n_rad ? i : M.n_rad);
    RowT r;
    r.rl_in = i == 0 ? M.r_lim_2[0] : nd_mul(M.r_lim_2[rin], 1.0 - GRID_PREC);
    r.rl_out = nd_mul(M.r_lim_2[rout], 1.0 + GRID_PREC);
    r.ch = M.ch[rin];
    r.zmax = M.zmax[rin];
    r.rzn = i == 0 ? 0.0 : (double)M.nz / M.zmax[rin];
    T.row[i] = r;
  }
  stage(T.tan_phi, M.tan_phi_lim, (size_t)M.n_az);
  stage(T.kappa, M.kappa, (size_t)M.n_lambda);
  stage(T.kabs, M.kappa_abs, (size_t)M.n_lambda);
  stage(T.lq, M.log_Qcool, (size_t)M.n_T);
  stage(T.cum, M.spec_cum, (size_t)M.n_lambda + 1);
  stage(T.fstar, M.frac_E_stars, (size_t)M.n_lambda);
  stage(T.cdf, M.cdf, (size_t)M.n_lambda * M.n_T);
  stage(T.cost, M.cos_tab, (size_t)M.nang + 1);
  stage(T.albedo, M.albedo, (size_t)M.n_lambda);
  stage(T.prob, M.prob_s11, (size_t)(M.nang + 1) * (M.p_lambda_fixed ? 1 : M.n_lambda));
  stage(T.g, M.tab_g, (size_t)M.n_lambda);
  for (int i = threadIdx.x; i < M.n_lambda; i += blockDim.x) T.nsent[i] = 0u;
}

// n_phot_envoyes(lambda) += 1 (dust_transfer.f90:536).  One global FP64 atomic per packet onto these few addresses
// was THE bound of the whole thermal loop (same-address atomics serialise at the memory side at ~3.4 ns each:
// 2.9e8 packets/s whatever the kernel did otherwise; measured r02: 346 -> 186 ms per 1e8 packets without it), so the
// workgroup counts in LDS and adds its histogram to the global array once, at the end of the launch.
__device__ inline void lds_count_sent(const Lds& T, int lambda) { atomicAdd(&T.nsent[lambda - 1], 1u); }
// (call after a __syncthreads() that follows the last emission of the workgroup)
__device__ inline void lds_flush_sent(const Lds& T, const DevModel& M, double* n_sent) {
  for (int i = threadIdx.x; i < M.n_lambda; i += blockDim.x) {
    const unsigned int c = T.nsent[i];
    if (c) unsafeAtomicAdd(&n_sent[i], (double)c);
  }
}

// the tables of the SED-mode kernels (lds_carve(..., mono = true)): T.prob is column p_lambda of prob_s11
__device__ inline void lds_stage_mono(const Lds& T, const DevModel& M, int p_lambda) {
  stage(T.r_lim_2, M.r_lim_2, (size_t)M.n_rad + 1);
  stage(T.zmax, M.zmax, (size_t)M.n_rad);
  stage(T.ch, M.ch, (size_t)M.n_rad);
  for (int i = threadIdx.x; i < M.n_rad; i += blockDim.x) T.rzn[i] = (double)M.nz / M.zmax[i];
  stage(T.tan_phi, M.tan_phi_lim, (size_t)M.n_az);
  stage(T.kappa, M.kappa, (size_t)M.n_lambda);
  stage(T.kabs, M.kappa_abs, (size_t)M.n_lambda);
  stage(T.cost, M.cos_tab, (size_t)M.nang + 1);
  stage(T.albedo, M.albedo, (size_t)M.n_lambda);
  stage(T.prob, M.prob_s11 + (size_t)(M.nang + 1) * (p_lambda - 1), (size_t)M.nang + 1);
  stage(T.g, M.tab_g, (size_t)M.n_lambda);
}

// lvariable_dust: the table pointers of one cell class -- the LDS copies hold one class only, so the per-class rows
// are read from HBM (a gather per crossing / interaction: the memory regime of SURVEY 8f rank 4)
__device__ inline Lds class_tables(const Lds& T, const DevModel& M, int cls) {
  Lds V = T;
  V.kappa = const_cast<double*>(M.v_kappa) + (size_t)cls * M.n_lambda;
  V.kabs = const_cast<double*>(M.v_kabs) + (size_t)cls * M.n_lambda;
  V.albedo = const_cast<float*>(M.v_albedo) + (size_t)cls * M.n_lambda;
  V.lq = const_cast<double*>(M.v_lq) + (size_t)cls * M.n_T;
  V.cdf = const_cast<double*>(M.v_cdf) + (size_t)cls * M.n_T * M.n_lambda;
  if (M.v_scatt) {
    V.prob = const_cast<float*>(M.v_prob) + (size_t)cls * (M.p_lambda_fixed ? 1 : M.n_lambda) * (M.nang + 1);
    V.g = const_cast<float*>(M.v_g) + (size_t)cls * M.n_lambda;
  }
  return V;
}

// the per-cell opacity pairs of the variable-dust role kernel (DevModel::v_kk): one thread per (cell, wavelength)
static __global__ void k_build_vkk(const DevModel M, double2* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = ((size_t)M.n_cells + 1) * M.n_lambda;
  if (i >= n) return;
  const int ic = (int)(i / M.n_lambda), l = (int)(i - (size_t)ic * M.n_lambda);
  double2 v = make_double2(0.0, 0.0);
  if (ic < M.n_cells) {
    const size_t row = (size_t)M.cell_class[ic] * M.n_lambda + l;
    v.x = M.v_kappa[row] * M.kappa_factor[ic];
    v.y = M.v_kabs[row];
  }
  out[i] = v;
}

// ---------------------------------------------------------------------------
// Cell identity
// ---------------------------------------------------------------------------
// z_lim(ri,j) of the reference (cylindrical_grid.f90:459-465,493): uniform in z
__device__ inline double z_lim_of(const Lds& T, int nz, int ri, int j) {
  if (j <= nz) return ((double)j - 1.0) * T.ch[ri - 1];
  if (j == nz + 1) return T.zmax[ri - 1];
  return 1.00000001504746621988e+30;  // real 1.0e30
}

template <bool L3D>
__device__ inline bool is_real_cell(int n_rad, int nz, int ri, int zj) {
  int az = L3D ? (zj < 0 ? -zj : zj) : zj;
  return (ri >= 1) & (ri <= n_rad) & (az >= 1) & (az <= nz);
}

// 0-based index of a real cell, the closed form of cell_map
// (cylindrical_grid.f90:90-107: k outermost, then j skipping 0, then i)
template <bool L3D>
__device__ inline int cell_index(int n_rad, int nz, int ri, int zj, int k) {
  if (L3D) {
    int jj = zj < 0 ? zj + nz : zj + nz - 1;
    return (ri - 1) + n_rad * (jj + 2 * nz * (k - 1));
  }
  return (ri - 1) + n_rad * (zj - 1);
}

// the icell the reference would return for (ri,zj,k), real or virtual:
// closed form of build_cylindrical_cell_mapping (cylindrical_grid.f90:45-179)
__host__ __device__ inline int icell_of(int n_rad, int nz, int n_az, int l3D, int i, int j, int k) {
  const int n_cells = l3D ? 2 * n_rad * nz * n_az : n_rad * nz;
  const int jlo = l3D ? -nz - 1 : 0, jhi = nz + 1;
  const int aj = j < 0 ? -j : j;
  if (i >= 1 && i <= n_rad && aj >= 1 && aj <= nz && (l3D || j > 0)) {
    if (l3D) {
      int jj = j < 0 ? j + nz : j + nz - 1;
      return 1 + (i - 1) + n_rad * (jj + 2 * nz * (k - 1));
    }
    return 1 + (i - 1) + n_rad * (j - 1);
  }
  if (j == jlo || j == jhi) {  // first virtual group (:123-141)
    int jrow = (j == jlo) ? 0 : 1;
    return n_cells + 1 + i + (n_rad + 2) * (jrow + 2 * (k - 1));
  }
  // second virtual group (:143-167): i = 0 or n_rad+1, j real
  const int base = n_cells + (n_rad + 2) * 2 * n_az;
  const int nj = l3D ? 2 * nz : nz;
  int jj = l3D ? (j < 0 ? j + nz : j + nz - 1) : j - 1;
  int ii = (i == 0) ? 0 : 1;
  return base + 1 + ii + 2 * (jj + nj * (k - 1));
}

// zj from |z| through default real (cylindrical_grid.f90:868,1116)
__device__ inline int zj_from_z_real(const Lds& T, int nz, double absz, int ri) {
  float q = (float)(absz / T.zmax[ri - 1] * (double)nz);
  const float mi = 2147483648.0f * (1.0f - 1.0e-5f);  // max_int (constants.f90:159)
  if (!(q < mi)) q = mi;
  return (int)floorf(q) + 1;
}

__device__ inline double modulo_d(double a, double p) { return a - floor(a / p) * p; }


// The azimuthal sector of a point, k = floor(phi / 2 pi * n_az) + 1 with phi = modulo(atan2(y, x), 2 pi)
// (cylindrical_grid.f90:1121-1126, 399-411), decided in default real where that is certain: an arctangent good to 1e-6
// rad (odd polynomial of the smaller over the larger coordinate: 7e-7 against atan over 2e6 arguments on the host, the
// reciprocal and the constants' roundings on top) gives the quotient q; unless q lies within delta = 2e-6 n_az + 1e-5 of
// an integer -- four times the error bound -- floor(q) is the floor of the reference's FP64 expression.  Returns
// false in that sliver (and for points the conversion loses: NaN compares false), where the caller runs the reference's
// expression with its atan2 (~170 vector instructions; 1.2 exits from the central hole per packet of ref4.1_3D kept that
// branch in a fifth of the flying loop's iterations, and every stopping point's index_cell pays one).
__device__ __forceinline__ bool az_sector_certain(double x, double y, int n_az, int& k) {
#ifdef MCGPU_NO_FAST_SECTOR   // (A/B builds)
  k = 0;
  return false;
#endif
  const float xf = (float)x, yf = (float)y;
  const float ax = fabsf(xf), ay = fabsf(yf);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
#ifdef MCGPU_LANE_EMULATION
  const float t = mn / mx;
#else
  const float t = mn * __builtin_amdgcn_rcpf(mx);
#endif
  const float t2 = t * t;
  float a = 0.00809729f;
  a = fmaf(a, t2, -0.03775171f);
  a = fmaf(a, t2, 0.0847597f);
  a = fmaf(a, t2, -0.13537675f);
  a = fmaf(a, t2, 0.19895026f);
  a = fmaf(a, t2, -0.33327976f);
  a = fmaf(a, t2, 0.99999972f);
  a = a * t;
  a = (ay > ax) ? 1.57079632679f - a : a;
  a = (xf < 0.0f) ? 3.14159265359f - a : a;
  a = (yf < 0.0f) ? 6.28318530718f - a : a;
  const float q = a * ((float)n_az * 0.159154943092f);
  const float fl = floorf(q);
  const float fr = q - fl;
  const float delta = 2.0e-6f * (float)n_az + 1.0e-5f;
  k = (int)fl + 1;
  return (fr > delta) && (fr < 1.0f - delta) && (k >= 1) && (k <= n_az);
}

// The sector of a point, out of line: the callers sit in the flying loop's rare branches (an exit from the central hole,
// 0.6 % of the crossings) and in the re-indexing of a stopping point, and inlined there the arctangent's temporaries
// cost the binned 3D role kernel 60 more spilled registers (68 -> 130 at its 168) and 4 % of its speed -- more than the
// atan2 they replace.  recip_form: phi * (1 / 2 pi) * n_az (the crossing, cylindrical_grid.f90:1125), else phi / (2 pi) * n_az
// (indice_cellule_3D, :405) -- the reference's two spellings, kept apart for the sliver where they could differ.
__device__ __attribute__((noinline)) int az_sector(double x, double y, int n_az, bool recip_form) {
  int k;
  if (__builtin_expect(!az_sector_certain(x, y, n_az, k), 0)) {
    const double a = atan2(y, x);
    const double phi = a - floor(a / (2 * PI)) * (2 * PI);   // (modulo_d)
    const double q = recip_form ? phi * (1.0 / (2.0 * PI)) * (double)(float)n_az : phi / (2 * PI) * (double)(float)n_az;
    k = (int)floor(q) + 1;
    if (k == n_az + 1) k = n_az;
  }
  return k;
}

// index_cell_cyl (cylindrical_grid.f90:833-890) -> (ri,zj,k)
// ri_hint (1 .. n_rad, else none): the radial index the point is expected in -- a stopping point lies in the cell its flight
// was crossing but for rounding --; where r_lim_2(hint - 1) < r^2 <= r_lim_2(hint) holds it IS what the bisection returns
// (the same comparisons), and its seven dependent table reads are skipped
template <bool L3D>
__device__ inline void index_cell(const Lds& T, const DevModel& M, double x, double y, double z,
                                  int& ri_out, int& zj_out, int& k_out, int ri_hint = -1) {
  double r2 = x * x + y * y;
  if (r2 < T.r_lim_2[0]) {
    ri_out = 0; zj_out = 1; k_out = 1;
  } else if (r2 > M.Rmax2) {
    ri_out = M.n_rad + 1; zj_out = 1; k_out = 1;
  } else {
    int ri;
    const bool hinted = ri_hint >= 1 && ri_hint <= M.n_rad;
    const int hs = hinted ? ri_hint : 1;
    if (hinted && (hs == 1 || r2 > T.r_lim_2[hs - 1]) && (hs == M.n_rad || !(r2 > T.r_lim_2[hs]))) ri = hs - 1;
    else {
      int ri_min = 0, ri_max = M.n_rad;
      ri = (ri_min + ri_max) / 2;
      while ((ri_max - ri_min) > 1) {
        if (r2 > T.r_lim_2[ri]) ri_min = ri; else ri_max = ri;
        ri = (ri_min + ri_max) / 2;
      }
    }
    ri_out = ri + 1;
    int zj = zj_from_z_real(T, M.nz, fabs(z), ri_out);
    if (zj > M.nz) zj = M.nz + 1;
    k_out = 1;
    if (L3D) {
      if (z < 0.0) zj = -zj;
      if (z != 0.0) k_out = az_sector(x, y, M.n_az, false);
    }
    zj_out = zj;
  }
}

// zj of a point (|z| = absz) in radial column ri, capped at nz+1: the reference's
//   floor(min(real(abs(z1)/zmax(ri1)*nz), max_int)) + 1        (cylindrical_grid.f90:1116)
// The default-real rounding only matters within ~nz*6e-8 of an integer, so the common case
// is one multiply by nz/zmax; the literal expression runs when the quotient is that close.
__device__ inline int zj_capped(const Lds& T, int nz, double absz, int ri) {
  const double qd = absz * T.rzn[ri - 1];
  const double fl = floor(qd);
  const double fr = qd - fl;
  int zj;
  if (!(qd < (double)nz + 0.5)) zj = nz + 1;         // far above the grid (or not finite)
  else if (fr < 1.0e-4 || fr > 1.0 - 1.0e-4) zj = zj_from_z_real(T, nz, absz, ri);
  else zj = (int)fl + 1;
  return zj > nz ? nz + 1 : zj;
}

// cross_cylindrical_cell (cylindrical_grid.f90:918-1175), the same arithmetic as
// the reference's branch tree flattened into selects: a wavefront
// executes ONE instruction stream whatever mix of inward/outward, up/down, radial/vertical
// moves its 64 packets make.  Every value that decides an index or a position is computed by
// the same expression as in the reference.
template <bool L3D>
__device__ inline void cross_cell_lean(const Lds& T, const DevModel& M, double x0, double y0, double z0,
                                       double u, double v, double w, double inv_a, double inv_w, int ri0,
                                       int zj0, int k0, double& x1, double& y1, double& z1, int& ri1,
                                       int& zj1, int& k1, double& l) {
  const int nz = M.nz, n_rad = M.n_rad, n_az = M.n_az;
  const double cm = 1.0 - GRID_PREC, cp = 1.0 + GRID_PREC;
  const bool hole = (ri0 == 0);

  // 1) radial wall (:959-1000)
  const double r_2 = x0 * x0 + y0 * y0;
  const double dot = x0 * u + y0 * v;
  const double b = dot * inv_a;
  const double rl_in = T.r_lim_2[hole ? 0 : ri0 - 1];
  const double rl_out = T.r_lim_2[hole ? 0 : ri0];
  const double c_in = (r_2 - (hole ? rl_in : rl_in * cm)) * inv_a;
  const double c_out = (r_2 - rl_out * cp) * inv_a;
  const double bb = b * b;
  const double d_in = bb - c_in;
  const double d_out = fmax(bb - c_out, 0.0);
  const bool use_in = hole || ((dot < 0.0) && !(d_in < 0.0));
  const double delta = use_in ? d_in : d_out;
  const int delta_rad = (use_in && !hole) ? -1 : 1;
  const double rac = sqrt(delta);
  const double s1 = (-b - rac) * cp, s2 = (-b + rac) * cp;
  double s = (s1 < 0.0) ? s2 : ((s1 == 0.0) ? GRID_PREC : s1);
  if (hole) s = s2;

  // 2) vertical wall (:1003-1055)
  const int azj = zj0 < 0 ? -zj0 : zj0;
  const double dz = w * z0;
  const bool away = dz > 0.0;
  const bool top = (azj == nz + 1);
  const bool flip2d = !L3D && !away && (zj0 == 1);  // 2D: through the midplane to the mirror side
  const int jsel = away ? azj + 1 : (L3D ? azj : (zj0 == 1 ? 2 : zj0));
  double zmag = (jsel <= nz) ? ((double)jsel - 1.0) * T.ch[hole ? 0 : ri0 - 1]
                             : ((jsel == nz + 1) ? T.zmax[hole ? 0 : ri0 - 1] : 1.00000001504746621988e+30);
  zmag = zmag * (away ? cp : cm);
  if (away && top) zmag = 1.0e10;
  const bool neg = (z0 < 0.0) != flip2d;
  const double zl = neg ? -zmag : zmag;
  int delta_zj;
  if (L3D) {
    if (away) delta_zj = top ? 0 : ((z0 < 0.0) ? -1 : 1);
    else delta_zj = (z0 > 0.0) ? ((zj0 == 1) ? -2 : -1) : ((zj0 == -1) ? 2 : 1);
  } else {
    delta_zj = away ? (top ? 0 : 1) : ((zj0 == 1) ? 1 : -1);
  }
  double t = (zl - z0) * inv_w;
  if (t < 0.0) t = GRID_PREC;
  if (dz == 0.0) t = 1.0e10;
  if (hole) t = HUGE_REAL;

  // 3) azimuthal wall (:1058-1094)
  double t_phi = HUGE_REAL;
  int delta_phi = 0;
  if (L3D) {
    const double r1e30 = 1.00000001504746621988e+30;
    const double dp = x0 * v - y0 * u;
    delta_phi = (dp > 0.0) ? 1 : -1;
    int kk = (dp > 0.0) ? k0 : k0 - 1;
    if (kk == 0) kk = n_az;
    const double tan_lim = T.tan_phi[kk - 1];
    const double den = v - u * tan_lim;
    double tp = (fabs(den) > (double)1.0e-6f) ? -(y0 - x0 * tan_lim) / den : r1e30;
    if (tan_lim > 1.0e299) tp = (fabs(u) > (double)1e-6f) ? -x0 / u : r1e30;
    if (tp < 0.0) tp = r1e30;
    if (fabs(dp) < (double)1.0e-10f) tp = r1e30;
    t_phi = hole ? HUGE_REAL : tp;
  }

  // 4) nearest wall (:1098-1156)
  const bool rad = (s < t) && (s < t_phi);
  const bool vert = !rad && (t < t_phi);
  l = rad ? s : (vert ? t : t_phi);
  const double dv = (rad || vert) ? l : cp * t_phi;
  x1 = x0 + dv * u;
  y1 = y0 + dv * v;
  // products rounded before the sum, like the reference build (no FMA contraction): at the midplane zl = 0 has
  // no grid_prec margin, so whether z1 comes out as exactly 0 (-> sign(grid_prec,w), :1158-1165) or as a rounding
  // residue of either sign is decided by the rounding of t*w
  z1 = nd_add(z0, nd_mul(dv, w));
  ri1 = rad ? ri0 + delta_rad : ri0;
  k1 = k0;
  if (rad) {
    if (ri1 == 0) {
      zj1 = 1;
      k1 = 1;
    } else if (ri1 > n_rad) {
      zj1 = zj0;
    } else {
      int zj = zj_capped(T, nz, fabs(z1), ri1);
      if (L3D && (z1 < 0.0)) zj = -zj;
      zj1 = zj;
    }
    if (L3D && hole) k1 = az_sector(x1, y1, n_az, true);
  } else if (vert) {
    zj1 = zj0 + delta_zj;
    if (L3D && M.midplane_snap && (delta_zj == 2 || delta_zj == -2)) z1 = copysign(GRID_PREC, w);
  } else {
    int zj = (int)floor(fabs(z1) / T.zmax[ri1 - 1] * (double)nz) + 1;
    if (zj > nz) zj = nz + 1;
    if (z1 < 0.0) zj = -zj;
    zj1 = zj;
    int kk = k0 + delta_phi;
    if (kk == 0) kk = n_az;
    if (kk == n_az + 1) kk = 1;
    k1 = kk;
  }
  if (z1 == 0.0) z1 = L3D ? copysign(GRID_PREC, w) : GRID_PREC;
}

// move_to_grid_cyl (cylindrical_grid.f90:1284-1411)
template <bool L3D>
__device__ inline bool move_to_grid(const Lds& T, const DevModel& M, double& x, double& y, double& z,
                                    double u, double v, double w, int& ri, int& zj, int& k) {
  const double correct_moins = 1.0 - 1.0e-10;
  double x0 = x, y0 = y, z0 = z, a, inv_a, inv_w, r_2, b, c, delta, s1, s2, t1, t2, delta_vol;
  a = u * u + v * v;
  inv_a = (a > TINY_REAL) ? 1.0 / a : HUGE_REAL;
  inv_w = (fabs(w) > TINY_REAL) ? 1.0 / w : copysign(HUGE_DP, w);
  r_2 = x0 * x0 + y0 * y0;
  b = (x0 * u + y0 * v) * inv_a;
  c = (r_2 - T.r_lim_2[M.n_rad] * correct_moins) * inv_a;
  delta = b * b - c;
  if (delta < 0.0) {
    s1 = HUGE_REAL; s2 = HUGE_REAL;
  } else {
    double rac = sqrt(delta);
    s1 = -b - rac;
    s2 = -b + rac;
  }
  double dotprod = w * z0;
  if (fabs(dotprod) < TINY_REAL) {
    t1 = HUGE_REAL; t2 = HUGE_REAL;
  } else {
    double zl = M.zmaxmax * correct_moins;
    double zlim1 = (z0 > 0.0) ? zl : -zl;
    t1 = (zlim1 - z0) * inv_w;
    t2 = (-zlim1 - z0) * inv_w;
  }
  if (t1 > (double)1e20f) {
    if (s1 > (double)1e20f) return false;
  }
  if (t1 > s1) {
    if (t1 > s2) {
      delta_vol = s1;
      double z1 = z0 + delta_vol * w;
      if (fabs(z1) > M.zmaxmax) return false;
    } else {
      delta_vol = t1;
    }
  } else {
    if (t2 < s1) return false;
    delta_vol = s1;
  }
  x = x0 + delta_vol * u;
  y = y0 + delta_vol * v;
  z = z0 + delta_vol * w;
  index_cell<L3D>(T, M, x, y, z, ri, zj, k);
  return true;
}

// pos_em_cell_cyl (cylindrical_grid.f90:1415-1466)
template <bool L3D>
__device__ inline void pos_em_cell(const Lds& T, const DevModel& M, int ri, int zj, int k, float rand1,
                                   float rand2, float rand3, double& x, double& y, double& z) {
  double r = sqrt(T.r_lim_2[ri - 1] + (double)rand1 * (T.r_lim_2[ri] - T.r_lim_2[ri - 1]));
  const int nz = M.nz;
  if (L3D) {
    int aj = zj > 0 ? zj : -zj;
    double zz = z_lim_of(T, nz, ri, aj) + (double)rand2 * (z_lim_of(T, nz, ri, aj + 1) - z_lim_of(T, nz, ri, aj));
    z = zj > 0 ? zz : -zz;
  } else {
    double z0 = z_lim_of(T, nz, ri, zj), z1 = z_lim_of(T, nz, ri, zj + 1);
    if ((double)rand2 > 0.5) z = z0 + (2.0 * ((double)rand2 - 0.5)) * (z1 - z0);
    else z = -(z0 + (2.0 * (double)rand2) * (z1 - z0));
  }
  double phi = 2.0 * PI * ((double)k - 1.0 + (double)rand3) / (double)M.n_az;
  x = r * cos(phi);
  y = r * sin(phi);
}

// ---------------------------------------------------------------------------
// Spherical grid operators (spherical_grid.f90) on (ri, thetaj, phik): the same cell identity and mapping as the
// cylindrical grid (build_cylindrical_cell_mapping serves both, grid.f90:356); thetaj counts the polar bins from the
// midplane, negative below it in 3D.  Only the outer radius is an exit (test_exit_grid_sph, :24-44).
// ---------------------------------------------------------------------------
constexpr double PREC_GRILLE_SPH = 1.0e-7;  // spherical_grid.f90:19

// indice_cellule_sph_theta (:129-178) = the polar / azimuthal part of index_cell_sph (:83-120)
template <bool L3D>
__device__ inline void sph_theta_phi(const DevModel& M, double x, double y, double z, int& tj_out, int& k_out) {
  const double r02 = x * x + y * y;
  const double tan_theta = (r02 > TINY_DP) ? fabs(z) / sqrt(r02) : (double)1.0e30f;
  int tmin = 0, tmax = M.nz, tj = (tmin + tmax) / 2;
  while ((tmax - tmin) > 1) {
    if (tan_theta > M.tan_theta_lim[tj]) tmin = tj; else tmax = tj;
    tj = (tmin + tmax) / 2;
  }
  tj_out = tj + 1;
  k_out = 1;
  if (L3D) {
    if (z < 0.0) tj_out = -tj_out;
    if (z != 0.0) k_out = az_sector(x, y, M.n_az, false);
  }
}

// index_cell_sph (:48-125)
template <bool L3D>
__device__ inline void index_cell_sph(const Lds& T, const DevModel& M, double x, double y, double z, int& ri_out,
                                      int& tj_out, int& k_out) {
  const double r2 = x * x + y * y + z * z;
  if (r2 < T.r_lim_2[0]) {
    ri_out = 0; tj_out = 1; k_out = 1;
  } else if (r2 > M.Rmax2) {
    ri_out = M.n_rad + 1; tj_out = 1; k_out = 1;
  } else {
    int ri_min = 0, ri_max = M.n_rad, ri = (ri_min + ri_max) / 2;
    while ((ri_max - ri_min) > 1) {
      if (r2 > T.r_lim_2[ri]) ri_min = ri; else ri_max = ri;
      ri = (ri_min + ri_max) / 2;
    }
    ri_out = ri + 1;
    sph_theta_phi<L3D>(M, x, y, z, tj_out, k_out);
  }
}

// one polar cone (:247-307): the smallest positive root of the crossing with tan(theta) = tan_lim.  The
// discriminant is a difference of products of nearly equal size: kept unfused, like the reference build.
__device__ inline double sph_theta_root(double x0, double y0, double z0, double u, double v, double w, double tan_lim) {
  const double precision = 1.0e-15;
  const double tan2 = nd_mul(tan_lim, tan_lim);
  const double a_theta = nd_add(nd_mul(w, w), -nd_mul(tan2, nd_add(nd_mul(u, u), nd_mul(v, v))));
  const double a_theta_m1 = 1.0 / a_theta;
  const double b_theta = nd_add(nd_mul(w, z0), -nd_mul(tan2, nd_add(nd_mul(x0, u), nd_mul(y0, v))));
  const double c_theta = nd_add(nd_mul(z0, z0), -nd_mul(tan2, nd_add(nd_mul(x0, x0), nd_mul(y0, y0))));
  const double delta = nd_add(nd_mul(b_theta, b_theta), -nd_mul(a_theta, c_theta));
  if (delta < 0.0) return 1.0e30;
  const double rac = sqrt(delta);
  const double t_1 = nd_mul(-b_theta - rac, a_theta_m1);
  const double t_2 = nd_mul(-b_theta + rac, a_theta_m1);
  if (t_1 <= precision) return (t_2 <= precision) ? 1.0e30 : t_2;
  if (t_2 <= precision) return t_1;
  return t_1 < t_2 ? t_1 : t_2;
}

// cross_spherical_cell (:182-446)
template <bool L3D>
__device__ inline void cross_cell_sph(const Lds& T, const DevModel& M, double x0, double y0, double z0, double u,
                                      double v, double w, int ri0, int tj0, int k0, double& x1, double& y1,
                                      double& z1, int& ri1, int& tj1, int& k1, double& l) {
  const double cm = 1.0 - PREC_GRILLE_SPH, cp = 1.0 + PREC_GRILLE_SPH;
  const double r0_2 = nd_add(nd_add(nd_mul(x0, x0), nd_mul(y0, y0)), nd_mul(z0, z0));
  const double b = nd_add(nd_add(nd_mul(x0, u), nd_mul(y0, v)), nd_mul(z0, w));
  double s, t, t_phi;
  int delta_rad, delta_theta = 0, delta_phi = 0;
  if (ri0 == 0) {  // inside the inner boundary: the one positive root with rmin
    const double c = nd_add(r0_2, -nd_mul(T.r_lim_2[0], cp));
    const double rac = sqrt(nd_add(nd_mul(b, b), -c));
    s = nd_mul(-b + rac, cp);
    t = HUGE_REAL;
    t_phi = HUGE_REAL;
    delta_rad = 1;
  } else {
    double delta;
    if (b < 0.0) {
      const double c = nd_add(r0_2, -nd_mul(T.r_lim_2[ri0 - 1], cm));
      delta = nd_add(nd_mul(b, b), -c);
      if (delta < 0.0) {
        const double c2 = nd_add(r0_2, -nd_mul(T.r_lim_2[ri0], cp));
        delta = fmax(nd_add(nd_mul(b, b), -c2), 0.0);
        delta_rad = 1;
      } else {
        delta_rad = -1;
      }
    } else {
      const double c = nd_add(r0_2, -nd_mul(T.r_lim_2[ri0], cp));
      delta = fmax(nd_add(nd_mul(b, b), -c), 0.0);
      delta_rad = 1;
    }
    const double rac = sqrt(delta);
    s = -b - rac;
    if (s < 0.0) s = -b + rac;
    else if (s == 0.0) s = GRID_PREC;
    const int aj = tj0 < 0 ? -tj0 : tj0;
    const double sg = (z0 >= 0.0) ? 1.0 : -1.0;
    const double t1 = sph_theta_root(x0, y0, z0, u, v, w, sg * nd_mul(M.tan_theta_lim[aj], cp));
    const double t2 = sph_theta_root(x0, y0, z0, u, v, w, sg * nd_mul(M.tan_theta_lim[aj - 1], cm));
    if (t1 < t2) {
      t = t1;
      delta_theta = (aj == M.nz) ? 0 : 1;
    } else {
      t = t2;
      delta_theta = (aj == 1) ? 0 : -1;
    }
    t_phi = HUGE_REAL;
    if (L3D) {
      const double r1e30 = (double)1.0e30f;
      const double dp = nd_add(nd_mul(x0, v), -nd_mul(y0, u));
      if (fabs(dp) < (double)1.0e-10f) {
        t_phi = r1e30;
      } else {
        int kk = (dp > 0.0) ? k0 : k0 - 1;
        if (kk == 0) kk = M.n_az;
        delta_phi = (dp > 0.0) ? 1 : -1;
        const double tan_lim = T.tan_phi[kk - 1];
        if (tan_lim > 1.0e299) {
          t_phi = -x0 / u;
        } else {
          const double den = nd_add(v, -nd_mul(u, tan_lim));
          if (fabs(den) > (double)1.0e-6f) t_phi = -nd_add(y0, -nd_mul(x0, tan_lim)) / den;
          else { t_phi = r1e30; delta_phi = 0; }
        }
        if (t_phi < 0.0) { t_phi = r1e30; delta_phi = 0; }
      }
    }
  }
  if ((s < t) && (s < t_phi)) {
    l = s;
    x1 = nd_add(x0, nd_mul(s, u)); y1 = nd_add(y0, nd_mul(s, v)); z1 = nd_add(z0, nd_mul(s, w));
    ri1 = ri0 + delta_rad;
    tj1 = tj0; k1 = k0;
    if (ri0 == 0) sph_theta_phi<L3D>(M, x1, y1, z1, tj1, k1);
    if (ri1 == 0) { tj1 = 1; k1 = 1; }
  } else if (t < t_phi) {
    l = t;
    x1 = nd_add(x0, nd_mul(t, u)); y1 = nd_add(y0, nd_mul(t, v)); z1 = nd_add(z0, nd_mul(t, w));
    ri1 = ri0;
    tj1 = (tj0 < 0 ? -tj0 : tj0) + delta_theta;
    if (L3D && z1 < 0.0) tj1 = -tj1;
    k1 = k0;
  } else {
    l = t_phi;
    const double dv = nd_mul(cp, t_phi);
    x1 = nd_add(x0, nd_mul(dv, u)); y1 = nd_add(y0, nd_mul(dv, v)); z1 = nd_add(z0, nd_mul(dv, w));
    ri1 = ri0; tj1 = tj0;
    int kk = k0 + delta_phi;
    if (kk == 0) kk = M.n_az;
    if (kk == M.n_az + 1) kk = 1;
    k1 = kk;
  }
  if (z1 == 0.0) z1 = GRID_PREC;
}

// move_to_grid_sph (:562-615)
template <bool L3D>
__device__ inline bool move_to_grid_sph(const Lds& T, const DevModel& M, double& x, double& y, double& z, double u,
                                        double v, double w, int& ri, int& tj, int& k) {
  const double correct_moins = 1.0 - 1.0e-10;
  const double x0 = x, y0 = y, z0 = z;
  const double r0_2 = nd_add(nd_add(nd_mul(x0, x0), nd_mul(y0, y0)), nd_mul(z0, z0));
  const double b = nd_add(nd_add(nd_mul(x0, u), nd_mul(y0, v)), nd_mul(z0, w));
  const double c = nd_add(r0_2, -nd_mul(T.r_lim_2[M.n_rad], correct_moins));
  const double delta = nd_add(nd_mul(b, b), -c);
  if (delta < 0.0) return false;
  const double s1 = -b - sqrt(delta);
  x = nd_add(x0, nd_mul(s1, u)); y = nd_add(y0, nd_mul(s1, v)); z = nd_add(z0, nd_mul(s1, w));
  index_cell_sph<L3D>(T, M, x, y, z, ri, tj, k);
  return true;
}

// pos_em_cell_sph (:619-699); in 3D every packet starts in the upper hemisphere, as in the reference (:647)
template <bool L3D>
__device__ inline void pos_em_cell_sph(const DevModel& M, int ri, int tj, int k, float rand1, float rand2, float rand3,
                                       double& x, double& y, double& z) {
  const double r = pow(M.r_lim_3[ri - 1] + (double)rand1 * (M.r_lim_3[ri] - M.r_lim_3[ri - 1]), 1.0 / 3.0);
  double theta;
  if (L3D) {
    const int aj = tj < 0 ? -tj : tj;
    theta = M.theta_lim[aj - 1] + (double)rand2 * (M.theta_lim[aj] - M.theta_lim[aj - 1]);
  } else {
    if ((double)rand2 > 0.5) theta = M.theta_lim[tj - 1] + (2.0 * ((double)rand2 - 0.5)) * (M.theta_lim[tj] - M.theta_lim[tj - 1]);
    else theta = -(M.theta_lim[tj - 1] + (2.0 * (double)rand2) * (M.theta_lim[tj] - M.theta_lim[tj - 1]));
  }
  const double phi = 2.0 * PI * ((double)(float)k - 1.0 + (double)rand3) / (double)(float)M.n_az;
  double st, ct, sp, cph;
  sincos(theta, &st, &ct);
  sincos(phi, &sp, &cph);
  z = r * st;
  const double rc = r * ct;
  x = rc * cph;
  y = rc * sp;
}

// cdapres (utils.f90:1636-1690)
// (cdapres with sin and cos of phi given: the tail kernel computes them ahead, mc_tail.hip.h)
__device__ inline void cdapres_sc(double cospsi, double sphi, double cphi, double u0, double v0, double w0, double& u1,
                                  double& v1, double& w1);
__device__ inline void cdapres(double cospsi, double phi, double u0, double v0, double w0, double& u1,
                               double& v1, double& w1) {
  double sphi, cphi;
  sincos(phi, &sphi, &cphi);
  cdapres_sc(cospsi, sphi, cphi, u0, v0, w0, u1, v1, w1);
}
// sin and cos of pi * x: what every azimuth of the packet loop is -- phi = pi (2 rand - 1) (utils.f90 / dust_transfer.f90:
// 1297, random_numbers.f90:44) -- without forming pi * x first: sincospi reduces its argument exactly (71 vector
// instructions against the 158 of sincos); the same numbers to the rounding of the product pi * x
__device__ inline void sincos_pi(double x, double* s, double* c) {
#ifdef MCGPU_LANE_EMULATION
  sincos(PI * x, s, c);
#else
  sincospi(x, s, c);
#endif
}
// cdapres with the azimuth given in units of pi
__device__ inline void cdapres_pi(double cospsi, double phi_over_pi, double u0, double v0, double w0, double& u1,
                                  double& v1, double& w1) {
  double sphi, cphi;
  sincos_pi(phi_over_pi, &sphi, &cphi);
  cdapres_sc(cospsi, sphi, cphi, u0, v0, w0, u1, v1, w1);
}
__device__ inline void cdapres_sc(double cospsi, double sphi, double cphi, double u0, double v0, double w0, double& u1,
                                  double& v1, double& w1) {
  double cpsi = cospsi;
  double spsi = sqrt(1.0 - cpsi * cpsi);
  double a = spsi * cphi;
  double b = spsi * sphi;
  if (fabs(w0) <= (double)0.999999f) {
    double c = sqrt(1.0 - w0 * w0);
    double cm1 = 1.0 / c;
    double aw0 = a * w0;
    u1 = (aw0 * u0 - b * v0) * cm1 + cpsi * u0;
    v1 = (aw0 * v0 + b * u0) * cm1 + cpsi * v0;
    w1 = cpsi * w0 - a * c;
  } else {
    u1 = a;
    v1 = b;
    w1 = cpsi;
  }
}

// rotation (utils.f90:553-601)
__device__ inline void rotation(double xinit, double yinit, double zinit, double u1, double v1,
                                double w1, double& xfin, double& yfin, double& zfin) {
  double cost, sint, sing;
  if (w1 > 0.999999999) {
    cost = 1.0; sint = 0.0; sing = 0.0;
  } else if (fabs(u1) < TINY_REAL) {
    cost = 0.0; sint = 1.0;
    sing = sqrt(1.0 - w1 * w1);
  } else {
    // theta = atan2(v1, u1), cost = cos(theta), sint = sin(theta) in the reference (:577-579): the same two numbers
    // without the two transcendental calls (250 FP64 instructions per polarised scattering); they differ from the
    // composition's by its own rounding, 1e-16
    const double h = sqrt(u1 * u1 + v1 * v1);
    cost = u1 / h;
    sint = v1 / h;
    sing = sqrt(1.0 - w1 * w1);
  }
  double prod = cost * xinit + sint * yinit;
  xfin = sing * prod + w1 * zinit;
  yfin = cost * yinit - sint * xinit;
  zfin = sing * zinit - w1 * prod;
}

// update_Stokes (scattering.f90:1187-1298) with get_Mueller_matrix_per_cell
// (:1328-1350) folded in: M11 = 1, M12 = M21, M34 = -M43.
// (two halves, so that the tail kernel can compute the first -- the expensive one: a rotation, two square roots, three
// divisions -- for many scatterings at once and apply the second in sequence, mc_tail.hip.h)
// the rotation of the Stokes vector into the scattering plane: cos and sin of omega (scattering.f90:1187-1226)
__device__ inline void stokes_rotation(double u0, double v0, double w0, double u1, double v1, double w1, double& cw, double& sw) {
  double v1pi, v1pj, v1pk;
  rotation(u0, v0, w0, u1, v1, w1, v1pi, v1pj, v1pk);
  float xnyp = (float)sqrt(v1pk * v1pk + v1pj * v1pj);
  float costhet;
  if (xnyp < 1e-10f) costhet = 1.0f;
  else costhet = (float)(-1.0 * v1pj / (double)xnyp);
  // theta = acos(costhet) (reset to 0 at pi), theta += pi / 2, omega = 2 theta, omega = -omega for v1pk < 0
  // (scattering.f90:1206-1217): cos(omega) = cos(2 acos(c) + pi) = 1 - 2 c^2 and sin(omega) = -2 c sqrt(1 - c^2), its
  // sign with v1pk's -- the same two default-real numbers to their last place (1e-7) without acosf, cosf and sinf (150 of
  // this routine's 446 vector instructions); the reset at pi (c = -1) gives cos = -1, sin = 0 either way
  // (formed in double from the default-real costhet -- c^2 and 1 - c^2 are then exact, no cancellation next to c = +-1
  // -- and rounded to default real like the reference's cosw, sinw)
  const double cd = (double)costhet, c2 = cd * cd;
  float sinw = (float)(-2.0 * cd * sqrt(fmax(1.0 - c2, 0.0)));
  if (v1pk < 0.0) sinw = -sinw;
  float cosw = (float)(1.0 - 2.0 * c2);
  if (fabsf(cosw) < 1e-06f) cosw = 0.0f;
  if (fabsf(sinw) < 1e-06f) sinw = 0.0f;
  cw = (double)cosw; sw = (double)sinw;
}
// ... and the Mueller matrix applied between the two rotations, I renormalised (scattering.f90:1228-1298)
__device__ inline void stokes_apply(double S[4], double cw, double sw, double M12, double M22, double M33, double M34,
                                    double M44, double M11 = 1.0) {
  double C0 = S[0], C1 = cw * S[1] - sw * S[2], C2 = sw * S[1] + cw * S[2], C3 = S[3];
  // D = M*C with M = [[M11,M12,0,0],[M12,M22,0,0],[0,0,M33,M34],[0,0,-M34,M44]] (M11 = 1 but for scattering method 1)
  double D0 = M11 * C0 + M12 * C1;
  double D1 = M12 * C0 + M22 * C1;
  double D2 = M33 * C2 + M34 * C3;
  double D3 = -M34 * C2 + M44 * C3;
  double S1_0 = S[0];
  S[0] = D0;
  S[1] = cw * D1 + sw * D2;
  S[2] = -sw * D1 + cw * D2;
  S[3] = D3;
  if (S[0] > TINY_REAL) {
    double f = M11 * S1_0 / S[0];
    S[0] *= f; S[1] *= f; S[2] *= f; S[3] *= f;
  }
}
__device__ inline void update_stokes(double S[4], double u0, double v0, double w0, double u1, double v1,
                                     double w1, double M12, double M22, double M33, double M34,
                                     double M44, double M11 = 1.0) {
  double cw, sw;
  stokes_rotation(u0, v0, w0, u1, v1, w1, cw, sw);
  stokes_apply(S, cw, sw, M12, M22, M33, M34, M44, M11);
}

// intersect_stars (stars.f90:812-884): returns star index (1-based) or 0
__device__ inline int intersect_stars(const DevModel& M, double x, double y, double z, double u,
                                      double v, double w) {
  double d_to_star = HUGE_DP;
  int i_star = 0;
  for (int i = 0; i < M.n_stars; ++i) {
    double dx = x - M.star_xyzr[4 * i + 0], dy = y - M.star_xyzr[4 * i + 1], dz = z - M.star_xyzr[4 * i + 2];
    double r = M.star_xyzr[4 * i + 3];
    double b = dx * u + dy * v + dz * w;
    double c = dx * dx + dy * dy + dz * dz - r * r;
    double delta = b * b - c;
    if (delta >= 0.0) {
      double rac = sqrt(delta);
      double s1 = -b - rac;
      if (s1 < 0) {
        double s2 = -b + rac;
        if (s2 > 0) { d_to_star = 0.0; i_star = i + 1; }
      } else if (s1 < d_to_star) {
        d_to_star = s1; i_star = i + 1;
      }
    }
  }
  return i_star;
}

// The walk's table searches when a whole wave runs ONE packet (the tail kernel, mc_tail.hip.h: every lane holds the same
// state): wl = the lane, and a search is one probe per lane and a ballot instead of a chain of dependent loads -- the
// same "first entry that ..." the bisections return for the monotone tables.  wl < 0: one packet per lane, bisections.
#ifdef MCGPU_LANE_EMULATION
constexpr int WAVE_LANES = 1;    // (tests/emu: a wave of one lane)
#else
constexpr int WAVE_LANES = 64;
#endif
template <typename Pred>
__device__ inline int wave_first(int lo, int hi, int wl, Pred pred) {   // smallest k in [lo, hi) with pred(k), else hi
  for (int base = lo; base < hi; base += WAVE_LANES) {
    const int k = base + wl;
    const bool hit = (k < hi) && pred(k < hi ? k : lo);
    const unsigned long long m = __ballot(hit);
    if (m) return base + (__ffsll((long long)m) - 1);
  }
  return hi;
}

// Temp_LTE (thermal_emission.f90:649-706) -> (Ti, frac); E_scaled already
// includes the replica factor.  First Ti >= 2 with lq(Ti) >= log Qheat,
// equal to the reference's cached linear scan because Qheat only grows.
__device__ inline void temp_lte(const double* lq, int n_T, double E_scaled, double L_packet_th,
                                double volume, int& Ti, double& frac, int wl = -1) {
  double Qheat = E_scaled * L_packet_th / volume;
  frac = 0.0;
  Ti = 2;
  if (Qheat < TINY_DP) return;
  double log_Qheat = log_pos(Qheat);   // (Qheat >= tiny_dp: positive and normal)
  if (log_Qheat < lq[0]) return;
  int lo = 2, hi = n_T;  // 1-based
  if (wl >= 0) lo = hi = wave_first(1, n_T - 1, wl, [&](int k) { return !(lq[k] < log_Qheat); }) + 1;   // (see wave_first)
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (lq[mid - 1] < log_Qheat) lo = mid + 1; else hi = mid;
  }
  Ti = lo;
  frac = (log_Qheat - lq[Ti - 2]) / (lq[Ti - 1] - lq[Ti - 2]);
}

// native FP64 atomic add (global_atomic_add_f64), no CAS loop
__device__ inline void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

// one path-length deposit (save_radiation_field, radiation_field.f90:53)
template <bool LDSE>
__device__ inline void deposit(double* E_glob, double* E_lds, int ic, double v) {
  if (LDSE) atomicAdd(&E_lds[ic], v);  // ds_add_f64
  else atomic_add_f64(&E_glob[ic], v);
}

// the optional accumulators of save_radiation_field's thermal branch (radiation_field.f90:54-55)
__device__ inline void radiation_field_extras(const DevModel& M, const RunArgs& A, int ic, int lambda, double l_S0) {
  if (A.xN_abs) atomicAdd(&A.xN_abs[ic], 1ull);
  if (A.xJ_abs) atomic_add_f64(&A.xJ_abs[(size_t)ic + (size_t)M.n_cells * (size_t)(lambda - 1)], l_S0);
}

// ---------------------------------------------------------------------------
// Deposit cache: 2^log_ns slots of (cell id, partial sum) in LDS, for grids whose absorbed-energy array does not
// fit in LDS (3D cylindrical: 5.76 MB; Voronoi).  The deposits are extremely concentrated (every packet starts in
// the few cells around the star) and same-address global atomics serialise at the memory side (~3.4 ns each,
// chip-wide), so a workgroup sums the hot cells in LDS: slots are claimed first-come with an LDS compare-and-swap
// and then belong to that cell for the launch, hits are ds_add_f64, misses go straight to HBM, and the waves fold
// rotating slices into HBM without a barrier (a slot's owner never changes, so the fold is race-free).
// ---------------------------------------------------------------------------
struct DepCache {
  double* val;
  int* tag;  // 0 = free, else the 1-based cell id that owns the slot for the whole launch
  int log_ns;
  __device__ inline int slot_of(int icell) const { return (int)(((unsigned)icell * 2654435761u) >> (32 - log_ns)); }
  // returns true when the deposit went to the cache
  __device__ inline bool add(int icell, double v) const {
    const int sl = slot_of(icell);
    int t = tag[sl];
    if (t == 0) {
      t = atomicCAS(&tag[sl], 0, icell);
      if (t == 0) t = icell;
    }
    if (t != icell) return false;
    atomic_add_f64(&val[sl], v);
    return true;
  }
  // this workgroup's not yet folded part of a cell's energy
  __device__ inline double pending(int icell) const {
    const int sl = slot_of(icell);
    return tag[sl] == icell ? val[sl] : 0.0;
  }
};

enum : int { S_EMIT = 0, S_INTERACT = 1, S_NEWFLIGHT = 2, S_FLIGHT = 3, S_DONE = 4, S_EXITED = 5, S_KILLED = 6 };
constexpr unsigned long long PK_BATCH = 128;  // packet ids reserved per wave and global atomic

// capteur, SED branch (output.f90:294-397,572-592)
// WEIGHTED: the packet carries a weight in S[0] even when Q,U,V are not tracked (SED mode).
// sed == nullptr: only determine the inclination bin.  Returns capt (0: packet not binned).
template <bool POLA, bool WEIGHTED = false>
__device__ inline int capteur(const DevModel& M, double* sed, int lambda, double u1, double v1,
                              double w1, const double S[4], bool flag_star, bool flag_scatt) {
  double s2 = POLA ? S[2] : 0.0;
  if (w1 < 0.0) {
    if (!M.sym_c) return 0;
    u1 = -u1; v1 = -v1; w1 = -w1;
    s2 = -s2;
  }
  int capt = (int)((-1.0 * w1 + 1.0) * (double)M.N_thet) + 1;
  if (capt == M.N_thet + 1) capt = M.N_thet;
  if (!sed) return capt;
  int c_phi = 1;
  if (M.sym_a) {
    if (v1 < 0.0) { v1 = -v1; s2 = -s2; }
    if (M.N_phi > 1 && w1 != 1.0) c_phi = (int)(atan2(v1, u1) / PI * (double)M.N_phi) + 1;
  } else {
    if (M.N_phi > 1 && w1 != 1.0)
      c_phi = (int)(modulo_d(atan2(u1, v1) + PI / 2, 2 * PI) / (2 * PI) * (double)M.N_phi) + 1;
  }
  if (c_phi == M.N_phi + 1) c_phi = M.N_phi;
  else if (c_phi == 0) c_phi = 1;
  const size_t plane = (size_t)M.n_lambda * M.N_thet * M.N_phi;
  const size_t idx = (size_t)(lambda - 1) + (size_t)M.n_lambda * ((capt - 1) + (size_t)M.N_thet * (c_phi - 1));
  const double I = (POLA || WEIGHTED) ? S[0] : 1.0;
  atomic_add_f64(&sed[0 * plane + idx], I);
  if (POLA) {
    atomic_add_f64(&sed[1 * plane + idx], S[1]);
    atomic_add_f64(&sed[2 * plane + idx], s2);
    atomic_add_f64(&sed[3 * plane + idx], S[3]);
  }
  atomic_add_f64(&sed[4 * plane + idx], 1.0);
  const int tp = flag_star ? (flag_scatt ? 6 : 5) : (flag_scatt ? 8 : 7);
  atomic_add_f64(&sed[tp * plane + idx], I);
  return capt;
}

// ---------------------------------------------------------------------------
// Emission and interaction pieces shared by every grid's packet kernel
// ---------------------------------------------------------------------------
// select_wl_em (thermal_emission.f90:364-400)
__device__ inline int select_wl_em(const Lds& T, const DevModel& M, float rand) {
  int kmin = 0, kmax = M.n_lambda, kk = (kmin + kmax) / 2;
  while (T.cum[kk] != (double)rand) {
    if (T.cum[kk] < (double)rand) kmin = kk; else kmax = kk;
    kk = (kmin + kmax) / 2;
    if ((kmax - kmin) <= 1) break;
  }
  return kmax;
}

// select_star (stars.f90:75-104)
__device__ inline int select_star(const DevModel& M, int lambda, float rand) {
  int kmin = 0, kmax = M.n_stars, kk = (kmax - kmin) / 2;
  while ((kmax - kmin) > 1) {
    if (M.CDF_E_star[(lambda - 1) + (size_t)M.n_lambda * kk] < (double)rand) kmin = kk;
    else kmax = kk;
    kk = (kmin + kmax) / 2;
  }
  return kmax;
}

// select_cellule (thermal_emission.f90:2044-2073); p = prob_E_cell(0:n_cells, lambda)
__device__ inline int select_cellule(const double* p, int n_cells, float rand) {
  int kmin = 0, kmax = n_cells, kk = (kmin + kmax) / 2;
  while ((kmax - kmin) > 1) {
    if (p[kk] < (double)rand) kmin = kk; else kmax = kk;
    kk = (kmin + kmax) / 2;
  }
  return kmax;
}
__device__ inline int select_cellule(const DevModel& M, int lambda, float rand) {
  return select_cellule(M.prob_E_cell + (size_t)(M.n_cells + 1) * (lambda - 1), M.n_cells, rand);
}

// emit_packet_uniform_sphere (stars.f90:108-169), up to the cell lookup
__device__ inline void emit_uniform_sphere(const DevModel& M, int i_star, float r1, float r2, float r3,
                                           float r4, double& x, double& y, double& z, double& u, double& v,
                                           double& w) {
  z = 2.0 * (double)r1 - 1.0;
  const double srw02 = sqrt(1.0 - z * z);
  double sa, ca;
  sincos_pi(2.0 * (double)r2 - 1.0, &sa, &ca);   // (argmt = pi (2 r2 - 1))
  x = srw02 * ca;
  y = srw02 * sa;
  const double cospsi = sqrt((double)r3);
  cdapres_pi(cospsi, 2.0 * (double)r4, x, y, z, u, v, w);   // (phi = 2 pi r4)
  const double* st4 = &M.star_xyzr[4 * (i_star - 1)];
  const double r_star = st4[3] * (1.0 + 1e-6);
  x = x * r_star + st4[0];
  y = y * r_star + st4[1];
  z = z * r_star + st4[2];
}

// random_isotropic_direction (random_numbers.f90:32-51)
__device__ inline void random_isotropic_direction(float r1, float r2, double& u, double& v, double& w) {
  w = 2.0 * (double)r1 - 1.0;
  const double uv = sqrt(1.0 - w * w);
  double sp, cp;
  sincos_pi(2.0 * (double)r2 - 1.0, &sp, &cp);   // (phi = pi (2 r2 - 1))
  u = uv * cp;
  v = uv * sp;
}

// the wavelength of im_reemission_LTE (thermal_emission.f90:740-771) for the temperature (Ti, frac_T2) of Temp_LTE
__device__ inline int reemission_wavelength(const Lds& T, const DevModel& M, int Ti, double frac_T2, float rand2) {
  const double frac_T1 = 1.0 - frac_T2;
  const double* cdf1 = T.cdf + (size_t)M.n_lambda * (Ti - 2);
  const double* cdf2 = T.cdf + (size_t)M.n_lambda * (Ti - 1);
  int l1 = 0, l2 = M.n_lambda, l = (l1 + l2) / 2;
  while ((l2 - l1) > 1) {
    const double proba = frac_T1 * cdf1[l - 1] + frac_T2 * cdf2[l - 1];
    if ((double)rand2 > proba) l1 = l; else l2 = l;
    l = (l1 + l2) / 2;
  }
  return l + 1;
}

// One interaction (dust_transfer.f90:1260-1402): scatter (new direction, Stokes) or absorb +
// immediate re-emission (new wavelength from the cell's temperature, isotropic direction).
// g = the event's draws (see Rng); cell_energy() returns the cell's absorbed energy scaled like
// the reference's partial sum * nb_proc, and is evaluated for absorptions only.
// forced (SED mode, :1263-1278): always a scattering; p_lambda_scatt > 0 overrides the wavelength
// index of the phase-function CDF (the caller's p_lambda).
// The two halves of interact(), so that a kernel can run them as separate phases (mc_roles.hip.h):
// interact_direction: the kind of event, the new wavelength or scattering angle and the new direction
// (returns true for a scattering, with the angle bin `itheta` and the draw `rand2` the Stokes update needs);
// interact_stokes: update_Stokes for a scattering with the interpolated Mueller ratios, Q = U = V = 0 after a
// re-emission.
// select_scattering_grain (dust_prop.f90:1292-1336, low_mem_scattering): walk the CDF of C_sca n over the grain sizes of
// the cell's class from the small grains (rand < 0.5) or from the big ones; k_sca = kappa albedo / (AU_to_cm mum_to_cm^2)
__device__ inline int select_scattering_grain(const DevModel& M, int cls, int lambda, float rand, double norm) {
  const int ng = M.m1_ng;
  if (M.m1_ksca) {
    // select_grainsize_high_mem (dust_prop.f90:1245-1288): the dichotomy in the stored, normalised CDF, statement for
    // statement (prob is the default-real draw; an entry EQUAL to it ends the search with the current kmax)
    const double* cdf = M.m1_ksca + ((size_t)cls * M.n_lambda + (size_t)(lambda - 1)) * (size_t)(ng + 1);
    const double prob = (double)rand;
    int kmin = 0, kmax = ng, k = (kmin + kmax) / 2;
    while (cdf[k] != prob) {
      if (cdf[k] < prob) kmin = k; else kmax = k;
      k = (kmin + kmax) / 2;
      if ((kmax - kmin) <= 1) break;
    }
    return kmax;
  }
  const double* d = M.m1_dens + (size_t)ng * cls;
  const float* Cs = M.m1_Csca + (size_t)ng * (lambda - 1);
  double CDF = 0.0;
  int k;
  if (rand < 0.5f) {
    const double prob = (double)rand * norm;
    for (k = 1; k <= ng; ++k) {
      CDF = CDF + (double)Cs[k - 1] * (d[k - 1] * M.m1_nk[k - 1]);
      if (CDF > prob) break;
    }
    if (k > ng) k = ng;
  } else {
    const double prob = (double)(1.0f - rand) * norm;
    for (k = ng; k >= 1; --k) {
      CDF = CDF + (double)Cs[k - 1] * (d[k - 1] * M.m1_nk[k - 1]);
      if (CDF > prob) break;
    }
    if (k < 1) k = 1;
  }
  return k;
}

template <typename EnergyFn>
__device__ __forceinline__ bool interact_direction(const Lds& T, const DevModel& M, const float g[8], int& lambda,
                                                   double u, double v, double w, double& u1, double& v1, double& w1,
                                                   bool& flag_star, bool& flag_scatt, unsigned int& c_scatt,
                                                   unsigned int& c_abs, EnergyFn cell_energy,
                                                   const double* volume_of_cell, int& itheta, float& rand2_out,
                                                   bool forced = false, const float* prob_forced = nullptr,
                                                   int lds_col = -1, int m1_cls = -1, int* igrain_out = nullptr) {
  const bool scat = forced || (g[0] < T.albedo[lambda - 1]);  // dust_transfer.f90:1284
  // scattering method 1 (m1_cls >= 0: the cell's class): the draws are grain, angle, angle, azimuth (dust_transfer.f90:1289-1297)
  const bool m1 = M.m1 != 0 && m1_cls >= 0;
  const float rand = m1 ? g[2] : g[1], rand2 = m1 ? g[3] : g[2], rand_phi = m1 ? g[4] : g[3];
  itheta = 1;
  rand2_out = rand2;
  double cospsi, phi;   // (phi in units of pi)
  if (scat) {
    flag_scatt = true;
    c_scatt++;
    int igrain = 0;
    if (m1) {
      igrain = select_scattering_grain(M, m1_cls, lambda, g[1], T.kappa[lambda - 1] * (double)T.albedo[lambda - 1] / (149597870700.0 * 100.0 * (1.0e-4 * 1.0e-4)));
      if (igrain_out) *igrain_out = igrain;
    }
    if (m1 && M.aniso_method == 1) {
      // angle_diff_theta (scattering.f90:1387-1429) in the grain's own cumulative phase function
      const float* prob = M.m1_prob + ((size_t)(lambda - 1) * M.m1_ng + (igrain - 1)) * (size_t)(M.nang + 1);
      int kmin = 0, kmax = M.nang, kk = (kmin + kmax) / 2;
      while ((kmax - kmin) > 1) {
        if (prob[kk] < rand) kmin = kk; else kmax = kk;
        kk = (kmin + kmax) / 2;
      }
      itheta = kmax;
      const double c0 = T.cost[itheta - 1], c1 = T.cost[itheta];
      cospsi = c0 + (double)rand2 * (c1 - c0);
    } else if (M.aniso_method == 1) {
      // angle_diff_theta_pos (scattering.f90:1433-1475)
      // (lds_col >= 0: that column of T.prob -- the SED-mode tables hold only column p_lambda, as column 0)
      const size_t col = lds_col >= 0 ? (size_t)lds_col : (M.p_lambda_fixed ? (size_t)0 : (size_t)(lambda - 1));
      const float* prob = prob_forced ? prob_forced : T.prob + (size_t)(M.nang + 1) * col;
      int kmin = 0, kmax = M.nang, kk = (kmin + kmax) / 2;
      while ((kmax - kmin) > 1) {
        if (prob[kk] < rand) kmin = kk; else kmax = kk;
        kk = (kmin + kmax) / 2;
      }
      itheta = kmax;
      const double c0 = T.cost[itheta - 1], c1 = T.cost[itheta];
      cospsi = c0 + (double)rand2 * (c1 - c0);
    } else {
      // hg (scattering.f90:1354-1383); method 1: with the grain's asymmetry parameter (dust_transfer.f90:1307)
      const float gg = m1 ? M.m1_g[(size_t)(igrain - 1) + (size_t)M.m1_ng * (lambda - 1)] : T.g[lambda - 1];
      const double rand_dp = fmin((double)rand, 1.0 - 1e-6);
      if (fabsf(gg) > 1.17549435e-38f) {
        const double g1 = (double)gg, g2 = g1 * g1;
        const double q = (1.0 - g2) / (1.0 - g1 + 2.0 * g1 * rand_dp);
        cospsi = (1.0 + g2 - q * q) / (2.0 * g1);
      } else {
        cospsi = 2.0 * rand_dp - 1.0;
      }
      itheta = (int)floor(acos(cospsi) * 180.0 / PI) + 1;
      if (itheta > M.nang) itheta = M.nang;
    }
    if (M.lisotropic && !(m1 && M.aniso_method == 1)) { itheta = 1; cospsi = 2.0 * (double)rand - 1.0; }
    phi = 2.0 * (double)rand_phi - 1.0;
  } else {
    c_abs++;
    flag_star = false;
    flag_scatt = false;
    // im_reemission_LTE (thermal_emission.f90:710-771)
    const double E = cell_energy();
    int Ti;
    double frac_T2;
    temp_lte(T.lq, M.n_T, E, M.L_packet_th, *volume_of_cell, Ti, frac_T2);
    lambda = reemission_wavelength(T, M, Ti, frac_T2, g[2]);
    // random_isotropic_direction (random_numbers.f90:32-51): w = 2r-1, (u,v) = sqrt(1-w^2)
    // (cos,sin)(phi) is cdapres' own |w0| > 0.999999 branch applied to the z axis
    cospsi = 2.0 * (double)g[3] - 1.0;
    phi = 2.0 * (double)g[4] - 1.0;
  }
  // new direction: one instruction stream for both kinds of event
  cdapres_pi(cospsi, phi, scat ? u : 0.0, scat ? v : 0.0, scat ? w : 1.0, u1, v1, w1);
  return scat;
}

// get_Mueller_matrix_per_cell (scattering.f90:1328-1350): the ratios S12/S11 ... interpolated between the angle bins
__device__ __forceinline__ void mueller_pos(const DevModel& M, int lambda, int itheta, float rand2, int cls, double& M12,
                                            double& M22, double& M33, double& M34, double& M44) {
  const size_t o = (size_t)(M.nang + 1) * (lambda - 1) + itheta;
  const float fr = rand2, fm = 1.0f - rand2;
  const size_t co = cls >= 0 ? (size_t)cls * M.n_lambda * (M.nang + 1) : 0;
  const float* s22 = (cls >= 0 ? M.v_s22 : M.s22) + co;
  const float* s12 = (cls >= 0 ? M.v_s12 : M.s12) + co;
  const float* s33 = (cls >= 0 ? M.v_s33 : M.s33) + co;
  const float* s44 = (cls >= 0 ? M.v_s44 : M.s44) + co;
  const float* s34 = (cls >= 0 ? M.v_s34 : M.s34) + co;
  // (default-real products and sums, unfused like the reference's: a contracted multiply-add differs in the last place of
  // default real, and which copy of this code contracts is the compiler's choice per kernel -- Q, U, V of a packet must not
  // depend on the kernel that finishes it)
  M22 = (double)nf_add(nf_mul(s22[o], fr), nf_mul(s22[o - 1], fm));
  M12 = (double)nf_add(nf_mul(s12[o], fr), nf_mul(s12[o - 1], fm));
  M33 = (double)nf_add(nf_mul(s33[o], fr), nf_mul(s33[o - 1], fm));
  M44 = (double)nf_add(nf_mul(s44[o], fr), nf_mul(s44[o - 1], fm));
  M34 = (double)nf_sub(nf_mul(-s34[o], fr), nf_mul(s34[o - 1], fm));
}

// lambda: the packet's wavelength at the time of the scattering (a scattering does not change it)
// cls >= 0: the Mueller ratios of that cell class (lvariable_dust with per-class scattering tables)
__device__ __forceinline__ void interact_stokes(const DevModel& M, bool scat, int lambda, int itheta, float rand2,
                                                double u, double v, double w, double u1, double v1, double w1,
                                                double S[4], int cls = -1, int igrain = 0) {
  if (scat && M.aniso_method == 1 && igrain > 0) {
    // get_Mueller_matrix_per_grain (scattering.f90:1302-1324): the grain's own matrix, s11 included
    const size_t o = (size_t)(M.nang + 1) * ((size_t)(igrain - 1) + (size_t)M.m1_ng * (lambda - 1)) + itheta;
    const float fr = rand2, fm = 1.0f - rand2;
    const double M11 = (double)(M.m1_s11[o] * fr + M.m1_s11[o - 1] * fm);
    const double M22 = (double)(M.m1_s22[o] * fr + M.m1_s22[o - 1] * fm);
    const double M12 = (double)(M.m1_s12[o] * fr + M.m1_s12[o - 1] * fm);
    const double M33 = (double)(M.m1_s33[o] * fr + M.m1_s33[o - 1] * fm);
    const double M44 = (double)(M.m1_s44[o] * fr + M.m1_s44[o - 1] * fm);
    const double M34 = (double)(-M.m1_s34[o] * fr - M.m1_s34[o - 1] * fm);
    update_stokes(S, u, v, w, u1, v1, w1, M12, M22, M33, M34, M44, M11);
  } else if (scat && M.aniso_method == 1) {
    double M12, M22, M33, M34, M44;
    mueller_pos(M, lambda, itheta, rand2, cls, M12, M22, M33, M34, M44);
    update_stokes(S, u, v, w, u1, v1, w1, M12, M22, M33, M34, M44);
  }
  if (!scat) { S[1] = 0.0; S[2] = 0.0; S[3] = 0.0; }
}

template <bool POLA, typename EnergyFn>
__device__ __forceinline__ void interact(const Lds& T, const DevModel& M, const float g[8], int& lambda,
                                         double u, double v, double w, double& u1, double& v1, double& w1,
                                         double S[4], bool& flag_star, bool& flag_scatt,
                                         unsigned int& c_scatt, unsigned int& c_abs, EnergyFn cell_energy,
                                         const double* volume_of_cell, bool forced = false,
                                         const float* prob_forced = nullptr, int lds_col = -1, int mueller_class = -1,
                                         int m1_cls = -1) {
  int itheta, igrain = 0;
  float rand2;
  const int lambda_in = lambda;
  const bool scat = interact_direction(T, M, g, lambda, u, v, w, u1, v1, w1, flag_star, flag_scatt, c_scatt, c_abs,
                                       cell_energy, volume_of_cell, itheta, rand2, forced, prob_forced, lds_col, m1_cls, &igrain);
  if (POLA) interact_stokes(M, scat, lambda_in, itheta, rand2, u, v, w, u1, v1, w1, S, mueller_class, igrain);
}

// ---------------------------------------------------------------------------
// Modified random walk (Min et al. 2009; Robitaille 2010) for the 2D cylindrical grid.  The reference carries the
// pieces (MRW.f90: zeta table :16-53, gamma_MRW :11, cst_ct :12, the step :74-115; distance_to_closest_wall_cyl,
// cylindrical_grid.f90:1179; the trigger and the loop, dust_transfer.f90:1222-1239) but its step is an unfinished
// stub behind a commented-out call; this is the working algorithm, described in DESIGN.md (section "Modified random walk") and
// validated against the brute-force loop.  The walk's draws come from the packet's own counter
// sub-space (block k of event e: Philox counter (k, e, packet id)), so the result does not depend on the schedule.
// ---------------------------------------------------------------------------
// distance_to_closest_wall_cyl (cylindrical_grid.f90:1179-1226), 2D
__device__ inline double distance_to_closest_wall_cyl(const Lds& T, const DevModel& M, int ri, int zj, double x,
                                                      double y, double z, int kaz = 1) {
  if (M.grid_sph) {
    // distance_to_closest_wall_sph (spherical_grid.f90:451-499): the shells and the cones of the polar walls --
    // |rcyl sin(a) - z0 cos(a)| with tan a = tan_theta_lim; the reference's sketch multiplies z0 with cos_phi_lim, the
    // azimuthal walls' table, where this cosine is meant -- and the azimuthal walls of a 3D grid
    const double r2c = x * x + y * y, rcyl = sqrt(r2c), rr = sqrt(r2c + z * z), z0 = fabs(z);
    const int tj = zj < 0 ? -zj : zj;
    double s = fmin(M.r_lim[ri] - rr, rr - M.r_lim[ri - 1]);
    for (int j = tj - 1; j <= tj; ++j) {
      const double t = M.tan_theta_lim[j];
      const double c = 1.0 / sqrt(1.0 + t * t), sn = t * c;
      s = fmin(s, fabs(rcyl * sn - z0 * c));
    }
    if (M.l3D && M.n_az > 1) {
      const int km = kaz > 1 ? kaz - 1 : M.n_az;
      s = fmin(s, fmin(fabs(x * M.sin_phi[kaz - 1] - y * M.cos_phi[kaz - 1]), fabs(x * M.sin_phi[km - 1] - y * M.cos_phi[km - 1])));
    }
    return s;
  }
  const double r = sqrt(x * x + y * y);
  const double s1 = M.r_lim[ri] - r, s2 = r - M.r_lim[ri - 1];
  const double z0 = fabs(z);
  const int azj = zj < 0 ? -zj : zj;
  const double s3 = z_lim_of(T, M.nz, ri, azj + 1) - z0, s4 = z0 - z_lim_of(T, M.nz, ri, azj);
  double s = fmin(fmin(s1, s2), fmin(s3, s4));
  if (M.l3D && M.n_az > 1) {
    // the azimuthal walls (:1198-1218): |x sin(phi) - y cos(phi)| for the walls k and k - 1 (wall 0 = wall n_az: the
    // reference indexes sin_phi_lim(0), out of bounds; see mcgpu_set_mrw for the tables)
    const int km = kaz > 1 ? kaz - 1 : M.n_az;
    const double s5 = fabs(x * M.sin_phi[kaz - 1] - y * M.cos_phi[kaz - 1]);
    const double s6 = fabs(x * M.sin_phi[km - 1] - y * M.cos_phi[km - 1]);
    s = fmin(s, fmin(s5, s6));
  }
  return s;
}

// y with zeta(y) = xi (MRW.f90:58-70 read as the inverse it means)
constexpr int MRW_GUIDE = 1024;
__device__ inline double mrw_sample_y(const DevModel& M, float xi, int wl = -1) {
  const double* zt = M.mrw_zeta;
  const double x = xi > 0.0f ? (double)xi : 2.9802322387695312e-08;
  // zeta[lo] <= x < zeta[hi], hi - lo = 1: unique in a non-decreasing table, so the guide (the answers at the bucket
  // edges) only narrows the bracket the bisection starts from
  int lo = 0, hi = M.mrw_n_zeta - 1;
  if (M.mrw_guide) {
    const int b = (int)(x * (double)MRW_GUIDE);   // x < 1
    lo = M.mrw_guide[b];
    const int h2 = M.mrw_guide[b + 1] + 1;
    hi = h2 < hi ? h2 : hi;
  }
  if (wl >= 0) {   // the first entry above x in (lo, hi] (zt[hi] > x): hi; lo is the one before
    hi = wave_first(lo + 1, hi, wl, [&](int k) { return !(zt[k] <= x); });
    lo = hi - 1;
  }
  while (hi - lo > 1) {
    const int mid = (lo + hi) / 2;
    if (zt[mid] <= x) lo = mid; else hi = mid;
  }
  const double f = (x - zt[lo]) / (zt[hi] - zt[lo]);
  return ((double)lo + f) / (double)(M.mrw_n_zeta - 1);
}

// One walk of a packet the cell (ri, zj; index ic) has just re-emitted.  cell_energy() as in interact();
// add_energy(v) deposits v into the cell.  Returns false (nothing drawn, nothing changed) when the sphere is
// not optically thick enough.
// (mrw_walk_with: the walk with the cell's closest-wall distance and kappa_factor supplied by the caller -- the grids differ
// only there; mrw_walk: the structured grids)
template <typename DistFn, typename EnergyFn, typename DepositFn>
__device__ inline bool mrw_walk_with(const Lds& T, const DevModel& M, uint32_t k0, uint32_t k1, uint32_t p_lo, uint32_t p_hi,
                                     uint32_t event, int ic, double kf, double S0, double& x, double& y, double& z,
                                     double& u, double& v, double& w, int& lambda, DistFn closest_wall, EnergyFn cell_energy,
                                     DepositFn add_energy, unsigned int& c_walks, unsigned int& c_steps, int wl = -1,
                                     int Ti0 = 0, double frac0 = 0.0) {
  double d = closest_wall(x, y, z);
  int Ti = Ti0;
  double frac = frac0;
  // (Ti0 > 0: the caller has the cell's temperature bracket from the absorption that precedes the walk -- same cell, same
  // energy, same expression: k_tail hands it over instead of paying a logarithm and a search twice per event)
  if (Ti0 <= 0) temp_lte(T.lq, M.n_T, cell_energy(), M.L_packet_th, M.volume[ic], Ti, frac, wl);
  // (lvariable_dust: the mean opacities of the cell's class, [n_classes][n_T]; T then holds the class's lq / cdf)
  const size_t co = M.n_classes ? (size_t)M.cell_class[ic] * M.n_T : 0;
  const double *t_chi = M.mrw_chi + co, *t_kdep = M.mrw_kdep + co, *t_ext = M.mrw_ext + co;
  const double chi = (t_chi[Ti - 2] * (1.0 - frac) + t_chi[Ti - 1] * frac) * kf;
  if (!(d * chi > (double)M.mrw_gamma)) return false;
  const double kdep = t_kdep[Ti - 2] * (1.0 - frac) + t_kdep[Ti - 1] * frac;
  const double ext = (t_ext[Ti - 2] * (1.0 - frac) + t_ext[Ti - 1] * frac) / kf;
  const double cst_ct = 3.0 / (PI * PI);
  double su, sv, sw;
  uint32_t blk = 0, o[4];
  do {
    philox4x32_10(blk++, event, p_lo, p_hi, k0, k1, o);
    random_isotropic_direction(Rng::real(o[0]), Rng::real(o[1]), su, sv, sw);
    x += su * d;
    y += sv * d;
    z += sw * d;
    const double yv = mrw_sample_y(M, Rng::real(o[2]), wl);
    const double de = d + ext;
    const double ct = -log_pos(yv) * cst_ct * chi * (de * de);
    add_energy(kdep * ct * S0);
    c_steps++;
    d = closest_wall(x, y, z);
  } while (d * chi > (double)M.mrw_gamma);
  philox4x32_10(blk, event, p_lo, p_hi, k0, k1, o);
  // the cell's temperature now, the walk's deposits included (im_reemission_LTE)
  temp_lte(T.lq, M.n_T, cell_energy(), M.L_packet_th, M.volume[ic], Ti, frac, wl);
  if (M.mrw_exit_cdf) {
    // the packet crosses the sphere in the middle of a flight: it carries the spectrum of the packets IN FLIGHT in the
    // thick cell (weights dB/dT), not that of a packet which has just been emitted (kappa_abs dB/dT) -- see
    // mcgpu_set_mrw_exit_spectrum (include/mcgpu.h); the same search as im_reemission_LTE's, in the table in HBM
    const double* c1 = M.mrw_exit_cdf + (co + (size_t)(Ti - 2)) * M.n_lambda;
    const double* c2 = c1 + M.n_lambda;
    const double f1 = 1.0 - frac, r = (double)Rng::real(o[0]);
    int l1 = 0, l2 = M.n_lambda, l = (l1 + l2) / 2;
    if (wl >= 0) {   // the first l in [1, n_lambda) whose interpolated cumulative spectrum reaches the draw, else n_lambda
      l2 = wave_first(1, M.n_lambda, wl, [&](int k) { return !(r > f1 * c1[k - 1] + frac * c2[k - 1]); });
      l1 = l2 - 1; l = l1;
    }
    while ((l2 - l1) > 1) {
      const double proba = f1 * c1[l - 1] + frac * c2[l - 1];
      if (r > proba) l1 = l; else l2 = l;
      l = (l1 + l2) / 2;
    }
    lambda = l + 1;
  } else lambda = reemission_wavelength(T, M, Ti, frac, Rng::real(o[0]));
  cdapres_pi(sqrt((double)Rng::real(o[1])), 2.0 * (double)Rng::real(o[2]) - 1.0, su, sv, sw, u, v, w);
  c_walks++;
  return true;
}

template <typename EnergyFn, typename DepositFn>
__device__ inline bool mrw_walk(const Lds& T, const DevModel& M, uint32_t k0, uint32_t k1, uint32_t p_lo, uint32_t p_hi,
                                uint32_t event, int ri, int zj, int ic, double S0, double& x, double& y, double& z,
                                double& u, double& v, double& w, int& lambda, EnergyFn cell_energy,
                                DepositFn add_energy, unsigned int& c_walks, unsigned int& c_steps, int kaz = 1, int wl = -1,
                                int Ti0 = 0, double frac0 = 0.0) {
  return mrw_walk_with(T, M, k0, k1, p_lo, p_hi, event, ic, M.kappa_factor[ic], S0, x, y, z, u, v, w, lambda,
                       [&](double px, double py, double pz) { return distance_to_closest_wall_cyl(T, M, ri, zj, px, py, pz, kaz); },
                       cell_energy, add_energy, c_walks, c_steps, wl, Ti0, frac0);
}

// ---------------------------------------------------------------------------
// emit_packet (dust_transfer.f90:1047-1151) for any grid.  f = the emission event's draws (see
// Rng): f1 chooses star / disk / ISM against frac_E_stars, frac_E_disk; star: f2 select_star,
// f3..f6 emit_packet_uniform_sphere; disk: f2 select_cellule, f3..f5 pos_em_cell, f6,f7 direction;
// ISM: f2..f5 emit_packet_ISM (stars.f90:728-785).  Ops supplies the grid operators and keeps the
// packet's cell: star_cell(i_star, x,y,z), bool enter_grid(x,y,z,u,v,w), disk_cell(icell, r1,r2,r3,
// x,y,z).  Returns 0, or the error code of a source the tables do not provide.
// ---------------------------------------------------------------------------
template <class Ops>
__device__ inline int emit_packet(const DevModel& M, const float f[12], int lambda, double frac_E_stars,
                                  double frac_E_disk, const double* prob_E_cell, Ops& ops, double& x, double& y,
                                  double& z, double& u, double& v, double& w, bool& flag_star, bool& flag_ism,
                                  bool& lintersect) {
  lintersect = true;
  flag_ism = false;
  if ((double)f[1] <= frac_E_stars) {
    flag_star = true;
    const int i_star = select_star(M, lambda, f[2]);
    emit_uniform_sphere(M, i_star, f[3], f[4], f[5], f[6], x, y, z, u, v, w);
    ops.star_cell(i_star, x, y, z);
    if (M.star_cell[4 * (i_star - 1) + 3]) lintersect = ops.enter_grid(x, y, z, u, v, w);
    return 0;
  }
  flag_star = false;
  if ((double)f[1] <= frac_E_disk) {
    if (!prob_E_cell) return 12;
    const int icell = select_cellule(prob_E_cell, M.n_cells, f[2]);
    ops.disk_cell(icell, f[3], f[4], f[5], x, y, z);
    random_isotropic_direction(f[6], f[7], u, v, w);
    return 0;
  }
  if (!(M.R_ISM > 0.0)) return 12;
  flag_ism = true;  // emit_packet_ISM: a point of the sphere, cosine law towards the interior
  z = 2.0 * (double)f[2] - 1.0;
  const double srw02 = sqrt(1.0 - z * z);
  double sa, ca;
  sincos_pi(2.0 * (double)f[3] - 1.0, &sa, &ca);
  x = srw02 * ca;
  y = srw02 * sa;
  const double cospsi = -sqrt((double)f[4]);
  cdapres_pi(cospsi, 2.0 * (double)f[5], x, y, z, u, v, w);
  x = M.centre_ISM[0] + x * M.R_ISM;
  y = M.centre_ISM[1] + y * M.R_ISM;
  z = M.centre_ISM[2] + z * M.R_ISM;
  lintersect = ops.enter_grid(x, y, z, u, v, w);
  return 0;
}

// the cylindrical grid's operators for emit_packet; the packet's cell lives in (ri, zj, k)
template <bool L3D>
struct CylEmitOps {
  const Lds& T;
  const DevModel& M;
  int &ri, &zj, &k;
  __device__ inline void star_cell(int, double x, double y, double z) { index_cell<L3D>(T, M, x, y, z, ri, zj, k); }
  __device__ inline bool enter_grid(double& x, double& y, double& z, double u, double v, double w) {
    return move_to_grid<L3D>(T, M, x, y, z, u, v, w, ri, zj, k);
  }
  __device__ inline void disk_cell(int icell, float r1, float r2, float r3, double& x, double& y, double& z) {
    int q = icell - 1;  // inverse of the closed-form mapping
    ri = q % M.n_rad + 1;
    q /= M.n_rad;
    if (L3D) {
      const int jj = q % (2 * M.nz);
      k = q / (2 * M.nz) + 1;
      zj = jj < M.nz ? jj - M.nz : jj - M.nz + 1;
    } else {
      zj = q + 1;
      k = 1;
    }
    pos_em_cell<L3D>(T, M, ri, zj, k, r1, r2, r3, x, y, z);
  }
};

// the spherical grid's operators for emit_packet
template <bool L3D>
struct SphEmitOps {
  const Lds& T;
  const DevModel& M;
  int &ri, &zj, &k;
  __device__ inline void star_cell(int, double x, double y, double z) { index_cell_sph<L3D>(T, M, x, y, z, ri, zj, k); }
  __device__ inline bool enter_grid(double& x, double& y, double& z, double u, double v, double w) {
    return move_to_grid_sph<L3D>(T, M, x, y, z, u, v, w, ri, zj, k);
  }
  __device__ inline void disk_cell(int icell, float r1, float r2, float r3, double& x, double& y, double& z) {
    int q = icell - 1;  // inverse of the closed-form mapping
    ri = q % M.n_rad + 1;
    q /= M.n_rad;
    if (L3D) {
      const int jj = q % (2 * M.nz);
      k = q / (2 * M.nz) + 1;
      zj = jj < M.nz ? jj - M.nz : jj - M.nz + 1;
    } else {
      zj = q + 1;
      k = 1;
    }
    pos_em_cell_sph<L3D>(M, ri, zj, k, r1, r2, r3, x, y, z);
  }
};

// ---------------------------------------------------------------------------
// The thermal packet kernel
// ---------------------------------------------------------------------------
// LDSE: the absorbed-energy grid of this workgroup lives in LDS (2D grids:
// n_cells * 8 B + tables fit the 160 KB of a CU); deposits are LDS atomics
// (ds_add_f64); the waves fold rotating slices of the private grid into HBM
// every A.flush_every outer iterations, without a workgroup barrier.  Otherwise deposits go straight to HBM
// (global_atomic_add_f64), the only option for 3D grids (5.76 MB at 720 000
// cells).
// SPH: the grid operators of spherical_grid.f90 instead of cylindrical_grid.f90 (same cell identity and mapping).
template <bool L3D, bool POLA, bool DARK, bool LDSE, bool SPH = false, bool MRW = false, bool VAR = false>
__device__ __forceinline__ void thermal_body(const DevModel& M, const RunArgs& A, double* lds_base) {
  double* const E_lds = lds_base;  // [n_cells] when LDSE
  const Lds T = lds_carve(lds_base + (LDSE ? M.n_cells : 0), M);
  lds_stage(T, M);
  if (LDSE)
    for (int i = threadIdx.x; i < M.n_cells; i += blockDim.x) E_lds[i] = 0.0;
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int n_rad = M.n_rad, nz = M.nz;

  // packet state
  int st = S_EMIT;
  double x = 0, y = 0, z = 0, u = 0, v = 0, w = 1, extr = 0, inv_a = 0, inv_w = 0;
  double xo = 0, yo = 0, zo = 0;  // entry point of the previous cell (dark-zone mirror)
  int ri = 0, zj = 1, k = 1, ri_o = 0, zj_o = 1, k_o = 1;
  int lambda = 1;
  int star_key = -1;  // packed (ri,zj,k) of the star the flight would hit
  bool flag_star = false, flag_scatt = false, flag_ism = false;
  double S[4] = {1.0, 0.0, 0.0, 0.0};
  Rng rng;
  rng.init(0, 0);
  unsigned int c_cross = 0, c_flight = 0, c_scatt = 0, c_abs = 0, c_esc = 0, c_kill = 0, c_dark = 0,
               c_pack = 0;
  unsigned int pk_cross = 0;  // crossings of the current packet (runaway guard)
  int n_inter = 0;            // MRW: interactions in a row whose flights never left the cell (0..7)
  bool first_cross = true;    // MRW: the flight is still in the cell it started in
  unsigned int c_walks = 0, c_steps = 0;
  unsigned long long pk_next = 0, pk_end = 0;  // this wave's reserved packet ids (wave-uniform)
  float tau_rand = 0.0f;  // the current event's draw for the next flight's optical depth
  double kf = 0.0;  // kappa_factor of the current cell (0 in virtual cells), fetched one crossing ahead

#ifdef MCGPU_PHASE_TIMING
  unsigned long long tp_emit = 0, tp_int = 0, tp_new = 0, tp_fly = 0, tp0;
#define TP_START() tp0 = clock64()
#define TP_ADD(acc) do { unsigned long long t1_ = clock64(); acc += t1_ - tp0; tp0 = t1_; } while (0)
#else
#define TP_START()
#define TP_ADD(acc)
#endif
  for (int ep = 0;; ++ep) {  // outer iterations
    TP_START();
    // ---- EXITED: bin the packets that left the grid (capteur) --------------
    if (st == S_EXITED) {
      if (!flag_ism) {  // ISM packets that were never absorbed are not binned (dust_transfer.f90:549)
        capteur<POLA>(M, A.sed, lambda, u, v, w, S, flag_star, flag_scatt);
        c_esc++;
      }
      st = S_EMIT;
    }

    // ---- EMIT: hand out packet ids from this wave's reserved batch; one global atomic per
    // PK_BATCH packets and wave (the ids, hence the random streams, are independent of who
    // runs them) ---------------------------------------------------------------------------
    {
      const bool need = (st == S_EMIT);
      const unsigned long long mask = __ballot(need);
      if (mask) {
        if (pk_next >= pk_end) {  // wave-uniform
          const int leader = __ffsll((long long)mask) - 1;
          unsigned long long base = 0;
          if (lane == leader) base = atomicAdd(A.next_packet, (unsigned long long)PK_BATCH);
          base = __shfl(base, leader);
          pk_next = base < A.n_packets ? base : A.n_packets;
          pk_end = (base + PK_BATCH < A.n_packets) ? base + PK_BATCH : A.n_packets;
          if (pk_end < pk_next) pk_end = pk_next;
        }
        const unsigned long long avail = pk_end - pk_next;
        const unsigned long long rank = (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
        const unsigned long long cnt = (unsigned long long)__popcll(mask);
        const unsigned long long my = pk_next + rank;
        const bool served = need && (rank < avail);
        if (need && !served && pk_next >= A.n_packets) st = S_DONE;  // nothing left anywhere
        pk_next += (cnt < avail) ? cnt : avail;
        if (served) {
          {
            // mc_photon_loop body (dust_transfer.f90:529-541)
            rng.init(A.seed, A.first_packet + my);
            c_pack++;
            pk_cross = 0;
            n_inter = 0;
            float f[12];
            rng.emission_event(f);
            tau_rand = f[8];
            float rand = f[0];
            lambda = select_wl_em(T, M, rand);
            lds_count_sent(T, lambda);
            bool lintersect;
            flag_scatt = false;
            S[0] = 1.0; S[1] = 0.0; S[2] = 0.0; S[3] = 0.0;
            int rc;
            if (SPH) {
              SphEmitOps<L3D> ops{T, M, ri, zj, k};
              rc = emit_packet(M, f, lambda, T.fstar[lambda - 1], M.frac_E_disk[lambda - 1],
                               M.prob_E_cell ? M.prob_E_cell + (size_t)(M.n_cells + 1) * (lambda - 1) : nullptr,
                               ops, x, y, z, u, v, w, flag_star, flag_ism, lintersect);
            } else {
              CylEmitOps<L3D> ops{T, M, ri, zj, k};
              rc = emit_packet(M, f, lambda, T.fstar[lambda - 1], M.frac_E_disk[lambda - 1],
                               M.prob_E_cell ? M.prob_E_cell + (size_t)(M.n_cells + 1) * (lambda - 1) : nullptr,
                               ops, x, y, z, u, v, w, flag_star, flag_ism, lintersect);
            }
            if (rc) {  // a source the tables do not provide
              *A.err = rc;
              st = S_DONE;
            }
            if (st != S_DONE) st = lintersect ? S_NEWFLIGHT : S_EXITED;  // never entered the grid:
                                                                         // binned directly (:549-550)
          }
        }
      }
    }

    TP_ADD(tp_emit);
    // ---- INTERACT: scatter or absorb + re-emit (dust_transfer.f90:1260-1402)
    if (st == S_INTERACT) {
      float g[8];
      rng.interaction_event(g, M.m1 != 0);
      tau_rand = g[5];
      double u1, v1, w1;
      const int ic = cell_index<L3D>(n_rad, nz, ri, zj, k);
      const Lds Tc = VAR ? class_tables(T, M, M.cell_class[ic]) : T;   // (lvariable_dust: this cell's tables)
      interact<POLA>(Tc, M, g, lambda, u, v, w, u1, v1, w1, S, flag_star, flag_scatt, c_scatt, c_abs, [&]() {
        // the cell's absorbed energy for Temp_LTE (thermal_emission.f90:649-706)
        double E;
        if (A.frozen) E = A.E_prior[ic];
        else {
          // running absorbed energy of the cell: what every workgroup has folded into HBM so
          // far, plus (LDSE) this workgroup's not yet folded part; * n_replicas like the
          // reference's `* nb_proc` (thermal_emission.f90:670)
          E = __hip_atomic_load(&A.E_abs[ic], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // the other workgroups' unfolded parts are estimated by this one's, exactly the
          // reference's partial * nb_proc with workgroups in the role of threads
          if (LDSE) E += E_lds[ic] * (double)gridDim.x;
          E *= A.qscale;
        }
        return E;
      }, M.volume + ic, false, nullptr, -1, (VAR && M.v_scatt) ? M.cell_class[ic] : -1, (VAR && M.m1) ? M.cell_class[ic] : -1);
      if (!flag_scatt) flag_ism = false;  // absorbed and re-emitted by the dust (:1367)
      u = u1; v = v1; w = w1;
      if (MRW) {
        // modified random walk of a packet the cell has just re-emitted for the (n_inter+1)-th time in a row
        // (dust_transfer.f90:1222-1239; mrw_walk above)
        if (__builtin_expect(!flag_scatt && !flag_star && n_inter > M.mrw_n_inter, 0)) {
          mrw_walk(Tc, M, rng.k0, rng.k1, rng.p_lo, rng.p_hi, rng.event, ri, zj, ic, S[0], x, y, z, u, v, w, lambda,
                   [&]() {
                     double E;
                     if (A.frozen) E = A.E_prior[ic];
                     else {
                       E = __hip_atomic_load(&A.E_abs[ic], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                       if (LDSE) E += E_lds[ic] * (double)gridDim.x;
                       E *= A.qscale;
                     }
                     return E;
                   },
                   [&](double e) { deposit<LDSE>(A.E_abs, E_lds, ic, e); }, c_walks, c_steps, k);
        }
      }
      st = S_NEWFLIGHT;
    }

    TP_ADD(tp_int);
    // ---- NEWFLIGHT: optical depth to the next event + per-flight constants
    if (st == S_NEWFLIGHT) {
      const float rand = tau_rand;  // dust_transfer.f90:1208-1215 (tau in FP64)
      extr = tau_of_draw(rand);
      const double a = u * u + v * v;  // cylindrical_grid.f90:941-952
      inv_a = (a > TINY_REAL) ? 1.0 / a : HUGE_REAL;
      inv_w = (fabs(w) > TINY_REAL) ? 1.0 / w : copysign(HUGE_DP, w);
      const int i_star = intersect_stars(M, x, y, z, u, v, w);  // optical_depth.f90:68
      star_key = -1;
      if (i_star > 0) {
        const int* sc = &M.star_cell[4 * (i_star - 1)];
        star_key = sc[0] + (n_rad + 2) * ((sc[1] + nz + 1) + (2 * nz + 3) * (sc[2] - 1));
      }
      c_flight++;
      ri_o = 0; zj_o = 0; k_o = 0;
      xo = x; yo = y; zo = z;
      first_cross = true;
      kf = is_real_cell<L3D>(n_rad, nz, ri, zj) ? M.kappa_factor[cell_index<L3D>(n_rad, nz, ri, zj, k)] : 0.0;
      st = S_FLIGHT;
    }

    TP_ADD(tp_new);
    if (__ballot(st != S_DONE) == 0ull) break;

    // ---- FLIGHT: cell crossings (physical_length, optical_depth.f90:77-178)
#pragma unroll 1
    for (int it = 0; it < A.inner_iters; ++it) {
      if (A.min_active > 0 && it > 0) {
        // leave early once fewer than min_active/64 of the lanes that still own a packet fly
        const int flying = __popcll(__ballot(st == S_FLIGHT)), alive = __popcll(__ballot(st != S_DONE));
        if (flying * 64 < A.min_active * alive) break;
      }
#ifdef MCGPU_COUNT_ITERS  // diagnostic build (tools/loop_utilisation.py): wave iterations of this loop
      if (lane == 0) c_dark++;
#endif
      if (st == S_FLIGHT) {
        const int azj = zj < 0 ? -zj : zj;
        // test_exit_grid_cyl (cylindrical_grid.f90:680-704) / test_exit_grid_sph (spherical_grid.f90:24-44) in closed form
        const bool out = (ri == n_rad + 1) || (!SPH && (azj == nz + 1) && (fabs(z) > M.zmaxmax));
        bool killed = false;
        if (star_key >= 0) {
          const int key = ri + (n_rad + 2) * ((zj + nz + 1) + (2 * nz + 3) * (k - 1));
          killed = (key == star_key);
        }
        if (out) {
          st = S_EXITED;  // binned by capteur in the next outer phase
        } else if (killed) {
          c_kill++;
          st = S_EMIT;
        } else {
          const bool real_cell = is_real_cell<L3D>(n_rad, nz, ri, zj);
          double opacity = 0.0, kabs_c = 0.0;
          int ic = 0;
          bool mirrored = false;
          if (real_cell) {
            ic = cell_index<L3D>(n_rad, nz, ri, zj, k);
            if (VAR) { const size_t row = (size_t)M.cell_class[ic] * M.n_lambda + (lambda - 1); opacity = M.v_kappa[row] * kf; kabs_c = M.v_kabs[row]; }
            else { opacity = T.kappa[lambda - 1] * kf; kabs_c = T.kabs[lambda - 1]; }
            if (DARK) {
              if (M.dark[ic]) {  // optical_depth.f90:104-112
                u = -u; v = -v; w = -w;
                x = xo; y = yo; z = zo;
                ri = ri_o; zj = zj_o; k = k_o;
                c_dark++;
                mirrored = true;
                st = S_INTERACT;
                n_inter = 0;
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
              if (real_cell && !MCGPU_DIAG(A.flags, 1)) deposit<LDSE>(A.E_abs, E_lds, ic, kabs_c * lc * S[0]);
              if (real_cell) radiation_field_extras(M, A, ic, lambda, lc * S[0]);
              x = x + lc * u;
              y = y + lc * v;
              z = z + lc * w;
              if (L3D && !SPH) index_cell<L3D>(T, M, x, y, z, ri, zj, k, ri);  // optical_depth.f90:162-165 (lcylindrical only)
              st = S_INTERACT;
              if (MRW) n_inter = first_cross ? (n_inter < 7 ? n_inter + 1 : 7) : 0;  // dust_transfer.f90:1244-1249
            } else {
              first_cross = false;
              extr = extr - tau;
              if (real_cell && !MCGPU_DIAG(A.flags, 1)) deposit<LDSE>(A.E_abs, E_lds, ic, kabs_c * l * S[0]);
              if (real_cell) radiation_field_extras(M, A, ic, lambda, l * S[0]);
              if (DARK) { xo = x; yo = y; zo = z; ri_o = ri; zj_o = zj; k_o = k; }
              x = x1; y = y1; z = z1;
              ri = ri1; zj = zj1; k = k1;
              // the next cell's opacity factor travels through L2 while the next crossing computes
              kf = is_real_cell<L3D>(n_rad, nz, ri, zj) ? M.kappa_factor[cell_index<L3D>(n_rad, nz, ri, zj, k)] : 0.0;
            }
            if (++pk_cross > 200000000u) {  // a packet that never leaves: flag it, drop it
              *A.err = 13;
              st = S_EMIT;
            }
          }
        }
      }
    }
    TP_ADD(tp_fly);
    // ---- barrier-free partial fold: every A.flush_every outer iterations this wave swaps
    // one slice of the workgroup's private grid to zero (ds_wrxchg_rtn_b64) and adds what it
    // took to HBM.  Slices rotate over the waves, so the whole grid keeps flowing into the
    // global sum that the in-flight temperature reads, without any workgroup barrier.
    if (LDSE && ((ep + 1) % A.flush_every) == 0) {
      const int n_waves = (blockDim.x + 63) >> 6, wave = threadIdx.x >> 6;
      const int slice = (wave + (ep + 1) / A.flush_every) % n_waves;
      const int per = (M.n_cells + n_waves - 1) / n_waves;
      const int i0 = slice * per, i1 = (i0 + per < M.n_cells) ? i0 + per : M.n_cells;
      for (int i = i0 + lane; i < i1; i += 64) {
        const unsigned long long bits = atomicExch(reinterpret_cast<unsigned long long*>(&E_lds[i]), 0ull);
        const double e = __longlong_as_double((long long)bits);
        if (e != 0.0) atomic_add_f64(&A.E_abs[i], e);
      }
    }
  }  // outer iterations
  __syncthreads();  // every wave of the workgroup is done emitting and depositing
  lds_flush_sent(T, M, A.n_sent);
  if (LDSE) {  // final fold
    for (int i = threadIdx.x; i < M.n_cells; i += blockDim.x) {
      const double e = E_lds[i];
      if (e != 0.0) atomic_add_f64(&A.E_abs[i], e);
    }
  }

  // ---- counters: wave reduce, one atomic per wave and counter -------------
  unsigned int cs[8] = {c_pack, c_cross, c_flight, c_scatt, c_abs, c_esc, c_kill, c_dark};
#ifdef MCGPU_PHASE_TIMING  // diagnostic build: four counters carry per-wave phase cycles / 1024
  cs[3] = lane == 0 ? (unsigned int)(tp_emit >> 10) : 0u;
  cs[4] = lane == 0 ? (unsigned int)(tp_int >> 10) : 0u;
  cs[6] = lane == 0 ? (unsigned int)(tp_new >> 10) : 0u;
  cs[7] = lane == 0 ? (unsigned int)(tp_fly >> 10) : 0u;
#endif
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    unsigned long long vsum = cs[q];
    for (int off = 32; off > 0; off >>= 1) vsum += __shfl_down(vsum, off);
    if (lane == 0 && vsum) atomicAdd(&A.counters[q], vsum);
  }
  if (MRW) {
    unsigned long long v8 = c_walks, v9 = c_steps;
    for (int off = 32; off > 0; off >>= 1) { v8 += __shfl_down(v8, off); v9 += __shfl_down(v9, off); }
    if (lane == 0 && v8) atomicAdd(&A.counters[8], v8);
    if (lane == 0 && v9) atomicAdd(&A.counters[9], v9);
  }
}

// HBM-deposit variant: 256-thread workgroups, several per CU.
template <bool L3D, bool POLA, bool DARK, bool MRW = false>
__global__ void __launch_bounds__(256) k_thermal(const DevModel M, const RunArgs A) {
  extern __shared__ double lds_raw[];
  thermal_body<L3D, POLA, DARK, false, false, MRW>(M, A, lds_raw);
}

// lvariable_dust: per-class opacity / re-emission / scattering tables gathered from HBM (any cylindrical grid; LDSE: the
// workgroup's private absorbed-energy grid in LDS like k_thermal_lds, otherwise HBM deposits like k_thermal)
template <bool L3D, bool POLA, bool DARK, bool LDSE, bool MRW = false>
__global__ void __launch_bounds__(LDSE ? MCGPU_LDS_BLOCK : 256) k_thermal_var(const DevModel M, const RunArgs A) {
  extern __shared__ double lds_raw[];
  thermal_body<L3D, POLA, DARK, LDSE, false, MRW, true>(M, A, lds_raw);
}

// the spherical grid (spherical_grid.f90): the same packet loop with that grid's operators; no dark zone
template <bool L3D, bool POLA, bool LDSE, bool MRW = false>
__global__ void __launch_bounds__(LDSE ? MCGPU_LDS_BLOCK : 256) k_thermal_sph(const DevModel M, const RunArgs A) {
  extern __shared__ double lds_raw[];
  thermal_body<L3D, POLA, false, LDSE, true, MRW>(M, A, lds_raw);
}

// ... with a dark zone (DARK: the mirror of optical_depth.f90:104-112 at the wall of a flagged cell) and / or dust classes
// (VAR: lvariable_dust) -- round 5; without the random walk
template <bool L3D, bool POLA, bool DARK, bool LDSE, bool VAR>
__global__ void __launch_bounds__(LDSE ? MCGPU_LDS_BLOCK : 256) k_thermal_sph_ext(const DevModel M, const RunArgs A) {
  extern __shared__ double lds_raw[];
  thermal_body<L3D, POLA, DARK, LDSE, true, false, VAR>(M, A, lds_raw);
}

// LDS-deposit variant: one MCGPU_LDS_BLOCK-thread workgroup per CU shares one private grid.
template <bool L3D, bool POLA, bool DARK, bool MRW = false>
__global__ void __launch_bounds__(MCGPU_LDS_BLOCK) k_thermal_lds(const DevModel M, const RunArgs A) {
  extern __shared__ double lds_raw[];
  thermal_body<L3D, POLA, DARK, true, false, MRW>(M, A, lds_raw);
}

// ---------------------------------------------------------------------------
// Temp_finale (thermal_emission.f90:870-906)
// ---------------------------------------------------------------------------
static __global__ void k_temp_finale(const DevModel M, const double* E_abs, const float* tab_Temp, float T_min,
                              float* Tdust) {
  const int ic = blockIdx.x * blockDim.x + threadIdx.x;
  if (ic >= M.n_cells) return;
  int Ti;
  double frac;
  const double E = E_abs[ic];
  const double Qheat = E * M.L_packet_th / M.volume[ic];
  float Temp = T_min;
  const double* lq = M.n_classes ? M.v_lq + (size_t)M.cell_class[ic] * M.n_T : M.log_Qcool;  // (lvariable_dust)
  if (!(Qheat < TINY_DP) && !(log(Qheat) < lq[0])) {
    temp_lte(lq, M.n_T, E, M.L_packet_th, M.volume[ic], Ti, frac);
    // log of a default real is a default-real log (thermal_emission.f90:697)
    Temp = (float)exp((double)logf(tab_Temp[Ti - 1]) * frac + (double)logf(tab_Temp[Ti - 2]) * (1.0 - frac));
  }
  Tdust[ic] = Temp;
}

// ---------------------------------------------------------------------------
// init_reemission (thermal_emission.f90:404-550), the LTE tables: log_Qcool_minus_extra_heating(T, p_icell) and
// kdB_dT_CDF(lambda, T, p_icell), built where they are used (with lvariable_dust they are 280 MB at 7000 cells,
// mem.f90:213-244).  One thread per (class, T): the wavelength sums run in the reference's order.  Without extra heating
// the floor is the cooling rate at tab_Temp(1) (:483-485); with it (round 5) see k_init_reemission's last arguments.
// ---------------------------------------------------------------------------
// sum over lambda of kappa_abs_LTE * B(lambda, T) (:431-452, :468-473); row (may be null) receives the running sum of
// kappa_abs_LTE * dB_dT (:536-541)
__device__ inline double reemission_sums(double Temp, int n_lambda, const double* tab_lambda, const double* tab_delta_lambda,
                                         const double* ka, double* row) {
  const float thermal_const = (float)(299792458.0 * 6.626070040e-34 / 1.38064852e-23);  // constants.f90:24
  const double cst = (double)thermal_const / Temp;
  double integ = 0.0, integ3 = 0.0;
  for (int l = 0; l < n_lambda; ++l) {
    const double wl = tab_lambda[l] * (double)1.e-6f;  // default-real literals (:439-440)
    const double delta_wl = tab_delta_lambda[l] * (double)1.e-6f;
    const double cst_wl = cst / wl;
    double B = 0.0, dB_dT = 0.0;
    if (cst_wl < 500.0) {
      const double coeff_exp = exp(cst_wl);
      const double wl2 = wl * wl, wl5 = (wl2 * wl2) * wl;
      B = 1.0 / (wl5 * (coeff_exp - 1.0)) * delta_wl;
      dB_dT = B * cst_wl * coeff_exp / (coeff_exp - 1.0);
    }
    integ = integ + ka[l] * B;
    integ3 = integ3 + ka[l] * dB_dT;
    if (row) row[l] = integ3;
  }
  return integ;
}

// dudt / hnorm (per class, or null): lextra_heating (:486-494) -- the non-radiative heating of the Phantom coupling: the
// floor becomes max(Qcool0, dudt / hnorm) with hnorm = AU_to_m^2 volume kappa_factor, or, ldudt_implicit (ufac > 0),
// max(Qcool0, (ufac tab_Temp(T) - dudt) / hnorm)
static __global__ void k_init_reemission(int n_classes, int n_T, int n_lambda, const float* tab_Temp, const double* tab_lambda,
                                  const double* tab_delta_lambda, const double* kabs, double* lq, double* cdf,
                                  const double* dudt, const double* hnorm, double ufac) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_classes * n_T) return;
  const int c = idx / n_T, t = idx - c * n_T;
  const double* ka = kabs + (size_t)c * n_lambda;
  double* row = cdf + ((size_t)c * n_T + t) * n_lambda;
  const double cst_E = 2.0 * 6.626070040e-34 * (299792458.0 * 299792458.0) * (4.0 * PI);  // :427
  const double Qcool = reemission_sums((double)tab_Temp[t], n_lambda, tab_lambda, tab_delta_lambda, ka, row) * cst_E;
  // (T = 1 is its own floor, exactly: the two inlined sums need not contract alike)
  const double Qcool0 =
      (t == 0) ? Qcool : reemission_sums((double)tab_Temp[0], n_lambda, tab_lambda, tab_delta_lambda, ka, nullptr) * cst_E;
  double extra = Qcool0;   // (.not.lextra_heating: the equilibrium with the cloud at T_min, :483-485)
  if (dudt) {
    const double Temp = (double)tab_Temp[t];   // (default real in the reference: u_o_dt = ufac_implicit * Temp)
    const double h = (ufac > 0.0) ? (ufac * Temp - dudt[c]) / hnorm[c] : dudt[c] / hnorm[c];
    extra = fmax(Qcool0, h);
  }
  const double q = Qcool - extra;
  lq[(size_t)c * n_T + t] = (q > TINY_DP) ? log(q) : -1000.0;  // :496-504
  const double tot = row[n_lambda - 1];
  const bool ok = tot > TINY_DP;                               // :544-548 (the table stays 0 otherwise)
  for (int l = 0; l < n_lambda; ++l) row[l] = ok ? row[l] / tot : 0.0;
}

// ---------------------------------------------------------------------------
// Temp_approx_diffusion_vertical (diffusion.f90:292-374): the 1+1D diffusion fill of the dark zone (2D cylindrical
// grids).  One workgroup per radius; the column's energy density, its previous value and the diffusion coefficients
// live in LDS; a pseudo-time step of the explicit scheme is three block-wide phases (time step = minimum over the
// column, stencil + largest relative change, refresh of the coefficients that moved by more than 10 %).
//   clean_temperature (:183) is done by the caller's first kernel (k_clean_dark_temperature);
//   temperature_to_DensE (:131), setDiffusion_coeff0 (:78), iter_Temp_approx_diffusion_vertical (:504),
//   setDiffusion_coeff (:17), DensE_to_temperature (:162) are the phases below.
// ---------------------------------------------------------------------------
constexpr int DELTA_CELL_DARK_ZONE = 3;  // cylindrical_grid.f90:39

static __global__ void k_clean_dark_temperature(int n_rad, int ri_in, int ri_out, const int* zj_sup, float T_min, float* Tdust) {
  const int i = ri_in + blockIdx.x;
  if (i > ri_out) return;
  for (int j = 1 + threadIdx.x; j <= zj_sup[i - 1]; j += blockDim.x) Tdust[(i - 1) + n_rad * (j - 1)] = T_min;
}

// the Rosseland-type sum of setDiffusion_coeff[0] for one cell at temperature Temp
__device__ inline double diffusion_coeff(const DevModel& M, const double* tab_lambda, const double* tab_delta_lambda,
                                         int ic, double Temp) {
  const float thermal_const = (float)(299792458.0 * 6.626070040e-34 / 1.38064852e-23);  // constants.f90:24
  const double cst_Dcoeff = PI / (double)(12.0f * 5.670367e-8f);                          // pi/(12.*sigma)
  const double cst = (double)thermal_const / Temp;
  double total_sum = 0.0;
  for (int l = 0; l < M.n_lambda; ++l) {
    const double wl = tab_lambda[l] * (double)1.e-6f;
    const double delta_wl = tab_delta_lambda[l] * (double)1.e-6f;
    const double cst_wl = cst / wl;
    double dB_dT = 0.0;
    if (cst_wl < 200.0) {
      const double coeff_exp = exp(cst_wl);
      const double wl2 = wl * wl, wl5 = (wl2 * wl2) * wl;
      dB_dT = cst_wl * coeff_exp / (wl5 * ((coeff_exp - 1.0) * (coeff_exp - 1.0)));
    }
    // (kappa(p_icell, lambda): the cell's class with lvariable_dust, diffusion.f90:34-60)
    const double kap = M.n_classes ? M.v_kappa[(size_t)M.cell_class[ic] * M.n_lambda + l] : M.kappa[l];
    total_sum = total_sum + dB_dT / (kap * M.kappa_factor[ic]) * delta_wl;
  }
  return cst_Dcoeff * total_sum / (Temp * Temp * Temp);
}

static __global__ void __launch_bounds__(128) k_diffusion_vertical(const DevModel M, const double* tab_lambda,
                                                            const double* tab_delta_lambda, int i_lo, int i_hi,
                                                            const int* zj_sup, float* Tdust, int* n_iter_out,
                                                            int* err) {
  extern __shared__ double lds_raw[];
  const int nz = M.nz, n_rad = M.n_rad;
  double* DensE = lds_raw;            // [0..nz]
  double* DensE_m1 = DensE + nz + 2;  // [0..nz]
  double* Dcoeff = DensE_m1 + nz + 2; // [0..nz]
  __shared__ double red[128];
  const int i = i_lo + blockIdx.x;
  if (i > i_hi) return;
  const int tid = threadIdx.x, nt = blockDim.x;
  int jtop = zj_sup[i - 1] + DELTA_CELL_DARK_ZONE;
  if (jtop > nz - 1) jtop = nz - 1;  // the stencil reads j+1
  const double dz = M.ch[i - 1];     // cell_height(i,j): the uniform vertical grid
  const double dz2 = dz * dz;
  for (int j = 1 + tid; j <= nz; j += nt) {
    const float T = Tdust[(i - 1) + n_rad * (j - 1)];
    DensE[j] = (double)((T * T) * (T * T));  // Tdust**4 in default real (:151)
    Dcoeff[j] = diffusion_coeff(M, tab_lambda, tab_delta_lambda, (i - 1) + n_rad * (j - 1), (double)T);
  }
  __syncthreads();
  if (tid == 0) { DensE[0] = DensE[1]; Dcoeff[0] = Dcoeff[1]; }
  __syncthreads();
  const float precision = 1.0e-6f, stabilite = 2.0f;
  int n_iter = 0;
  for (;;) {
    n_iter++;
    // time step: stabilite * 0.5 * min(dz^2 / D) over the zone (:527-534)
    double tmin = HUGE_DP;
    for (int j = 1 + tid; j <= jtop; j += nt) tmin = fmin(tmin, dz2 / Dcoeff[j]);
    red[tid] = tmin;
    for (int j = tid; j <= nz; j += nt) DensE_m1[j] = DensE[j];
    __syncthreads();
    for (int sft = nt >> 1; sft > 0; sft >>= 1) {
      if (tid < sft) red[tid] = fmin(red[tid], red[tid + sft]);
      __syncthreads();
    }
    const double dt = (double)(stabilite * 0.5f) * red[0];
    __syncthreads();
    // one explicit step (:545-583)
    float dmax = 0.0f;
    for (int j = 1 + tid; j <= jtop; j += nt) {
      const double dE_dz_p1 = DensE_m1[j + 1] - DensE_m1[j];
      const double dE_dz_m1 = DensE_m1[j] - DensE_m1[j - 1];
      const double delta_E = Dcoeff[j] * (dE_dz_p1 - dE_dz_m1) / (2.0 * dz2) * dt;
      const double e = DensE_m1[j] + delta_E;
      DensE[j] = e;
      const double r = delta_E / e;
      if (r > (double)dmax) dmax = (float)r;
    }
    red[tid] = (double)dmax;
    __syncthreads();
    for (int sft = nt >> 1; sft > 0; sft >>= 1) {
      if (tid < sft) red[tid] = fmax(red[tid], red[tid + sft]);
      __syncthreads();
    }
    const float max_delta_E_r = (float)red[0];
    if (tid == 0) DensE[0] = DensE[1];  // no flux through the midplane (:586)
    __syncthreads();
    if (max_delta_E_r < precision) break;
    if (n_iter > 50000000) { if (tid == 0) *err = 23; break; }
    // setDiffusion_coeff (:17-74)
    for (int j = 1 + tid; j <= nz; j += nt)
      if (fabs(DensE[j] - DensE_m1[j]) > 1.0e-1 * DensE_m1[j])
        Dcoeff[j] = diffusion_coeff(M, tab_lambda, tab_delta_lambda, (i - 1) + n_rad * (j - 1), pow(DensE[j], 0.25));
    __syncthreads();
    if (tid == 0) Dcoeff[0] = Dcoeff[1];
    __syncthreads();
  }
  for (int j = 1 + tid; j <= jtop; j += nt) Tdust[(i - 1) + n_rad * (j - 1)] = (float)pow(DensE[j], 0.25);
  if (tid == 0) atomicAdd(n_iter_out, n_iter);
}

// ---------------------------------------------------------------------------
// Probes for the parity tests
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// repartition_energie (thermal_emission.f90:1771-1949), LTE grains: E_cell of one wavelength (one thread per cell),
// then the cumulative distribution prob_E_cell(0:n_cells) summed in the reference's own order (k_cumsum_in_order).
// ---------------------------------------------------------------------------
static __global__ void k_repart_E_cell(const DevModel M, int lambda, double wl, const float* Tdust, const float* weight,
                                double* E_cell, double* E_corr) {
  const int ic = blockIdx.x * blockDim.x + threadIdx.x;
  if (ic >= M.n_cells) return;
  const float thermal_const = (float)(299792458.0 * 6.626070040e-34 / 1.38064852e-23);  // real (constants.f90:24)
  const double cst_wl_max = 88.72283905206835 - (double)1.0e-4f;                          // log(huge_real) - 1.0e-4 (:1802)
  double E = 0.0;
  if (!(M.dark && M.dark[ic])) {
    const double Temp = (double)Tdust[ic];
    if (!(Temp < TINY_REAL)) {
      const double cst_wl = (double)thermal_const / (Temp * wl);
      if (cst_wl < cst_wl_max) {
        const double kabs = M.n_classes ? M.v_kabs[(size_t)M.cell_class[ic] * M.n_lambda + (lambda - 1)] : M.kappa_abs[lambda - 1];
        const double wl2 = wl * wl, wl5 = (wl2 * wl2) * wl;
        E = 4.0 * kabs * M.kappa_factor[ic] * M.volume[ic] / (wl5 * (exp(cst_wl) - 1.0));
      }
    }
  }
  E_cell[ic] = E;
  E_corr[ic] = weight ? E * (double)weight[ic] : E;
}

constexpr int SCAN_TILE = 1024;
// prob(0:n) = the running sum of in[0..n) in the REFERENCE's order (:1923-1926: cell by cell, which also makes the
// distribution monotone and its last entry, divided by itself, exactly 1), tot[0] = its last entry, tot[1] = the
// running sum of in2 (E_disk = sum(E_cell), :1897).  One workgroup: the tiles are staged in LDS by all threads, the
// additions are one thread's -- 3 ms per million cells, next to seconds of Monte Carlo per wavelength.
static __global__ void __launch_bounds__(SCAN_TILE) k_cumsum_in_order(const double* in, const double* in2, int n, double* prob, double* tot) {
  __shared__ double buf[SCAN_TILE], buf2[SCAN_TILE];
  __shared__ double carry, carry2;
  if (threadIdx.x == 0) { carry = 0.0; carry2 = 0.0; prob[0] = 0.0; }
  __syncthreads();
  for (int base = 0; base < n; base += SCAN_TILE) {
    const int i = base + threadIdx.x;
    buf[threadIdx.x] = i < n ? in[i] : 0.0;
    buf2[threadIdx.x] = i < n ? in2[i] : 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int m = n - base < SCAN_TILE ? n - base : SCAN_TILE;
      double c = carry, c2 = carry2;
      for (int q = 0; q < m; ++q) { c = c + buf[q]; buf[q] = c; c2 = c2 + buf2[q]; }
      carry = c; carry2 = c2;
    }
    __syncthreads();
    if (i < n) prob[1 + i] = buf[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) { tot[0] = carry; tot[1] = carry2; }
}
// prob(:) = prob(:) / prob(n_cells), or 0 when that is not positive (:1933-1937)
static __global__ void k_cumsum_normalise(double* prob, const double* tot, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  const double last = tot[0];
  prob[i] = last > 2.2250738585072014e-308 ? prob[i] / last : 0.0;
}

template <bool L3D, bool SPH = false>
__global__ void k_probe_cross(const DevModel M, int n, const double* x0, const double* y0,
                              const double* z0, const double* u, const double* v, const double* w,
                              const int* cmi, const int* cmj, const int* cmk, const int* cell, double* x1,
                              double* y1, double* z1, int* next_cell, double* l) {
  extern __shared__ double lds_raw[];
  const Lds T = lds_carve(lds_raw, M);
  lds_stage(T, M);
  __syncthreads();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double a = u[i] * u[i] + v[i] * v[i];
    const double inv_a = (a > TINY_REAL) ? 1.0 / a : HUGE_REAL;
    const double inv_w = (fabs(w[i]) > TINY_REAL) ? 1.0 / w[i] : copysign(HUGE_DP, w[i]);
    const int c = cell[i] - 1;
    int ri1, zj1, k1;
    if (SPH) cross_cell_sph<L3D>(T, M, x0[i], y0[i], z0[i], u[i], v[i], w[i], cmi[c], cmj[c], cmk[c], x1[i], y1[i], z1[i],
                                 ri1, zj1, k1, l[i]);
    else MCGPU_CROSS<L3D>(T, M, x0[i], y0[i], z0[i], u[i], v[i], w[i], inv_a, inv_w, cmi[c], cmj[c], cmk[c],
                          x1[i], y1[i], z1[i], ri1, zj1, k1, l[i]);
    next_cell[i] = icell_of(M.n_rad, M.nz, M.n_az, M.l3D, ri1, zj1, k1);
  }
}

template <bool L3D, bool SPH = false>
__global__ void k_probe_index(const DevModel M, int n, const double* x, const double* y, const double* z,
                              int* icell) {
  extern __shared__ double lds_raw[];
  const Lds T = lds_carve(lds_raw, M);
  lds_stage(T, M);
  __syncthreads();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int ri, zj, k;
    if (SPH) index_cell_sph<L3D>(T, M, x[i], y[i], z[i], ri, zj, k);
    else index_cell<L3D>(T, M, x[i], y[i], z[i], ri, zj, k);
    icell[i] = icell_of(M.n_rad, M.nz, M.n_az, M.l3D, ri, zj, k);
  }
}

static __global__ void k_probe_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                               uint32_t k1, uint32_t* out) {
  uint32_t o[4];
  philox4x32_10(c0, c1, c2, c3, k0, k1, o);
  out[0] = o[0]; out[1] = o[1]; out[2] = o[2]; out[3] = o[3];
}

static __global__ void k_probe_rand(uint64_t seed, uint64_t packet, int n, float* out) {
  Rng r;
  r.init(seed, packet);
  for (int i = 0; i < n; i += 4) {
    float f[4];
    r.block((uint32_t)(i / 4), f);
    for (int q = 0; q < 4 && i + q < n; ++q) out[i + q] = f[q];
  }
}

}  // namespace mcgpu
