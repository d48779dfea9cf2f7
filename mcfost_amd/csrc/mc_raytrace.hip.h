// Ray-traced SED of the dust, ray-tracing method 1: what dust_map(lambda,ibin,iaz)
// (dust_transfer.f90:1413-1600) adds to Stokes_ray_tracing(lambda,1,1,ibin,iaz,:) with RT_sed_method = 1.
//   k_calc_Jth      calc_Jth (dust_ray_tracing.f90:810-846, LTE grains): thermal emissivity per cell
//   k_rt1_dust_map  one ray per lane: the 128 x 30 log-r / uniform-phi sampling of the image plane of every
//                   observer direction (:1481-1535); intensite_pixel_dust with one sub-pixel (:1899-2004):
//                   move_to_grid from far away, then integ_ray_dust (optical_depth.f90:1327-1421), the formal
//                   solution sum exp(-tau) (1 - exp(-dtau)) S with the RT1 source function
//                   eps_dust1(k,psup,:,icell) (dust_ray_tracing.f90:1455-1475), built on the fly from the
//                   xI_scatt records the SED Monte Carlo left in HBM and J_th (init_dust_source_fct1 :636-716).
//   k_rt1_image     dust_map method 2 (:1537-1577): square pixels, one WAVEFRONT per pixel; intensite_pixel_dust's
//                   refinement 1, 2x2, 4x4 ... 32x32 sub-pixels (at least 2 iterations, at most 6, until Stokes I
//                   changes by < 1 %) with the sub-pixels of an iteration spread over the 64 lanes.
// The stellar term (compute_stars_map: 1024 random rays per star on the host) is not part of these kernels.
#pragma once
#include "mc_mono.hip.h"
#include "mc_voronoi.hip.h"

namespace mcgpu {

struct RtArgs {
  int lambda, RT_n_incl, nRT, n_az_rt, n_theta_rt, N_type_flux, contrib, l_sym_ima;
  double wl, photon_energy, pix_scale;  // metres; (:661-663); 1 / (distance * pc_to_AU)
  double ang_disque, tau_dark_zone_obs, rmin_RT, fact_r, fact_A, cst_phi, l_far;
  const double* rt_u; const double* rt_v; const double* rt_w;  // observer directions (see MonoArgs)
  const float* rt_az;                                           // tab_RT_az [RT_n_az], degrees
  const double* xI;                                             // device layout [cell][psup][phik][iRT][XI_LINE]
  int xI_f32; Xi32Lay xi;                                       // ... or the packed default-real layout (MonoArgs::xI_f32, xi32_*)
  const double* J_th;                                           // [n_cells]
  double* out;                                                  // [nRT * N_type_flux]
  // images (k_rt1_image)
  int npix_x, npix_y, npix_x_max;
  double taille_pix;                                            // AU
  double* image;                                                // [N_type_flux][RT_n_az][RT_n_incl][npix_y][npix_x]
  unsigned long long* n_rays;
  const VoroGrid* voro;                                         // stars' maps on a Voronoi grid: the grid's record in HBM (else null)
  // method 2 (mcgpu_rt2_dust_map / mcgpu_rt2_image): the source function of ONE inclination, observer q_only
  int method2, q_only, nang_rt, nang_star;
  const float* eps2;       // eps_dust2(N_type_flux, nang_rt, 0:1, n_cells)
  const float* eps2_star;  // eps_dust2_star(n_Stokes, nang_star, 0:1, n_cells)
  const double* z_grid;    // z_grid(n_cells)
};

constexpr int RT_N_RAD = 128, RT_N_PHI = 30;  // dust_map (:1434)

static __global__ void k_calc_Jth(const DevModel M, int lambda, double wl, const float* Tdust, double* J_th) {
  const int ic = blockIdx.x * blockDim.x + threadIdx.x;
  if (ic >= M.n_cells) return;
  const double cst_E = 2.0 * 6.626070040e-34 * 299792458.0 * 299792458.0;
  const float thermal_const = (float)(299792458.0 * 6.626070040e-34 / 1.38064852e-23);  // real (constants.f90:24)
  const double Temp = (double)Tdust[ic];
  double j = 0.0;
  if (Temp * wl > 3.e-4) {
    const double cst_wl = (double)thermal_const / (Temp * wl);
    const double coeff_exp = exp(cst_wl);
    // (lvariable_dust: kappa_abs_LTE of the cell's class)
    const double kabs = M.n_classes ? M.v_kabs[(size_t)M.cell_class[ic] * M.n_lambda + (lambda - 1)] : M.kappa_abs[lambda - 1];
    j = cst_E / (pow(wl, 5) * (coeff_exp - 1.0)) * wl * kabs * M.kappa_factor[ic];
  }
  J_th[ic] = j;
}

// rotation_3d (utils.f90:1545-1589)
__device__ inline void rotation_3d(const double axis[3], double angle_deg, const double v[3], double out[3]) {
  const double d = v[0] * axis[0] + v[1] * axis[1] + v[2] * axis[2];
  const double vp[3] = {d * axis[0], d * axis[1], d * axis[2]};
  double vn[3] = {v[0] - vp[0], v[1] - vp[1], v[2] - vp[2]};
  const double norm = sqrt(vn[0] * vn[0] + vn[1] * vn[1] + vn[2] * vn[2]);
  if (norm < TINY_DP) { out[0] = vp[0]; out[1] = vp[1]; out[2] = vp[2]; return; }
  vn[0] /= norm; vn[1] /= norm; vn[2] /= norm;
  const double vn2[3] = {axis[1] * vn[2] - axis[2] * vn[1], axis[2] * vn[0] - axis[0] * vn[2],
                         axis[0] * vn[1] - axis[1] * vn[0]};
  double sa, ca;
  sincos(angle_deg * (PI / 180.0), &sa, &ca);
  for (int q = 0; q < 3; ++q) out[q] = vp[q] + norm * (ca * vn[q] + sa * vn2[q]);
}

// image-plane basis of an observer direction (dust_transfer.f90:1440-1455)
__device__ inline void rt_image_plane(const RtArgs& A, int q, double uvw[3], double xpi[3], double ypi[3]) {
  uvw[0] = A.rt_u[q]; uvw[1] = A.rt_v[q]; uvw[2] = A.rt_w[q % A.RT_n_incl];
  double sa, ca;
  sincos((double)A.rt_az[q / A.RT_n_incl] * (PI / 180.0), &sa, &ca);
  const double xv[3] = {ca, sa, 0.0};
  if (fabs(A.ang_disque) > TINY_REAL) rotation_3d(uvw, A.ang_disque, xv, xpi);
  else { xpi[0] = xv[0]; xpi[1] = xv[1]; xpi[2] = xv[2]; }
  ypi[0] = -(xpi[1] * uvw[2] - xpi[2] * uvw[1]);
  ypi[1] = -(xpi[2] * uvw[0] - xpi[0] * uvw[2]);
  ypi[2] = -(xpi[0] * uvw[1] - xpi[1] * uvw[0]);
}

// ---- ray tracing method 2: dust_source_fct (dust_ray_tracing.f90:1478-1700) on the eps_dust2 / eps_dust2_star that
// mcgpu_rt2_source left in HBM.  interpolate_Stokes_QU (:1705-1742): between two (P I, 2 theta) pairs, back to (Q, U).
__device__ inline void interpolate_stokes_qu(const float* a, const float* b, double frac1, float out[2]) {
  const float PxI1 = a[0], PxI2 = b[0];
  float two_theta1 = a[1], two_theta2 = b[1];
  const float PxI = (float)((double)PxI2 * (1.0 - frac1) + (double)PxI1 * frac1);
  if ((double)fabsf(two_theta2 - two_theta1) >= PI) {
    if (two_theta2 > two_theta1) two_theta1 = (float)((double)two_theta1 + 2 * PI);
    else two_theta2 = (float)((double)two_theta2 + 2 * PI);
  }
  const float two_theta = (float)((double)two_theta2 * (1.0 - frac1) + (double)two_theta1 * frac1);
  out[0] = PxI * cosf(two_theta);
  out[1] = PxI * (-sinf(two_theta));
}

// one corner of the interpolation: cell ic (0-based), between the tabulated directions around phi_pos
template <bool POLA>
__device__ inline void rt2_corner(const RtArgs& A, int ic, int dir, double phi_pos, double SF[8]) {
  const int n_Stokes = POLA ? 4 : 1, ntf = A.N_type_flux;
#pragma unroll
  for (int t = 0; t < 8; ++t) SF[t] = 0.0;
  {
    const int N = A.nang_rt;
    const double xiscatt = fmax(phi_pos / (2 * PI) * (double)N, 0.0);
    int iscatt1 = (int)floor(xiscatt);
    const double frac = xiscatt - iscatt1, un_m_frac = 1.0 - frac;
    int iscatt2 = iscatt1 + 1;
    iscatt1 = ((iscatt1 % N) + N) % N; if (iscatt1 == 0) iscatt1 = N;
    iscatt2 = ((iscatt2 % N) + N) % N; if (iscatt2 == 0) iscatt2 = N;
    const float* e1 = A.eps2 + (size_t)ntf * ((size_t)(iscatt1 - 1) + (size_t)N * (dir + 2 * (size_t)ic));
    const float* e2 = A.eps2 + (size_t)ntf * ((size_t)(iscatt2 - 1) + (size_t)N * (dir + 2 * (size_t)ic));
    SF[0] = (double)e2[0] * frac + (double)e1[0] * un_m_frac;
    if (POLA) {
      float qu[2];
      interpolate_stokes_qu(e1 + 1, e2 + 1, un_m_frac, qu);
      SF[1] = (double)qu[0]; SF[2] = (double)qu[1];
    }
    if (A.contrib)
      for (int t = n_Stokes; t < ntf; ++t) SF[t] = (double)e2[t] * frac + (double)e1[t] * un_m_frac;
  }
  {
    const int N = A.nang_star;
    const double xiscatt = fmax(phi_pos / (2 * PI) * (double)N, 0.0);
    int iscatt1 = (int)floor(xiscatt);
    const double frac = xiscatt - iscatt1, un_m_frac = 1.0 - frac;
    int iscatt2 = iscatt1 + 1;
    iscatt1 = ((iscatt1 % N) + N) % N; if (iscatt1 == 0) iscatt1 = N;
    iscatt2 = ((iscatt2 % N) + N) % N; if (iscatt2 == 0) iscatt2 = N;
    const float* e1 = A.eps2_star + (size_t)n_Stokes * ((size_t)(iscatt1 - 1) + (size_t)N * (dir + 2 * (size_t)ic));
    const float* e2 = A.eps2_star + (size_t)n_Stokes * ((size_t)(iscatt2 - 1) + (size_t)N * (dir + 2 * (size_t)ic));
    SF[0] = (SF[0] + (double)e2[0] * frac) + (double)e1[0] * un_m_frac;
    if (POLA) {
      float qu[2];
      interpolate_stokes_qu(e1 + 1, e2 + 1, un_m_frac, qu);
      SF[1] = SF[1] + (double)qu[0]; SF[2] = SF[2] + (double)qu[1];
    }
    if (A.contrib) SF[n_Stokes + 1] = (SF[n_Stokes + 1] + (double)e2[0] * frac) + (double)e1[0] * un_m_frac;
  }
}

// dust_source_fct, method 2: linear in z between the cell and its vertical neighbour on the point's side (the radial
// interpolation is switched off in the reference: ri1 = ri, frac_r = 1), linear in azimuth between the directions
template <bool POLA>
__device__ inline void dust_source_fct2(const DevModel& M, const RtArgs& A, int ri, int zj, double x, double y, double z,
                                        double SF[8]) {
#pragma clang fp contract(off)
  const int n_rad = M.n_rad, nz = M.nz;
  zj = zj < 0 ? -zj : zj;   // cell_map_j of the 2D cell
  const int ic = (ri - 1) + n_rad * (zj - 1);
  int zj1, zj2;
  double frac_z;
  if (fabs(z) > A.z_grid[ic]) { zj1 = zj; zj2 = zj + 1; } else { zj1 = zj - 1; zj2 = zj; }
  if (zj2 > nz) { zj2 = nz; frac_z = 1.0; }
  else if (zj1 < 1) { zj1 = 1; frac_z = 1.0; }
  else {
    const double za = A.z_grid[(ri - 1) + n_rad * (zj2 - 1)], zb = A.z_grid[(ri - 1) + n_rad * (zj1 - 1)];
    frac_z = (za - fabs(z)) / (za - zb);
  }
  frac_z = fmax(fmin(1.0, frac_z), 0.0);
  const double phi_pos = modulo_d(atan2(x, y) + 2 * PI, 2 * PI);
  const int dir = z > 0.0 ? 1 : 0;
  double SF1[8], SF3[8];
  rt2_corner<POLA>(A, (ri - 1) + n_rad * (zj1 - 1), dir, phi_pos, SF1);
  rt2_corner<POLA>(A, (ri - 1) + n_rad * (zj2 - 1), dir, phi_pos, SF3);
  const double frac_r = 1.0;
#pragma unroll
  for (int t = 0; t < 8; ++t) SF[t] = frac_r * frac_z * SF1[t] + frac_r * (1.0 - frac_z) * SF3[t];
}

// move_to_grid + integ_ray_dust (optical_depth.f90:1327-1421) for observer q from the point (x,y,z) of the image
// plane, propagating along (u0,v0,w0) = -(direction to the observer).  S[0..N_type_flux) is overwritten.
template <bool L3D, bool POLA>
__device__ inline void rt1_integ_ray(const Lds& T, const DevModel& M, const RtArgs& A, int q, double x, double y,
                                     double z, double u0, double v0, double w0, double S[8]) {
  const int n_rad = M.n_rad, nz = M.nz;
  const int n_Stokes = POLA ? 4 : 1;
#pragma unroll
  for (int t = 0; t < 8; ++t) S[t] = 0.0;
  int ri, zj, k;
  const bool sph = M.grid_sph != 0;   // (the ray tracer picks the grid's operators at run time: not the hot path)
  if (!(sph ? move_to_grid_sph<L3D>(T, M, x, y, z, u0, v0, w0, ri, zj, k) : move_to_grid<L3D>(T, M, x, y, z, u0, v0, w0, ri, zj, k))) return;
  const double a = u0 * u0 + v0 * v0;
  const double inv_a = (a > TINY_REAL) ? 1.0 / a : HUGE_REAL;
  const double inv_w = (fabs(w0) > TINY_REAL) ? 1.0 / w0 : copysign(HUGE_DP, w0);
  const int i_star = intersect_stars(M, x, y, z, u0, v0, w0);
  int star_key = -1;
  if (i_star > 0) {
    const int* sc = &M.star_cell[4 * (i_star - 1)];
    star_key = sc[0] + (n_rad + 2) * ((sc[1] + nz + 1) + (2 * nz + 3) * (sc[2] - 1));
  }
  double tau = 0.0;
  for (long guard = 0; guard < 100000000L; ++guard) {
    const int azj = zj < 0 ? -zj : zj;
    if ((ri == n_rad + 1) || (!sph && (azj == nz + 1) && (fabs(z) > M.zmaxmax))) break;  // test_exit_grid
    if (star_key >= 0 && (ri + (n_rad + 2) * ((zj + nz + 1) + (2 * nz + 3) * (k - 1))) == star_key) break;
    double x1, y1, z1, l;
    int ri1, zj1, k1;
    if (sph) cross_cell_sph<L3D>(T, M, x, y, z, u0, v0, w0, ri, zj, k, x1, y1, z1, ri1, zj1, k1, l);
    else MCGPU_CROSS<L3D>(T, M, x, y, z, u0, v0, w0, inv_a, inv_w, ri, zj, k, x1, y1, z1, ri1, zj1, k1, l);
    if (is_real_cell<L3D>(n_rad, nz, ri, zj)) {
      const int ic = cell_index<L3D>(n_rad, nz, ri, zj, k);
      const size_t vrow = M.n_classes ? (size_t)M.cell_class[ic] * M.n_lambda + (A.lambda - 1) : 0;  // (lvariable_dust)
      const double kappa_ext = (M.n_classes ? M.v_kappa[vrow] : T.kappa[A.lambda - 1]) * M.kappa_factor[ic];
      const double dtau = l * kappa_ext;
      int phik = 1, psup = 1;
      rt1_subbin_of(A.n_az_rt, L3D, x, y, z, x1, y1, z1, phik, psup);
      if (!L3D && A.method2) {  // the interpolated source function of method 2 at the middle of the path (:1396-1404)
        double SF[8];
        dust_source_fct2<POLA>(M, A, ri, zj, 0.5 * (x + x1), 0.5 * (y + y1), 0.5 * (z + z1), SF);
        const double wgt = exp(-tau) * (1.0 - exp(-dtau));
        for (int t = 0; t < A.N_type_flux; ++t) S[t] += wgt * SF[t];
      } else if (kappa_ext > TINY_DP) {
        const double factor = A.photon_energy / M.volume[ic] * A.n_az_rt * A.n_theta_rt;
        const double kappa_sca = kappa_ext * (double)(M.n_classes ? M.v_albedo[vrow] : T.albedo[A.lambda - 1]);
        const size_t bin = ((size_t)ic * A.n_theta_rt + (psup - 1)) * A.n_az_rt + (phik - 1);
        double rec[XI_LINE];
        if (A.xI_f32) {
          // (the packed default-real layout, mc_mono.hip.h xi32_*: the observer's values side by side; a flux type no
          // deposit reaches reads as 0)
          const float* b32 = reinterpret_cast<const float*>(A.xI) + bin * A.xi.binf;
#pragma unroll
          for (int t = 0; t < XI_LINE; ++t) rec[t] = t < A.N_type_flux ? xi32_value(b32, A.xi, q, t, n_Stokes) : 0.0;
        } else {
          const double* r64 = A.xI + (bin * A.nRT + q) * XI_LINE;
#pragma unroll
          for (int t = 0; t < XI_LINE; ++t) rec[t] = r64[t];
        }
        const double wgt = exp(-tau) * (1.0 - exp(-dtau));
        const double jth = A.J_th[ic];
        const double fs = factor * kappa_sca / kappa_ext;
        S[0] += wgt * (rec[0] * fs + jth / kappa_ext);
        if (POLA) { S[1] += wgt * rec[1] * fs; S[2] += wgt * rec[2] * fs; S[3] += wgt * rec[3] * fs; }
        if (A.contrib) {
          S[n_Stokes + 1] += wgt * rec[n_Stokes + 1] * fs;
          S[n_Stokes + 2] += wgt * (jth / kappa_ext);
          S[n_Stokes + 3] += wgt * rec[n_Stokes + 3] * fs;
        }
      }
      tau += dtau;
      if (tau > A.tau_dark_zone_obs) break;
    }
    x = x1; y = y1; z = z1;
    ri = ri1; zj = zj1; k = k1;
  }
}

// (Ray: the ray integration of the grid -- rt1_integ_ray, or rt1_integ_ray_voro of mc_raytrace_voronoi.hip.h)
template <typename Ray>
__device__ __forceinline__ void rt1_dust_map_body(const DevModel& M, const RtArgs& A, double* lds_raw, Ray integ_ray) {
  const Lds T = lds_carve(lds_raw, M, true);  // geometry, kappa, albedo: the SED-mode table set
  lds_stage_mono(T, M, 1);
  __syncthreads();
  const int rays_per_dir = RT_N_RAD * RT_N_PHI;  // 3840 = 60 wavefronts: a wavefront never straddles two directions
  const int n_rays = (A.method2 ? 1 : A.nRT) * rays_per_dir;  // (method 2: the one inclination of its source function)
  const int lane = threadIdx.x & 63;

  for (int base = (blockIdx.x * blockDim.x + (threadIdx.x & ~63)); base < n_rays; base += gridDim.x * blockDim.x) {
    const int ray = base + lane;  // (n_rays is a multiple of 64)
    const int q0 = ray / rays_per_dir, rem = ray - q0 * rays_per_dir;
    const int q = A.method2 ? A.q_only : q0;
    const int ri_RT = rem / RT_N_PHI, phi_RT = rem - ri_RT * RT_N_PHI + 1;
    double uvw[3], xpi[3], ypi[3];
    rt_image_plane(A, q, uvw, xpi, ypi);
    // tab_r(ri) = rmin_RT * fact_r**(ri-1), built by repeated products like the reference (:1499-1503)
    double r = A.rmin_RT;
    for (int i = 0; i < ri_RT; ++i) r = r * A.fact_r;
    const double taille_pix = A.fact_A * r;
    const double phi = A.cst_phi * ((double)phi_RT - 0.5);
    double sp, cp;
    sincos(phi, &sp, &cp);
    const double x = uvw[0] * A.l_far + r * sp * xpi[0] + r * cp * ypi[0];
    const double y = uvw[1] * A.l_far + r * sp * xpi[1] + r * cp * ypi[1];
    const double z = uvw[2] * A.l_far + r * sp * xpi[2] + r * cp * ypi[2];
    double S[8];
    integ_ray(T, q, x, y, z, -uvw[0], -uvw[1], -uvw[2], S);  // reverse propagation
    const double pix = taille_pix * A.pix_scale;
    for (int t = 0; t < A.N_type_flux; ++t) {
      double vsum = S[t] * pix * pix;
      for (int off = 32; off > 0; off >>= 1) vsum += __shfl_down(vsum, off);
      if (lane == 0 && vsum != 0.0) atomic_add_f64(&A.out[(size_t)q * A.N_type_flux + t], vsum);
    }
  }
}

template <bool L3D, bool POLA>
__global__ void __launch_bounds__(256) k_rt1_dust_map(const DevModel M, const RtArgs A) {
  extern __shared__ double lds_raw[];
  rt1_dust_map_body(M, A, lds_raw, [&](const Lds& T, int q, double x, double y, double z, double u, double v, double w, double* S) {
    rt1_integ_ray<L3D, POLA>(T, M, A, q, x, y, z, u, v, w, S);
  });
}

template <typename Ray>
__device__ __forceinline__ void rt1_image_body(const DevModel& M, const RtArgs& A, double* lds_raw, Ray integ_ray) {
  const Lds T = lds_carve(lds_raw, M, true);  // geometry, kappa, albedo: the SED-mode table set
  lds_stage_mono(T, M, 1);
  __syncthreads();
  const int lanes = blockDim.x < 64 ? (int)blockDim.x : 64;  // (the CPU emulation of the tests runs one lane)
  const int lane = threadIdx.x % lanes;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) / lanes;
  const long n_waves = (long)gridDim.x * blockDim.x / lanes;
  const long pix_per_dir = (long)A.npix_x_max * A.npix_y, n_pix = pix_per_dir * (A.method2 ? 1 : A.nRT);
  const int n_iter_min = 2, n_iter_max = 6;  // dust_map (:1566-1567)
  const double precision = 1.e-2;            // intensite_pixel_dust (:1921)
  unsigned long long rays = 0;

  for (long pix = wave; pix < n_pix; pix += n_waves) {
    const int q0 = (int)(pix / pix_per_dir);
    const long rem = pix - (long)q0 * pix_per_dir;
    const int q = A.method2 ? A.q_only : q0;
    const int i = (int)(rem / A.npix_y) + 1, j = (int)(rem - (long)(i - 1) * A.npix_y) + 1;
    double uvw[3], xpi[3], ypi[3], corner[3], dx[3], dy[3];
    rt_image_plane(A, q, uvw, xpi, ypi);
    for (int c = 0; c < 3; ++c) {
      dx[c] = xpi[c] * A.taille_pix;
      dy[c] = ypi[c] * A.taille_pix;
      const double Icorner = uvw[c] * A.l_far - (0.5 * A.npix_x * dx[c] + 0.5 * A.npix_y * dy[c]);
      corner[c] = Icorner + (i - 1) * dx[c] + (j - 1) * dy[c];
    }
    double S[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int subpixels = 1;
    for (int iter = 1;; ++iter) {
      const double S_old = S[0];
      double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      double sdx[3], sdy[3];
      for (int c = 0; c < 3; ++c) { sdx[c] = dx[c] / (double)subpixels; sdy[c] = dy[c] / (double)subpixels; }
      const int n_sub = subpixels * subpixels;
      for (int s = lane; s < n_sub; s += lanes) {
        const int si = s / subpixels + 1, sj = s - (si - 1) * subpixels + 1;
        const double x = corner[0] + (si - 0.5) * sdx[0] + (sj - 0.5) * sdy[0];
        const double y = corner[1] + (si - 0.5) * sdx[1] + (sj - 0.5) * sdy[1];
        const double z = corner[2] + (si - 0.5) * sdx[2] + (sj - 0.5) * sdy[2];
        double R[8];
        integ_ray(T, q, x, y, z, -uvw[0], -uvw[1], -uvw[2], R);
        rays++;
        for (int t = 0; t < 8; ++t) acc[t] += R[t];
      }
      // all-reduce over the wavefront (xor butterfly: every lane ends with the same bits, so the refinement
      // decision below is wave-uniform)
      const double npix2 = (double)n_sub;
      for (int t = 0; t < A.N_type_flux; ++t) {
        double vsum = acc[t];
        for (int off = 32; off > 0; off >>= 1) vsum += __shfl_xor(vsum, off);
        S[t] = vsum / npix2;
      }
      if (iter < n_iter_min) subpixels *= 2;
      else if (iter >= n_iter_max) break;
      else if (fabs(S[0] - S_old) > precision * S_old) subpixels *= 2;
      else break;
    }
    if (lane == 0) {
      const double pixs = A.taille_pix * A.pix_scale;
      const int iaz = q / A.RT_n_incl, ibin = q - iaz * A.RT_n_incl, n_az = A.nRT / A.RT_n_incl;
      for (int t = 0; t < A.N_type_flux; ++t)
        A.image[((((size_t)t * n_az + iaz) * A.RT_n_incl + ibin) * A.npix_y + (j - 1)) * A.npix_x + (i - 1)] =
            S[t] * (pixs * pixs);
    }
  }
  if (A.n_rays) {
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off);
    if (lane == 0) atomicAdd(A.n_rays, rays);
  }
}

template <bool L3D, bool POLA>
__global__ void __launch_bounds__(256) k_rt1_image(const DevModel M, const RtArgs A) {
  extern __shared__ double lds_raw[];
  rt1_image_body(M, A, lds_raw, [&](const Lds& T, int q, double x, double y, double z, double u, double v, double w, double* S) {
    rt1_integ_ray<L3D, POLA>(T, M, A, q, x, y, z, u, v, w, S);
  });
}

// ---------------------------------------------------------------------------
// compute_stars_map for the SED (dust_transfer.f90:1604-1854 with lresolved = .false., no limb darkening): the
// stars' flux towards every observer.  Per (observer, star): a 21 x 21 screen of optical depths in front of the star
// (optical_length_tot from points of the star's disc towards the observer), then n_ray_star_SED / n_stars random
// points of the stellar sphere, each with the screen's bilinearly interpolated optical depth; the flux is
// star_flux * sum(exp(-tau) cos_thet) / sum(cos_thet).  One workgroup per (observer, star).  The reference draws
// the points from SPRNG; here ray k of (observer q, star s) uses Philox block (k, 2, q * n_stars + s) of the seed.
// ---------------------------------------------------------------------------
// optical_length_tot (optical_depth.f90:248-324): optical depth from (x,y,z) to the edge of the grid along (u,v,w)
template <bool L3D>
__device__ inline float optical_length_tot(const Lds& T, const DevModel& M, int lambda, double x, double y, double z,
                                           double u, double v, double w) {
  const int n_rad = M.n_rad, nz = M.nz;
  int ri, zj, k;
  const bool sph = M.grid_sph != 0;
  if (sph) index_cell_sph<L3D>(T, M, x, y, z, ri, zj, k);
  else index_cell<L3D>(T, M, x, y, z, ri, zj, k);
  const double a = u * u + v * v;
  const double inv_a = (a > TINY_REAL) ? 1.0 / a : HUGE_REAL;
  const double inv_w = (fabs(w) > TINY_REAL) ? 1.0 / w : copysign(HUGE_DP, w);
  double tau = 0.0;
  for (long guard = 0; guard < 100000000L; ++guard) {
    const int azj = zj < 0 ? -zj : zj;
    if ((ri == n_rad + 1) || (!sph && (azj == nz + 1) && (fabs(z) > M.zmaxmax))) break;  // test_exit_grid
    double x1, y1, z1, l;
    int ri1, zj1, k1;
    if (sph) cross_cell_sph<L3D>(T, M, x, y, z, u, v, w, ri, zj, k, x1, y1, z1, ri1, zj1, k1, l);
    else MCGPU_CROSS<L3D>(T, M, x, y, z, u, v, w, inv_a, inv_w, ri, zj, k, x1, y1, z1, ri1, zj1, k1, l);
    if (is_real_cell<L3D>(n_rad, nz, ri, zj)) {
      const int ic = cell_index<L3D>(n_rad, nz, ri, zj, k);
      const double kap = M.n_classes ? M.v_kappa[(size_t)M.cell_class[ic] * M.n_lambda + (lambda - 1)] : T.kappa[lambda - 1];
      tau += l * (kap * M.kappa_factor[ic]);
    }
    x = x1; y = y1; z = z1;
    ri = ri1; zj = zj1; k = k1;
  }
  return (float)tau;  // tau_tot_out is a default real
}

// ... on a Voronoi grid: index_cell_voronoi (the nearest site), then cross_Voronoi_cell to the box
__device__ inline float optical_length_tot_voro(const Lds& T, const DevModel& M, const VoroGrid& G, int lambda, double x, double y,
                                                double z, double u, double v, double w) {
  int next = voro_index_cell(G, x, y, z), icell0 = 0, prev = 0;
  double tau = 0.0;
  for (long guard = 0; guard < 100000000L; ++guard) {
    prev = icell0;
    icell0 = next;
    if (icell0 < 0) break;  // test_exit_grid
    const VoroCell C = G.cell[icell0 - 1];
    double opacity = 0.0;
    if (icell0 <= M.n_cells) {
      const double kap = M.n_classes ? M.v_kappa[(size_t)M.cell_class[icell0 - 1] * M.n_lambda + (lambda - 1)] : T.kappa[lambda - 1];
      opacity = kap * C.kf;
    }
    double x1, y1, z1, l, l_contrib, l_void;
    voro_cross_cell(G, M, C, x, y, z, u, v, w, icell0, prev, x1, y1, z1, next, l, l_contrib, l_void);
    tau += l_contrib * opacity;
    x = x1; y = y1; z = z1;
  }
  return (float)tau;
}

constexpr int STARS_NX_SCREEN = 10, STARS_N_RAY_SED = 1024;  // dust_transfer.f90:1615,1627

template <bool L3D>
__global__ void __launch_bounds__(512) k_stars_map_sed(const DevModel M, const RtArgs A, unsigned int key0, unsigned int key1,
                                                       const double* star_flux, double* out) {
  extern __shared__ double lds_raw[];
  const Lds T = lds_carve(lds_raw, M, true);  // geometry, kappa, albedo: the SED-mode table set
  lds_stage_mono(T, M, 1);
  __shared__ float tau_screen[(2 * STARS_NX_SCREEN + 1) * (2 * STARS_NX_SCREEN + 1)];
  __shared__ double red_a[512], red_b[512];
  __syncthreads();
  const int q = blockIdx.x / M.n_stars, istar = blockIdx.x % M.n_stars;
  const int tid = threadIdx.x, nt = blockDim.x, ns = 2 * STARS_NX_SCREEN + 1;
  double uvw[3], xpi[3], ypi[3];
  rt_image_plane(A, q, uvw, xpi, ypi);
  const double* s4 = &M.star_xyzr[4 * istar];
  const double delta = s4[3] / (double)STARS_NX_SCREEN;
  const double nx = sqrt(xpi[0] * xpi[0] + xpi[1] * xpi[1] + xpi[2] * xpi[2]);
  const double ny = sqrt(ypi[0] * ypi[0] + ypi[1] * ypi[1] + ypi[2] * ypi[2]);
  const double dxs[3] = {delta * xpi[0] / nx, delta * xpi[1] / nx, delta * xpi[2] / nx};
  const double dys[3] = {delta * ypi[0] / ny, delta * ypi[1] / ny, delta * ypi[2] / ny};
  for (int p = tid; p < ns * ns; p += nt) {
    const int i = p % ns - STARS_NX_SCREEN, j = p / ns - STARS_NX_SCREEN;
    const double x = s4[0] + dxs[0] * i + dys[0] * j, y = s4[1] + dxs[1] * i + dys[1] * j, z = s4[2] + dxs[2] * i + dys[2] * j;
    tau_screen[p] = A.voro ? optical_length_tot_voro(T, M, *A.voro, A.lambda, x, y, z, uvw[0], uvw[1], uvw[2])
                           : optical_length_tot<L3D>(T, M, A.lambda, x, y, z, uvw[0], uvw[1], uvw[2]);
  }
  __syncthreads();
  const int n_ray = STARS_N_RAY_SED / M.n_stars > 1 ? STARS_N_RAY_SED / M.n_stars : 1;
  const double norm_screen2 = 1.0 / (delta * delta);
  double sum_f = 0.0, sum_n = 0.0;
  for (int iray = tid; iray < n_ray; iray += nt) {
    uint32_t o[4];
    philox4x32_10((uint32_t)iray, 2u, (uint32_t)blockIdx.x, 0u, key0, key1, o);
    const float rand = Rng::real(o[0]), rand2 = Rng::real(o[1]);
    const double z = 2.0 * (double)rand - 1.0;
    const double srw02 = sqrt(1.0 - z * z), argmt = PI * (2.0 * (double)rand2 - 1.0);
    double sa, ca;
    sincos(argmt, &sa, &ca);
    const double x = srw02 * ca, y = srw02 * sa;
    const float cos_thet = (float)fabs(x * uvw[0] + y * uvw[1] + z * uvw[2]);
    const double vec[3] = {x * s4[3], y * s4[3], z * s4[3]};
    const double offset_x = (vec[0] * dxs[0] + vec[1] * dxs[1] + vec[2] * dxs[2]) * norm_screen2;
    const double offset_y = (vec[0] * dys[0] + vec[1] * dys[1] + vec[2] * dys[2]) * norm_screen2;
    const int i = (int)floor(offset_x), j = (int)floor(offset_y);
    const double fx = offset_x - i, fy = offset_y - j;
    float tau = 0.0f;
    if (i >= -STARS_NX_SCREEN && i < STARS_NX_SCREEN && j >= -STARS_NX_SCREEN && j < STARS_NX_SCREEN) {
      const int p = (i + STARS_NX_SCREEN) + ns * (j + STARS_NX_SCREEN);
      tau = (float)((double)tau_screen[p] * (1 - fx) * (1 - fy) + (double)tau_screen[p + 1] * fx * (1 - fy) +
                    (double)tau_screen[p + ns] * (1 - fx) * fy + (double)tau_screen[p + ns + 1] * fx * fy);
    }
    sum_f += (double)(expf(-tau) * cos_thet);  // exp(-tau) * cos_thet * LimbDarkening in default real
    sum_n += (double)cos_thet;
  }
  red_a[tid] = sum_f; red_b[tid] = sum_n;
  __syncthreads();
  for (int sft = nt >> 1; sft > 0; sft >>= 1) {
    if (tid < sft) { red_a[tid] += red_a[tid + sft]; red_b[tid] += red_b[tid + sft]; }
    __syncthreads();
  }
  if (tid == 0) atomic_add_f64(&out[q], star_flux[istar] * red_a[0] / red_b[0]);
}

// interp (utils.f90:130-175, default real): linear interpolation in a table, the end values outside it
__device__ inline float interp_sp(const float* y, const float* x, int n, float xp) {
  float xmin = x[0], xmax = x[0];
  for (int i = 1; i < n; ++i) { xmin = fminf(xmin, x[i]); xmax = fmaxf(xmax, x[i]); }
  const bool inc = x[n - 1] > x[0];
  if (xp < xmin) return inc ? y[0] : y[n - 1];
  if (xp > xmax) return inc ? y[n - 1] : y[0];
  int j;
  if (inc) { for (j = 2; j <= n - 1; ++j) if (x[j - 1] > xp) break; }
  else { for (j = 2; j <= n - 1; ++j) if (x[j - 1] < xp) break; }
  const float frac = (xp - x[j - 2]) / (x[j - 1] - x[j - 2]);
  return y[j - 2] * (1.f - frac) + y[j - 1] * frac;
}

// compute_stars_map for images (dust_transfer.f90:1604-1854 with lresolved = .true.): the stars' discs in the pixel map
// of every observer, limb-darkened and polarised if asked.  One workgroup per (observer, star) like the SED version: the
// 21 x 21 screen of optical depths, then n_ray random points of the stellar sphere (1024 / n_stars, or 100 per pixel of
// the disc when the star is wider than a pixel, :1655-1667), each placed in its pixel (find_pixel, :1858-1893) with
// weight exp(-tau) cos_thet LimbDarkening; the workgroup's map is normalised by sum(cos_thet LimbDarkening) in a second
// pass (the rays are replayed: counter-based draws), so that the atomics add finished values.
struct StarsImageArgs {
  int npix_x, npix_y, n_maps, n_mu;
  double taille_pix, distance;
  float pix_size;
  const float *mu_ld, *ld, *pola_ld;
  double* map;            // (npix_x, npix_y, n_maps, nRT)
  double* star_position;  // (n_stars, nRT, 2) or null
};

template <bool L3D>
__global__ void __launch_bounds__(512) k_stars_map_image(const DevModel M, const RtArgs A, const StarsImageArgs I, unsigned int key0,
                                                         unsigned int key1, const double* star_flux) {
  extern __shared__ double lds_raw[];
  const Lds T = lds_carve(lds_raw, M, true);
  lds_stage_mono(T, M, 1);
  __shared__ float tau_screen[(2 * STARS_NX_SCREEN + 1) * (2 * STARS_NX_SCREEN + 1)];
  __shared__ double red_b[512];
  __syncthreads();
  const int q = blockIdx.x / M.n_stars, istar = blockIdx.x % M.n_stars;
  const int tid = threadIdx.x, nt = blockDim.x, ns = 2 * STARS_NX_SCREEN + 1;
  double uvw[3], xpi[3], ypi[3];
  rt_image_plane(A, q, uvw, xpi, ypi);
  const double* s4 = &M.star_xyzr[4 * istar];
  const double dx_map[3] = {xpi[0] * I.taille_pix, xpi[1] * I.taille_pix, xpi[2] * I.taille_pix};
  const double dy_map[3] = {ypi[0] * I.taille_pix, ypi[1] * I.taille_pix, ypi[2] * I.taille_pix};
  const double delta = s4[3] / (double)STARS_NX_SCREEN;
  const double nx = sqrt(xpi[0] * xpi[0] + xpi[1] * xpi[1] + xpi[2] * xpi[2]);
  const double ny = sqrt(ypi[0] * ypi[0] + ypi[1] * ypi[1] + ypi[2] * ypi[2]);
  const double dxs[3] = {delta * xpi[0] / nx, delta * xpi[1] / nx, delta * xpi[2] / nx};
  const double dys[3] = {delta * ypi[0] / ny, delta * ypi[1] / ny, delta * ypi[2] / ny};
  for (int p = tid; p < ns * ns; p += nt) {
    const int i = p % ns - STARS_NX_SCREEN, j = p / ns - STARS_NX_SCREEN;
    const double x = s4[0] + dxs[0] * i + dys[0] * j, y = s4[1] + dxs[1] * i + dys[1] * j, z = s4[2] + dxs[2] * i + dys[2] * j;
    tau_screen[p] = A.voro ? optical_length_tot_voro(T, M, *A.voro, A.lambda, x, y, z, uvw[0], uvw[1], uvw[2])
                           : optical_length_tot<L3D>(T, M, A.lambda, x, y, z, uvw[0], uvw[1], uvw[2]);
  }
  __syncthreads();
  int n_ray = STARS_N_RAY_SED / M.n_stars > 1 ? STARS_N_RAY_SED / M.n_stars : 1;
  if (2.0 * s4[3] > (double)I.pix_size) {
    const float ratio = (float)(s4[3] / (double)I.pix_size);
    const int n_res = 100 * (int)(4.0 * PI * (double)(ratio * ratio));
    n_ray = n_res > STARS_N_RAY_SED ? n_res : STARS_N_RAY_SED;
  }
  const double norm_screen2 = 1.0 / (delta * delta);
  const size_t n_pix = (size_t)I.npix_x * I.npix_y;
  double* mp = I.map + n_pix * I.n_maps * (size_t)q;
  const int x_center = I.npix_x / 2 + 1, y_center = I.npix_y / 2 + 1;
  double factor2 = 0.0;
  for (int pass = 0; pass < 2; ++pass) {
    double sum_n = 0.0;
    for (int iray = tid; iray < n_ray; iray += nt) {
      uint32_t o[4];
      philox4x32_10((uint32_t)iray, 2u, (uint32_t)blockIdx.x, 0u, key0, key1, o);
      const float rand = Rng::real(o[0]), rand2 = Rng::real(o[1]);
      const double z = 2.0 * (double)rand - 1.0;
      const double srw02 = sqrt(1.0 - z * z), argmt = PI * (2.0 * (double)rand2 - 1.0);
      double sa, ca;
      sincos(argmt, &sa, &ca);
      const double x = srw02 * ca, y = srw02 * sa;
      const float cos_thet = (float)fabs(x * uvw[0] + y * uvw[1] + z * uvw[2]);
      float LimbDarkening = 1.0f, Pola_LD = 0.0f;
      if (I.n_mu > 0) {
        LimbDarkening = interp_sp(I.ld, I.mu_ld, I.n_mu, cos_thet);
        if (I.pola_ld) Pola_LD = interp_sp(I.pola_ld, I.mu_ld, I.n_mu, cos_thet);
      }
      if (pass == 0) { sum_n += (double)(cos_thet * LimbDarkening); continue; }
      const double vec[3] = {x * s4[3], y * s4[3], z * s4[3]};
      const double px = s4[0] + vec[0], py = s4[1] + vec[1], pz = s4[2] + vec[2];
      const double offset_x = (vec[0] * dxs[0] + vec[1] * dxs[1] + vec[2] * dxs[2]) * norm_screen2;
      const double offset_y = (vec[0] * dys[0] + vec[1] * dys[1] + vec[2] * dys[2]) * norm_screen2;
      const int i = (int)floor(offset_x), j = (int)floor(offset_y);
      const double fx = offset_x - i, fy = offset_y - j;
      float tau = 0.0f;
      if (i >= -STARS_NX_SCREEN && i < STARS_NX_SCREEN && j >= -STARS_NX_SCREEN && j < STARS_NX_SCREEN) {
        const int p = (i + STARS_NX_SCREEN) + ns * (j + STARS_NX_SCREEN);
        tau = (float)((double)tau_screen[p] * (1 - fx) * (1 - fy) + (double)tau_screen[p + 1] * fx * (1 - fy) +
                      (double)tau_screen[p + ns] * (1 - fx) * fy + (double)tau_screen[p + ns + 1] * fx * fy);
      }
      const double factor = 1.0 / (I.taille_pix * I.taille_pix);  // find_pixel
      const double x_map = (px * dx_map[0] + py * dx_map[1] + pz * dx_map[2]) * factor;
      const double y_map = (px * dy_map[0] + py * dy_map[1] + pz * dy_map[2]) * factor;
      const int ip = (I.npix_x % 2 == 1) ? (int)llround(x_map) + I.npix_x / 2 + 1 : (int)llround(x_map + 0.5) + I.npix_x / 2;
      const int jp = (I.npix_y % 2 == 1) ? (int)llround(y_map) + I.npix_y / 2 + 1 : (int)llround(y_map + 0.5) + I.npix_y / 2;
      if (ip >= 1 && ip <= I.npix_x && jp >= 1 && jp <= I.npix_y) {
        const float wgt = expf(-tau) * cos_thet * LimbDarkening;
        const size_t pp = (size_t)(ip - 1) + (size_t)I.npix_x * (jp - 1);
        atomic_add_f64(&mp[pp], (double)wgt * factor2);
        if (I.n_maps == 3) {
          const float P = wgt * Pola_LD;
          const float phi = atan2f((float)(jp - y_center) * 1.0f, (float)(ip - x_center) * 1.0f);
          atomic_add_f64(&mp[pp + n_pix], (double)(P * cosf(2.0f * phi)) * factor2);
          atomic_add_f64(&mp[pp + 2 * n_pix], (double)(P * sinf(2.0f * phi)) * factor2);
        }
      }
    }
    if (pass == 0) {
      red_b[tid] = sum_n;
      __syncthreads();
      for (int sft = nt >> 1; sft > 0; sft >>= 1) {
        if (tid < sft) red_b[tid] += red_b[tid + sft];
        __syncthreads();
      }
      factor2 = star_flux[istar] / red_b[0];
      __syncthreads();
    }
  }
  if (tid == 0 && I.star_position) {
    const double factor_pix = 1.0 / (I.taille_pix * I.distance);
    I.star_position[(size_t)istar + (size_t)M.n_stars * q] = -(s4[0] * dx_map[0] + s4[1] * dx_map[1] + s4[2] * dx_map[2]) * factor_pix;
    I.star_position[(size_t)istar + (size_t)M.n_stars * (q + (size_t)A.nRT)] =
        (s4[0] * dy_map[0] + s4[1] * dy_map[1] + s4[2] * dy_map[2]) * factor_pix;
  }
}

// ---------------------------------------------------------------------------
// Optical-depth maps (options -tau_map / -tau_surface): compute_tau_map (dust_transfer.f90:2114-2210) and
// compute_tau_surface_map (:2006-2110).  One thread per pixel CENTRE of an observer's image: move_to_grid backwards from
// 10 Rmax, then
//   tau_map          optical_length_tot (optical_depth.f90:248-324) from the entry point to the far edge of the grid
//   tau_surface_map  physical_length (:21-182) until the optical depth tau_surface is used up: the point reached, or zeros
//                    when the ray leaves the grid or ends on a star first; a cell of the dark zone hands back the entry
//                    point of the cell before it (the mirror of :104-112).  Nothing is deposited (the reference's call
//                    also runs save_radiation_field: a side effect of reusing the packets' routine, not reproduced).
// Outputs are default reals in the reference's layouts (npix_x, npix_y, RT_n_incl, RT_n_az[, 3]); either may be null.
// ---------------------------------------------------------------------------
__device__ inline void tau_maps_pixel(const RtArgs& A, long pix, int& q, double pc[3], double uvw[3]) {
  const long per_dir = (long)A.npix_x * A.npix_y;
  q = (int)(pix / per_dir);
  const long rem = pix - (long)q * per_dir;
  const int j = (int)(rem / A.npix_x) + 1, i = (int)(rem - (long)(j - 1) * A.npix_x) + 1;
  double xpi[3], ypi[3];
  rt_image_plane(A, q, uvw, xpi, ypi);
  for (int c = 0; c < 3; ++c) {
    const double dx = xpi[c] * A.taille_pix, dy = ypi[c] * A.taille_pix;
    const double Icorner = uvw[c] * A.l_far - (0.5 * A.npix_x * dx + 0.5 * A.npix_y * dy);
    pc[c] = Icorner + (i - 0.5) * dx + (j - 0.5) * dy;
  }
}

template <bool L3D>
__global__ void __launch_bounds__(256) k_tau_maps(const DevModel M, const RtArgs A, float tau_surface, float* tau_map,
                                                  float* surf_map) {
  extern __shared__ double lds_raw[];
  const Lds T = lds_carve(lds_raw, M, true);
  lds_stage_mono(T, M, 1);
  __syncthreads();
  const int n_rad = M.n_rad, nz = M.nz;
  const bool sph = M.grid_sph != 0;
  const long n_pix = (long)A.npix_x * A.npix_y * A.nRT;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < n_pix; pix += (long)gridDim.x * blockDim.x) {
    int q;
    double pc[3], uvw[3];
    tau_maps_pixel(A, pix, q, pc, uvw);
    const double u = -uvw[0], v = -uvw[1], w = -uvw[2];  // reverse propagation
    double x = pc[0], y = pc[1], z = pc[2];
    int ri, zj, k;
    const bool hit = sph ? move_to_grid_sph<L3D>(T, M, x, y, z, u, v, w, ri, zj, k) : move_to_grid<L3D>(T, M, x, y, z, u, v, w, ri, zj, k);
    const double a = u * u + v * v;
    const double inv_a = (a > TINY_REAL) ? 1.0 / a : HUGE_REAL;
    const double inv_w = (fabs(w) > TINY_REAL) ? 1.0 / w : copysign(HUGE_DP, w);
    if (tau_map) {
      double tau = 0.0, xa = x, ya = y, za = z;
      int r = ri, j = zj, kk = k;
      for (long guard = 0; hit && guard < 100000000L; ++guard) {
        const int azj = j < 0 ? -j : j;
        if ((r == n_rad + 1) || (!sph && (azj == nz + 1) && (fabs(za) > M.zmaxmax))) break;  // test_exit_grid
        double x1, y1, z1, l;
        int r1, j1, k1;
        if (sph) cross_cell_sph<L3D>(T, M, xa, ya, za, u, v, w, r, j, kk, x1, y1, z1, r1, j1, k1, l);
        else MCGPU_CROSS<L3D>(T, M, xa, ya, za, u, v, w, inv_a, inv_w, r, j, kk, x1, y1, z1, r1, j1, k1, l);
        if (is_real_cell<L3D>(n_rad, nz, r, j)) {
          const int ic = cell_index<L3D>(n_rad, nz, r, j, kk);
          const double kap = M.n_classes ? M.v_kappa[(size_t)M.cell_class[ic] * M.n_lambda + (A.lambda - 1)] : T.kappa[A.lambda - 1];
          tau += l * (kap * M.kappa_factor[ic]);
        }
        xa = x1; ya = y1; za = z1;
        r = r1; j = j1; kk = k1;
      }
      tau_map[pix] = (float)tau;
    }
    if (surf_map) {
      float out[3] = {0.0f, 0.0f, 0.0f};
      if (hit) {
        const int i_star = intersect_stars(M, x, y, z, u, v, w);
        int star_key = -1;
        if (i_star > 0) {
          const int* sc = &M.star_cell[4 * (i_star - 1)];
          star_key = sc[0] + (n_rad + 2) * ((sc[1] + nz + 1) + (2 * nz + 3) * (sc[2] - 1));
        }
        double extr = (double)tau_surface, xo = x, yo = y, zo = z;  // (xo: entry point of the cell before the current one)
        for (long guard = 0; guard < 100000000L; ++guard) {
          const int azj = zj < 0 ? -zj : zj;
          if ((ri == n_rad + 1) || (!sph && (azj == nz + 1) && (fabs(z) > M.zmaxmax))) break;  // flag_sortie
          if (star_key >= 0 && (ri + (n_rad + 2) * ((zj + nz + 1) + (2 * nz + 3) * (k - 1))) == star_key) break;
          double opacity = 0.0;
          if (is_real_cell<L3D>(n_rad, nz, ri, zj)) {
            const int ic = cell_index<L3D>(n_rad, nz, ri, zj, k);
            if (M.dark && M.dark[ic]) { out[0] = (float)xo; out[1] = (float)yo; out[2] = (float)zo; break; }
            const double kap = M.n_classes ? M.v_kappa[(size_t)M.cell_class[ic] * M.n_lambda + (A.lambda - 1)] : T.kappa[A.lambda - 1];
            opacity = kap * M.kappa_factor[ic];
          }
          double x1, y1, z1, l;
          int ri1, zj1, k1;
          if (sph) cross_cell_sph<L3D>(T, M, x, y, z, u, v, w, ri, zj, k, x1, y1, z1, ri1, zj1, k1, l);
          else MCGPU_CROSS<L3D>(T, M, x, y, z, u, v, w, inv_a, inv_w, ri, zj, k, x1, y1, z1, ri1, zj1, k1, l);
          const double tau = l * opacity;
          if (tau > extr) {
            const double ls = l * (extr / tau);
            out[0] = (float)(x + ls * u); out[1] = (float)(y + ls * v); out[2] = (float)(z + ls * w);
            break;
          }
          extr = extr - tau;
          xo = x; yo = y; zo = z;
          x = x1; y = y1; z = z1;
          ri = ri1; zj = zj1; k = k1;
        }
      }
      surf_map[pix] = out[0]; surf_map[pix + n_pix] = out[1]; surf_map[pix + 2 * n_pix] = out[2];
    }
  }
}

// ---------------------------------------------------------------------------
// define_dark_zone, step 4 (optical_depth.f90:1522-1551; 2D): from the centre of every candidate cell, 11 rays in the
// (x, z) plane at angles pi n / 12; a ray that uses up the optical depth tau_max before it leaves the grid marks its
// cell.  One ray per thread: physical_length (optical_depth.f90:21-178) without deposits (Stokes = 0).
// flag[icell] = 1 when some ray of the cell does not leave.
// The reference decides the columns one after the other (i ascending) and sets l_dark_zone as it goes; physical_length
// reads those flags, so a ray that enters a cell of a column decided EARLIER is mirrored there and counts as "does not
// leave" (:104-112 -> flag_sortie = .false.).  dark_now[n_cells] (or null) = the flags of a previous pass; only the cells
// of columns before the ray's own are looked at, which is what the sequential loop has set when it tests this column.  The
// caller repeats the pass until the flags stop changing: column i is final after i - i_lo + 1 passes at the latest, in
// practice after one or two (mcgpu_define_dark_zone).
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) k_dark_zone_rays(const DevModel M, int lambda, float tau_max, int i_lo, int i_hi,
                                                        const int* zj_sup, const double* r_grid, const double* z_grid,
                                                        const unsigned char* dark_now, unsigned char* flag) {
  extern __shared__ double lds_raw[];
  const Lds T = lds_carve(lds_raw, M, true);
  lds_stage_mono(T, M, 1);
  __syncthreads();
  const int n_rad = M.n_rad, nz = M.nz;
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int n = (int)(tid % 11) + 1;
  const long long c = tid / 11;
  const int i = i_lo + (int)(c % (i_hi - i_lo + 1)), j = 1 + (int)(c / (i_hi - i_lo + 1));
  if (i > i_hi || j > nz || j > zj_sup[i - 1]) return;
  const int icell = (i - 1) + n_rad * (j - 1);
  const float angle = (float)(PI * (double)((float)n / 12.0f));  // pi * real(n)/real(nbre_angle+1)
  double x = r_grid[icell], y = 0.0, z = z_grid[icell];
  const double u = (double)cosf(angle), v = 0.0, w = (double)sinf(angle);
  int ri = i, zj = j, k = 1;
  const double a = u * u + v * v;
  const double inv_a = (a > TINY_REAL) ? 1.0 / a : HUGE_REAL;
  const double inv_w = (fabs(w) > TINY_REAL) ? 1.0 / w : copysign(HUGE_DP, w);
  const int i_star = intersect_stars(M, x, y, z, u, v, w);
  int star_key = -1;
  if (i_star > 0) {
    const int* sc = &M.star_cell[4 * (i_star - 1)];
    star_key = sc[0] + (n_rad + 2) * ((sc[1] + nz + 1) + (2 * nz + 3) * (sc[2] - 1));
  }
  double extr = (double)tau_max;
  for (long guard = 0; guard < 100000000L; ++guard) {
    const int azj = zj < 0 ? -zj : zj;
    if ((ri == n_rad + 1) || ((azj == nz + 1) && (fabs(z) > M.zmaxmax))) return;  // leaves the grid
    if (star_key >= 0 && (ri + (n_rad + 2) * ((zj + nz + 1) + (2 * nz + 3) * (k - 1))) == star_key) return;  // (:91-97: flag_sortie)
    double x1, y1, z1, l;
    int ri1, zj1, k1;
    MCGPU_CROSS<false>(T, M, x, y, z, u, v, w, inv_a, inv_w, ri, zj, k, x1, y1, z1, ri1, zj1, k1, l);
    double opacity = 0.0;
    if (is_real_cell<false>(n_rad, nz, ri, zj)) {
      const int ic = cell_index<false>(n_rad, nz, ri, zj, k);
      if (dark_now && ri < i && dark_now[ic]) { flag[icell] = 1; return; }  // mirrored in a column decided earlier
      const double kap = M.n_classes ? M.v_kappa[(size_t)M.cell_class[ic] * M.n_lambda + (lambda - 1)] : T.kappa[lambda - 1];
      opacity = kap * M.kappa_factor[ic];
    }
    const double tau = l * opacity;
    if (tau > extr) { flag[icell] = 1; return; }  // the ray stops inside: the cell is dark
    extr = extr - tau;
    x = x1; y = y1; z = z1;
    ri = ri1; zj = zj1; k = k1;
  }
}

}  // namespace mcgpu
