// Ray tracing method 2, the source function of one inclination: init_dust_source_fct2 (dust_ray_tracing.f90:717-806) =
// calc_Isca_rt2_star (:1245-1440, with angles_scatt_rt2 :304-405) + calc_Isca_rt2 (:907-1240) + calc_Jth + the
// division by kappa_ext and the (Q, U) -> (P, angle) form.  The packet loop has left I_spec / I_spec_star in HBM
// (mc_mono.hip.h, device layout [cell][phi_I][theta_I][8] of doubles); these kernels turn them into eps_dust2
// (N_type_flux, nang_ray_tracing, 0:1, n_cells) and eps_dust2_star (n_Stokes, nang_ray_tracing_star, 0:1, n_cells),
// default real like the reference's arrays -- the "slow lines" of the reference's method-2 ray tracer (its comments:
// 1.6 s + 0.7 s of 2.4 s per inclination).  2D grids.
// The bin -> direction tables (tab_k, tab_sin_scatt_norm, tab_cosw, tab_sinw: "many dimensions but small numbers") are
// built by the host side of the C-ABI exactly as :973-1072 and uploaded.  The per-cell sums keep the reference's order
// (phi_I outer, theta_I inner) and its default-real accumulator, so the result equals the CPU restatement's up to the
// last place of sin / cos / atan2f / sqrtf.  The contraction over the 225 direction bins is GEMM-shaped; it is not put on
// the matrix cores for that reason (and at 7000 cells x 30 directions x 225 bins it is microseconds of work).
#pragma once
#include "mc_device.hip.h"
#include "mc_mono.hip.h"

namespace mcgpu {

constexpr int RT2_N_SUPER = 5, RT2_NSUP2 = RT2_N_SUPER * RT2_N_SUPER;

struct Rt2Args {
  int lambda, p_lambda, n_theta_I, n_phi_I, nang_rt, nang_star, n_Stokes, N_type_flux, contrib, pola;
  double photon_energy, uv0, w0;
  const double* I_spec;       // device layout [cell][phi_I][theta_I][XI_LINE]
  const double* I_spec_star;  // [cell]
  const double* J_th;         // [cell]
  const double *r_grid, *z_grid;
  const int* tab_k;           // [dir][iscatt][phi_I][theta_I][i2][i1]
  const float* tab_sin;       // the same: tab_sin_scatt_norm
  const double *tab_cosw, *tab_sinw;  // [dir][iscatt][phi_I][theta_I]
  const float* s11_single;    // tab_s11_pos(0:nang, p_lambda) of the single class
  float* eps_dust2;           // reference layout (N_type_flux, nang_rt, 0:1, n_cells)
  float* eps_dust2_star;      // (n_Stokes, nang_star, 0:1, n_cells)
};

// the Mueller columns of a cell (its class with lvariable_dust), column p_lambda
struct Rt2Cols { const float *s11, *s12, *s22, *s33, *s34, *s44; };
__device__ inline Rt2Cols rt2_cols(const DevModel& M, const Rt2Args& A, int ic) {
  Rt2Cols C;
  const size_t na1 = (size_t)M.nang + 1;
  if (M.n_classes) {
    const size_t col = ((size_t)M.cell_class[ic] * M.n_lambda + (A.p_lambda - 1)) * na1;
    C.s11 = M.v_s11 + col; C.s12 = M.v_s12 + col; C.s22 = M.v_s22 + col; C.s33 = M.v_s33 + col; C.s34 = M.v_s34 + col; C.s44 = M.v_s44 + col;
  } else {
    const size_t col = na1 * (A.p_lambda - 1);
    C.s11 = A.s11_single; C.s12 = M.s12 + col; C.s22 = M.s22 + col; C.s33 = M.s33 + col; C.s34 = M.s34 + col; C.s44 = M.s44 + col;
  }
  return C;
}

__device__ inline void rt2_opacities(const DevModel& M, int lambda, int ic, double& kappa_ext, double& kappa_sca) {
  const size_t row = M.n_classes ? (size_t)M.cell_class[ic] * M.n_lambda + (lambda - 1) : 0;
  const double kap = M.n_classes ? M.v_kappa[row] : M.kappa[lambda - 1];
  const float alb = M.n_classes ? M.v_albedo[row] : M.albedo[lambda - 1];
  kappa_ext = kap * M.kappa_factor[ic];
  kappa_sca = kap * M.kappa_factor[ic] * (double)alb;
}

// calc_Isca_rt2 + the non-stellar part of init_dust_source_fct2: one thread per (iscatt, dir, cell)
template <bool POLA>
__global__ void k_rt2_source(const DevModel M, const Rt2Args A) {
#pragma clang fp contract(off)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)M.n_cells * 2 * A.nang_rt;
  if (i >= n) return;
  const int iscatt = (int)(i % A.nang_rt), dir = (int)((i / A.nang_rt) % 2), ic = (int)(i / ((size_t)2 * A.nang_rt));
  const Rt2Cols C = rt2_cols(M, A, ic);
  const int ntf = A.N_type_flux, n_Stokes = A.n_Stokes;
  double kappa_ext, kappa_sca;
  rt2_opacities(M, A.lambda, ic, kappa_ext, kappa_sca);
  const double factor = A.photon_energy / M.volume[ic];
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int phi_I = 0; phi_I < A.n_phi_I; ++phi_I)
    for (int theta_I = 0; theta_I < A.n_theta_I; ++theta_I) {
      const size_t b = (((size_t)dir * A.nang_rt + iscatt) * A.n_phi_I + phi_I) * A.n_theta_I + theta_I;
      const int* tk = A.tab_k + b * RT2_NSUP2;
      const float* ts = A.tab_sin + b * RT2_NSUP2;
      float s11 = 0.f;
      for (int t = 0; t < RT2_NSUP2; ++t) s11 = s11 + C.s11[tk[t]] * ts[t];
      const double* st = A.I_spec + (((size_t)ic * A.n_phi_I + phi_I) * A.n_theta_I + theta_I) * XI_LINE;
      if (POLA) {
        const int k = tk[(RT2_N_SUPER / 2) + RT2_N_SUPER * (RT2_N_SUPER / 2)];
        const double cosw = A.tab_cosw[b], sinw = A.tab_sinw[b];
        const float s12 = -s11 * C.s12[k], s22 = s11 * C.s22[k], s33 = -s11 * C.s33[k], s34 = -s11 * C.s34[k], s44 = -s11 * C.s44[k];
        const double C1 = st[0], C4 = st[3];
        const double C2 = cosw * st[1] + (-1.0 * sinw) * st[2];
        const double C3 = sinw * st[1] + cosw * st[2];
        const double D1 = (double)s11 * C1 + (double)s12 * C2;
        const double D2 = (double)s12 * C1 + (double)s22 * C2;
        const double D3 = (double)s33 * C3 + (double)(-s34) * C4;
        const double D4 = (double)s34 * C3 + (double)s44 * C4;
        const double S2 = cosw * D2 + sinw * D3;
        const double S3 = -((-1.0 * sinw) * D2 + cosw * D3);
        acc[0] = (float)((double)acc[0] + D1);
        acc[1] = (float)((double)acc[1] + S2);
        acc[2] = (float)((double)acc[2] + S3);
        acc[3] = (float)((double)acc[3] + D4);
      } else {
        acc[0] = (float)((double)acc[0] + (double)s11 * st[0]);
      }
      if (A.contrib) {
        acc[n_Stokes + 1] = (float)((double)acc[n_Stokes + 1] + (double)s11 * st[n_Stokes + 1]);
        acc[n_Stokes + 3] = (float)((double)acc[n_Stokes + 3] + (double)s11 * st[n_Stokes + 3]);
      }
    }
  float* e = A.eps_dust2 + (size_t)ntf * ((size_t)iscatt + (size_t)A.nang_rt * (dir + 2 * (size_t)ic));
  if (!(kappa_ext > TINY_DP)) {
    for (int t = 0; t < ntf; ++t) e[t] = 0.f;
    return;
  }
  float I2[8];
  for (int t = 0; t < 8; ++t) I2[t] = (float)(((double)acc[t] * factor) * kappa_sca);  // I_sca2 (:1228)
  const double jth = A.J_th[ic];
  e[0] = (float)(((double)I2[0] + jth) / kappa_ext);
  if (POLA) {
    const float Q = (float)((double)I2[1] / kappa_ext), U = (float)((double)I2[2] / kappa_ext);
    e[1] = sqrtf(Q * Q + U * U);
    e[2] = atan2f(U, Q);
    e[3] = (float)((double)I2[3] / kappa_ext);
  }
  if (A.contrib) {
    e[n_Stokes] = 0.f;
    e[n_Stokes + 1] = (float)((double)I2[n_Stokes + 1] / kappa_ext);
    e[n_Stokes + 2] = (float)(jth / kappa_ext);
    e[n_Stokes + 3] = (float)((double)I2[n_Stokes + 3] / kappa_ext);
  }
}

// nint(acos(cos_scatt) * real(nang_scatt) / pi), the correctly rounded default-real acos
__device__ inline int rt2_angle_index(float cos_scatt, int nang) {
  const float ac = (float)acos((double)cos_scatt);
  if (ac != ac) return nang;
  return (int)llrint(floor((double)(ac * (float)nang) / PI + 0.5));
}

// calc_Isca_rt2_star + its part of init_dust_source_fct2: one thread per (iscatt, dir, cell)
template <bool POLA>
__global__ void k_rt2_source_star(const DevModel M, const Rt2Args A) {
#pragma clang fp contract(off)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)M.n_cells * 2 * A.nang_star;
  if (i >= n) return;
  const int iscatt = (int)(i % A.nang_star) + 1, dir = (int)((i / A.nang_star) % 2), ic = (int)(i / ((size_t)2 * A.nang_star));
  float* e = A.eps_dust2_star + (size_t)A.n_Stokes * ((size_t)(iscatt - 1) + (size_t)A.nang_star * (dir + 2 * (size_t)ic));
  for (int t = 0; t < A.n_Stokes; ++t) e[t] = 0.f;
  const double Istar = A.I_spec_star[ic];
  double kappa_ext, kappa_sca;
  rt2_opacities(M, A.lambda, ic, kappa_ext, kappa_sca);
  if (Istar < 1.e-30 || !(kappa_ext > TINY_DP)) return;
  const Rt2Cols C = rt2_cols(M, A, ic);
  const double factor = A.photon_energy / M.volume[ic];
  const double yy = A.r_grid[ic], zz = A.z_grid[ic];
  const double norm = sqrt(yy * yy + zz * zz), u = 0.0, v = yy / norm, w = zz / norm;
  const double phi_pos = modulo_d(atan2(0.0, yy) + 2 * PI, 2 * PI);
  double phi = 2 * PI * (double)((float)iscatt / (float)A.nang_star);
  phi = phi - phi_pos;
  const double ur = A.uv0 * sin(phi), vr = -A.uv0 * cos(phi), wr = A.w0;
  const double prod1 = ur * u + vr * v;
  const double w2 = dir == 1 ? w : -w;
  const float cos_scatt = (float)(prod1 + wr * w2);
  int k = rt2_angle_index(cos_scatt, M.nang);
  if (k > M.nang) k = M.nang;
  if (k < 1) k = 1;
  const float s11 = C.s11[k];
  if (POLA) {
    double v1pi, v1pj, v1pk;
    rotation(u, v, w2, -ur, -vr, -wr, v1pi, v1pj, v1pk);
    const double xnyp = sqrt(v1pk * v1pk + v1pj * v1pj);
    const double costhet = (xnyp < 1e-10) ? 1.0 : v1pj / xnyp;
    double theta = acos(costhet);
    if (theta >= PI) theta = 0.0;
    double omega = 2.0 * theta;
    if (v1pk < 0.0) omega = -1.0 * omega;
    omega = (double)(float)omega;  // omega_ray_tracing_star is a default real
    double cosw = cos(omega), sinw = sin(omega);
    if (fabs(cosw) < 1e-06) cosw = 0.0;
    if (fabs(sinw) < 1e-06) sinw = 0.0;
    const float s12 = -s11 * C.s12[k];
    const double D1 = (double)s11 * Istar, D2 = (double)s12 * Istar;
    const float e0 = (float)((D1 * factor) * kappa_sca);
    const float e1 = (float)(((cosw * D2) * factor) * kappa_sca);
    const float e2 = (float)(((sinw * D2) * factor) * kappa_sca);
    e[0] = (float)((double)e0 / kappa_ext);
    const float Q = (float)((double)e1 / kappa_ext), U = (float)((double)e2 / kappa_ext);
    e[1] = sqrtf(Q * Q + U * U);
    e[2] = atan2f(U, Q);
    e[3] = 0.f;
  } else {
    const float e0 = (float)((((double)s11 * Istar) * factor) * kappa_sca);
    e[0] = (float)((double)e0 / kappa_ext);
  }
}

// mcgpu_set_I_spec: the reference's layout (N_type_flux, n_theta_I, n_phi_I, n_cells) -> the device's records
static __global__ void k_I_spec_put(double* dev, const double* in, int ntf, int nt, int np, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  size_t r = i;
  const int f = (int)(r % ntf); r /= ntf;
  const int th = (int)(r % nt); r /= nt;
  const int p = (int)(r % np); r /= np;  // r = cell
  dev[(((r * np + p) * nt) + th) * XI_LINE + f] = in[i];
}

}  // namespace mcgpu
