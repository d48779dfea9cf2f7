// Ray tracing method 1 on a Voronoi grid: integ_ray_dust (optical_depth.f90:1327-1421) with the grid's operators --
// move_to_grid_Voronoi (Voronoi.f90:1296-1442), cross_Voronoi_cell (:839-992), test_exit_grid_Voronoi (:1446) -- behind the
// kernels of mc_raytrace.hip.h (one ray per lane for the SED, one wavefront per pixel for images).  The grid is 3D: one
// xI_scatt record per cell and observer (n_az_rt = n_theta_rt = 1).
#pragma once
#include "mc_raytrace.hip.h"
#include "mc_voronoi.hip.h"

namespace mcgpu {

template <bool POLA>
__device__ inline void rt1_integ_ray_voro(const Lds& T, const DevModel& M, const VoroGrid& G, const RtArgs& A, int q, double x,
                                          double y, double z, double u0, double v0, double w0, double S[8]) {
  const int n_Stokes = POLA ? 4 : 1;
#pragma unroll
  for (int t = 0; t < 8; ++t) S[t] = 0.0;
  int icell = 0;
  if (!voro_move_to_grid(G, x, y, z, u0, v0, w0, icell)) return;
  const int i_star = intersect_stars(M, x, y, z, u0, v0, w0);
  const int star_icell = (i_star > 0) ? M.star_cell[4 * (i_star - 1)] : 0;
  double tau = 0.0;
  for (long guard = 0; guard < 100000000L; ++guard) {
    if (icell < 0) break;                                    // test_exit_grid
    if (star_icell > 0 && icell == star_icell) break;
    const VoroCell C = G.cell[icell - 1];
    double x1, y1, z1, l, l_contrib, l_void;
    int next;
    voro_cross_cell(G, M, C, x, y, z, u0, v0, w0, icell, 0, x1, y1, z1, next, l, l_contrib, l_void);  // (previous_cell = 0, :1393)
    if (icell <= M.n_cells) {
      const int ic = icell - 1;
      const size_t vrow = M.n_classes ? (size_t)M.cell_class[ic] * M.n_lambda + (A.lambda - 1) : 0;  // (lvariable_dust)
      const double kappa_ext = (M.n_classes ? M.v_kappa[vrow] : T.kappa[A.lambda - 1]) * C.kf;
      const double dtau = l_contrib * kappa_ext;
      if (kappa_ext > TINY_DP) {
        const double factor = A.photon_energy / M.volume[ic] * A.n_az_rt * A.n_theta_rt;
        const double kappa_sca = kappa_ext * (double)(M.n_classes ? M.v_albedo[vrow] : T.albedo[A.lambda - 1]);
        const size_t bin = (size_t)ic * A.n_theta_rt * A.n_az_rt;
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
    icell = next;
  }
}

template <bool POLA>
__global__ void __launch_bounds__(256) k_rt1_dust_map_voro(const DevModel M, const RtArgs A, const VoroGrid G) {
  extern __shared__ double lds_raw[];
  rt1_dust_map_body(M, A, lds_raw, [&](const Lds& T, int q, double x, double y, double z, double u, double v, double w, double* S) {
    rt1_integ_ray_voro<POLA>(T, M, G, A, q, x, y, z, u, v, w, S);
  });
}

template <bool POLA>
__global__ void __launch_bounds__(256) k_rt1_image_voro(const DevModel M, const RtArgs A, const VoroGrid G) {
  extern __shared__ double lds_raw[];
  rt1_image_body(M, A, lds_raw, [&](const Lds& T, int q, double x, double y, double z, double u, double v, double w, double* S) {
    rt1_integ_ray_voro<POLA>(T, M, G, A, q, x, y, z, u, v, w, S);
  });
}

// k_tau_maps (mc_raytrace.hip.h) on a Voronoi grid: move_to_grid_Voronoi, then cross_Voronoi_cell with the cut cells'
// l_contrib / l_void_before and the star's cell as the end of a ray.
__global__ void __launch_bounds__(256) k_tau_maps_voro(const DevModel M, const RtArgs A, const VoroGrid G, float tau_surface,
                                                       float* tau_map, float* surf_map) {
  extern __shared__ double lds_raw[];
  const Lds T = lds_carve(lds_raw, M, true);
  lds_stage_mono(T, M, 1);
  __syncthreads();
  const long n_pix = (long)A.npix_x * A.npix_y * A.nRT;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < n_pix; pix += (long)gridDim.x * blockDim.x) {
    int q;
    double pc[3], uvw[3];
    tau_maps_pixel(A, pix, q, pc, uvw);
    const double u = -uvw[0], v = -uvw[1], w = -uvw[2];
    double x = pc[0], y = pc[1], z = pc[2];
    int icell = 0;
    const bool hit = voro_move_to_grid(G, x, y, z, u, v, w, icell);
    auto opacity_of = [&](int ic, const VoroCell& C) {
      if (ic > M.n_cells) return 0.0;
      const double kap = M.n_classes ? M.v_kappa[(size_t)M.cell_class[ic - 1] * M.n_lambda + (A.lambda - 1)] : T.kappa[A.lambda - 1];
      return kap * C.kf;
    };
    if (tau_map) {
      double tau = 0.0, xa = x, ya = y, za = z;
      int next = icell, cur = 0, prev = 0;
      for (long guard = 0; hit && guard < 100000000L; ++guard) {
        prev = cur;
        cur = next;
        if (cur < 0) break;  // test_exit_grid
        const VoroCell C = G.cell[cur - 1];
        double x1, y1, z1, l, l_contrib, l_void;
        voro_cross_cell(G, M, C, xa, ya, za, u, v, w, cur, prev, x1, y1, z1, next, l, l_contrib, l_void);
        tau += l_contrib * opacity_of(cur, C);
        xa = x1; ya = y1; za = z1;
      }
      tau_map[pix] = (float)tau;
    }
    if (surf_map) {
      float out[3] = {0.0f, 0.0f, 0.0f};
      if (hit) {
        const int i_star = intersect_stars(M, x, y, z, u, v, w);
        const int star_icell = (i_star > 0) ? M.star_cell[4 * (i_star - 1)] : 0;
        double extr = (double)tau_surface, xo = x, yo = y, zo = z;
        int next = icell, cur = 0, prev = 0;
        for (long guard = 0; guard < 100000000L; ++guard) {
          prev = cur;
          cur = next;
          if (cur < 0) break;
          if (star_icell > 0 && cur == star_icell) break;
          const VoroCell C = G.cell[cur - 1];
          if (cur <= M.n_cells && M.dark && M.dark[cur - 1]) { out[0] = (float)xo; out[1] = (float)yo; out[2] = (float)zo; break; }
          double x1, y1, z1, l, l_contrib, l_void;
          voro_cross_cell(G, M, C, x, y, z, u, v, w, cur, prev, x1, y1, z1, next, l, l_contrib, l_void);
          const double tau = l_contrib * opacity_of(cur, C);
          if (tau > extr) {
            const double ls = l_void + l_contrib * (extr / tau);
            out[0] = (float)(x + ls * u); out[1] = (float)(y + ls * v); out[2] = (float)(z + ls * w);
            break;
          }
          extr = extr - tau;
          xo = x; yo = y; zo = z;
          x = x1; y = y1; z = z1;
        }
      }
      surf_map[pix] = out[0]; surf_map[pix + n_pix] = out[1]; surf_map[pix + 2 * n_pix] = out[2];
    }
  }
}

}  // namespace mcgpu
