// The branch-for-branch form of cross_cylindrical_cell (cylindrical_grid.f90:918-1175) on (ri,zj,k):
// kept for the CPU emulation tests (tests/test_kernel_emulation.py), where it cross-checks the
// branch-free cross_cell_lean the product runs.  Not part of the product build.
#pragma once
namespace mcgpu {
// cross_cylindrical_cell (cylindrical_grid.f90:918-1175) on (ri,zj,k).
// inv_a / inv_w (:941-952) are per-flight constants computed by the caller.
template <bool L3D>
__device__ inline void cross_cell(const Lds& T, const DevModel& M, double x0, double y0, double z0,
                                  double u, double v, double w, double inv_a, double inv_w, int ri0,
                                  int zj0, int k0, double& x1, double& y1, double& z1, int& ri1,
                                  int& zj1, int& k1, double& l) {
  const int nz = M.nz, n_rad = M.n_rad, n_az = M.n_az;
  const double correct_moins = 1.0 - GRID_PREC;
  const double correct_plus = 1.0 + GRID_PREC;
  double b, c, s, rac, t, t_phi, delta, r_2, zl, dotprod;
  int delta_rad = 0, delta_zj = 0, delta_phi = 0;

  r_2 = x0 * x0 + y0 * y0;
  b = (x0 * u + y0 * v) * inv_a;
  if (ri0 == 0) {
    c = (r_2 - T.r_lim_2[0]) * inv_a;
    delta = b * b - c;
    rac = sqrt(delta);
    s = (-b + rac) * correct_plus;
    t = HUGE_REAL;
    t_phi = HUGE_REAL;
    delta_rad = 1;
  } else {
    dotprod = u * x0 + v * y0;
    if (dotprod < 0.0) {
      c = (r_2 - T.r_lim_2[ri0 - 1] * correct_moins) * inv_a;
      delta = b * b - c;
      if (delta < 0.0) {
        c = (r_2 - T.r_lim_2[ri0] * correct_plus) * inv_a;
        delta = fmax(b * b - c, 0.0);
        delta_rad = 1;
      } else {
        delta_rad = -1;
      }
    } else {
      c = (r_2 - T.r_lim_2[ri0] * correct_plus) * inv_a;
      delta = fmax(b * b - c, 0.0);
      delta_rad = 1;
    }
    rac = sqrt(delta);
    s = (-b - rac) * correct_plus;
    if (s < 0.0) s = (-b + rac) * correct_plus;
    else if (s == 0.0) s = GRID_PREC;

    dotprod = w * z0;
    if (dotprod == 0.0) {
      t = 1.0e10;
    } else {
      const int azj0 = zj0 < 0 ? -zj0 : zj0;
      if (dotprod > 0.0) {
        if (azj0 == nz + 1) {
          delta_zj = 0;
          zl = copysign(1.0e10, z0);
        } else {
          zl = copysign(z_lim_of(T, nz, ri0, azj0 + 1) * correct_plus, z0);
          delta_zj = 1;
          if (L3D && (z0 < 0.0)) delta_zj = -1;
        }
      } else {
        if (L3D) {
          if (z0 > 0.0) {
            zl = z_lim_of(T, nz, ri0, azj0) * correct_moins;
            delta_zj = -1;
            if (zj0 == 1) delta_zj = -2;
          } else {
            zl = -z_lim_of(T, nz, ri0, azj0) * correct_moins;
            delta_zj = 1;
            if (zj0 == -1) delta_zj = 2;
          }
        } else {
          if (zj0 == 1) {
            delta_zj = 1;
            double zz = z_lim_of(T, nz, ri0, 2) * correct_moins;
            zl = (z0 > 0.0) ? -zz : zz;
          } else {
            double zz = z_lim_of(T, nz, ri0, zj0) * correct_moins;
            zl = (z0 > 0.0) ? zz : -zz;
            delta_zj = -1;
          }
        }
      }
      t = (zl - z0) * inv_w;
      if (t < 0.0) t = GRID_PREC;
    }

    if (L3D) {
      dotprod = x0 * v - y0 * u;
      const double r1e30 = 1.00000001504746621988e+30;  // real 1.0e30
      if (fabs(dotprod) < (double)1.0e-10f) {
        t_phi = r1e30;
      } else {
        double tan_angle_lim;
        if (dotprod > 0.0) {
          tan_angle_lim = T.tan_phi[k0 - 1];
          delta_phi = 1;
        } else {
          int k0m1 = k0 - 1;
          if (k0m1 == 0) k0m1 = n_az;
          tan_angle_lim = T.tan_phi[k0m1 - 1];
          delta_phi = -1;
        }
        if (tan_angle_lim > 1.0e299) {
          if (fabs(u) > (double)1e-6f) t_phi = -x0 / u;
          else t_phi = r1e30;
        } else {
          double den = v - u * tan_angle_lim;
          if (fabs(den) > (double)1.0e-6f) t_phi = -(y0 - x0 * tan_angle_lim) / den;
          else t_phi = r1e30;
        }
        if (t_phi < 0.0) t_phi = r1e30;
      }
    } else {
      t_phi = HUGE_REAL;
    }
  }

  if ((s < t) && (s < t_phi)) {
    l = s;
    x1 = x0 + s * u;
    y1 = y0 + s * v;
    z1 = z0 + s * w;
    ri1 = ri0 + delta_rad;
    if (ri1 == 0) {
      zj1 = 1;
      k1 = 1;
    } else {
      if (ri1 > n_rad) {
        zj1 = zj0;
      } else {
        int zj = zj_from_z_real(T, nz, fabs(z1), ri1);
        if (zj > nz) zj = nz + 1;
        if (L3D && (z1 < 0.0)) zj = -zj;
        zj1 = zj;
      }
      k1 = k0;
      if (L3D && (ri0 == 0)) {
        double phi = modulo_d(atan2(y1, x1), 2 * PI);
        int kk = (int)floor(phi * (1.0 / (2.0 * PI)) * (double)(float)n_az) + 1;
        if (kk == n_az + 1) kk = n_az;
        k1 = kk;
      }
    }
  } else if (t < t_phi) {
    l = t;
    x1 = x0 + t * u;
    y1 = y0 + t * v;
    // NOT fused: at the midplane zl = 0 has no grid_prec margin, so whether
    // z1 comes out as exactly 0 (-> sign(grid_prec,w), :1158-1165) or as a
    // rounding residue of either sign is decided by the rounding of t*w.
    // The reference build (no FMA contraction) rounds the product first.
    z1 = nd_add(z0, nd_mul(t, w));
    if (L3D && M.midplane_snap && (delta_zj == 2 || delta_zj == -2)) z1 = copysign(GRID_PREC, w);
    ri1 = ri0;
    zj1 = zj0 + delta_zj;
    k1 = k0;
  } else {
    l = t_phi;
    double dv = correct_plus * t_phi;
    x1 = x0 + dv * u;
    y1 = y0 + dv * v;
    z1 = z0 + dv * w;
    ri1 = ri0;
    int zj = (int)floor(fabs(z1) / T.zmax[ri1 - 1] * (double)nz) + 1;
    if (zj > nz) zj = nz + 1;
    if (z1 < 0.0) zj = -zj;
    zj1 = zj;
    int kk = k0 + delta_phi;
    if (kk == 0) kk = n_az;
    if (kk == n_az + 1) kk = 1;
    k1 = kk;
  }
  if (z1 == 0.0) {
    if (L3D) z1 = copysign(GRID_PREC, w);
    else z1 = GRID_PREC;
  }
}

}  // namespace mcgpu
