/*
 * mc_oracle.c -- CPU ORACLE (test infrastructure, never shipped) for the
 * MCFOST continuum Monte Carlo packet loop.  See mc_oracle.h for the status
 * header.  Every routine cites the reference routine it restates
 * (file:line into /root/reference/src/).
 *
 * Fortran default-real arithmetic is reproduced with C float where it
 * matters for discrete decisions (literals such as 1.0e30, the zj index
 * computed through real(), random numbers stored in default real).
 */
#include "mc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* constants.f90:8-13,151-159 */
static const double PI = 3.141592653589793238462643383279502884197;
static const double GRID_PREC = 1.0e-14;  /* cylindrical_grid.f90:16 */
#define TINY_REAL ((double)FLT_MIN)       /* tiny(0.0)   */
#define HUGE_REAL ((double)FLT_MAX)       /* huge(1.0)   */
#define HUGE_DP DBL_MAX                   /* huge(1.0_dp)*/
#define TINY_DP DBL_MIN                   /* tiny(0.0_dp)*/

static inline float max_int_real(void) {
  /* constants.f90:159  max_int = real(huge_integer) * (1.0-1.0e-5) */
  return (float)2147483647 * (1.0f - 1.0e-5f);
}
static inline double sign_d(double a, double b) {
  /* Fortran sign(a,b): |a| with the sign of b (b = +0 counts as positive) */
  return signbit(b) ? -fabs(a) : fabs(a);
}
static inline double modulo_d(double a, double p) {
  /* Fortran modulo(a,p) = a - floor(a/p)*p */
  return a - floor(a / p) * p;
}

/* ------------------------------------------------------------------------ */
/* Counter-based RNG: Philox4x32-10 (Salmon et al., SC'11).  The reference   */
/* uses SPRNG streams (random_numbers.f90:28); no reference test pins its    */
/* bit-stream, so any good generator is statistically equivalent.            */
/* ------------------------------------------------------------------------ */
void oracle_philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2],
                          uint32_t out[4]) {
  uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
  uint32_t k0 = key_in[0], k1 = key_in[1];
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

/*
 * Random stream layout (ours; the reference draws sequentially from SPRNG).  A packet's
 * stream is the sequence of Philox blocks ctr = (block, 0, packet_lo, packet_hi), four
 * default-real uniforms per block.  Draws are grouped per EVENT so that a wavefront computes
 * them in lock-step:
 *   event 0 (emission + first flight): blocks 0,1,2 -> f[0..11]
 *       f0 wavelength (select_wl_em)      f1 star / disk / ISM choice
 *       star:  f2 select_star, f3..f6 emit_packet_uniform_sphere
 *       disk:  f2 select_cellule, f3..f5 pos_em_cell, f6,f7 isotropic direction
 *       f8 optical depth of the first flight
 *   event e >= 1 (e-th interaction + next flight): ONE block, 3 + (e-1) -> five 24-bit uniforms: the upper 24 bits of
 *       the four words (h0..h3) and the three lower bytes of words 0, 1, 2 (h4)
 *       g0 = h0 scatter / absorb choice, g1 = h1 rand, g2 = h2 rand2
 *       scatter: g3 = h3 azimuth     absorb: g3 = h3, g4 = h1 isotropic direction (a re-emission does not use g1)
 *       g5 = h4 optical depth of the next flight
 *     (round 4: an event took two blocks; five draws fit in the 128 bits of one)
 *     scattering method 1 draws six numbers per scattering (grain, angle, angle, azimuth): two blocks as before,
 *       3+2(e-1), 4+2(e-1) -> g[0..7], g5 the optical depth
 */
typedef struct {
  uint32_t key[2];
  uint32_t p_lo, p_hi;
  uint32_t event;   /* next event index */
  float ev[12];     /* the current event's uniforms */
  int pos;          /* next unread entry of ev */
  int tau_idx;      /* where the current event keeps the flight's optical-depth draw */
  int two_blocks;   /* scattering method 1: two blocks per interaction (see above) */
} rng_t;

static inline float u32_to_real(uint32_t u) {
  /* Uniform default-real in [0,1) with 24 random bits (the reference rounds SPRNG's double
   * to default real at every call site, e.g. dust_transfer.f90:536,1073,1208). */
  return (float)(u >> 8) * (1.0f / 16777216.0f);
}

static void rng_init(rng_t *r, uint64_t seed, uint64_t packet) {
  r->key[0] = (uint32_t)seed;
  r->key[1] = (uint32_t)(seed >> 32);
  r->p_lo = (uint32_t)packet;
  r->p_hi = (uint32_t)(packet >> 32);
  r->event = 0;
  r->pos = 0;
  r->tau_idx = 8;
  r->two_blocks = 0;
}
static void rng_begin_event(rng_t *r) {
  if (r->event >= 1 && !r->two_blocks) {
    uint32_t ctr[4] = {3u + (r->event - 1u), 0u, r->p_lo, r->p_hi}, out[4];
    oracle_philox4x32_10(ctr, r->key, out);
    for (int q = 0; q < 4; ++q) r->ev[q] = u32_to_real(out[q]);
    r->ev[4] = r->ev[1];
    r->ev[5] = (float)(((out[0] & 0xFFu) << 16) | ((out[1] & 0xFFu) << 8) | (out[2] & 0xFFu)) * (1.0f / 16777216.0f);
    r->tau_idx = 5;
    r->pos = 0;
    r->event += 1;
    return;
  }
  const uint32_t first = r->event == 0 ? 0u : 3u + 2u * (r->event - 1u);
  const int nb = r->event == 0 ? 3 : 2;
  for (int b = 0; b < nb; ++b) {
    uint32_t ctr[4] = {first + (uint32_t)b, 0u, r->p_lo, r->p_hi}, out[4];
    oracle_philox4x32_10(ctr, r->key, out);
    for (int q = 0; q < 4; ++q) r->ev[4 * b + q] = u32_to_real(out[q]);
  }
  r->tau_idx = r->event == 0 ? 8 : 5;
  r->pos = 0;
  r->event += 1;
}
static inline float rng_float(rng_t *r) { return r->ev[r->pos++]; }
/* the modified random walk draws from its own counter sub-space: block k of the walk that follows event number
 * r->event (>= 1; the events themselves use ctr[1] = 0) */
static void rng_mrw_block(const rng_t *r, uint32_t k, float out4[4]) {
  uint32_t ctr[4] = {k, r->event, r->p_lo, r->p_hi}, out[4];
  oracle_philox4x32_10(ctr, r->key, out);
  for (int q = 0; q < 4; ++q) out4[q] = u32_to_real(out[q]);
}
static inline float rng_tau(const rng_t *r) { return r->ev[r->tau_idx]; }

float oracle_packet_rand(uint64_t seed, uint64_t packet, uint32_t n) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t ctr[4] = {n / 4u, 0u, (uint32_t)packet, (uint32_t)(packet >> 32)}, out[4];
  oracle_philox4x32_10(ctr, key, out);
  return u32_to_real(out[n % 4u]);
}

/* ------------------------------------------------------------------------ */
/* Cell mapping (cylindrical_grid.f90:45-179)                                */
/* ------------------------------------------------------------------------ */
void oracle_cell_mapping_sizes(int n_rad, int nz, int n_az, int l3D,
                               int *n_cells, int *ntot2, int *jdim_lo,
                               int *jdim_n) {
  int j_start = l3D ? -nz : 1;                 /* grid.f90:316-326 */
  int nrz = n_rad * nz;
  *n_cells = l3D ? 2 * nrz * n_az : nrz;       /* grid.f90:277-283 */
  int jstart2 = (j_start < 1 ? j_start : 1) - 1; /* :76 */
  int jend2 = nz + 1;
  if (jstart2 < 0)
    *ntot2 = (n_rad + 2) * (jend2 - jstart2) * n_az;      /* :83 */
  else
    *ntot2 = (n_rad + 2) * (jend2 - jstart2 + 1) * n_az;  /* :85 */
  *jdim_lo = jstart2;
  *jdim_n = jend2 - jstart2 + 1;
}

#define CM_IDX(n_rad, jlo, jn, i, j, k) \
  ((i) + ((n_rad) + 2) * (((j) - (jlo)) + (jn) * ((k)-1)))

int oracle_build_cell_mapping(int n_rad, int nz, int n_az, int l3D,
                              int *cell_map, int *cell_map_i, int *cell_map_j,
                              int *cell_map_k, int *lexit_cell) {
  int n_cells, ntot2, jlo, jn;
  oracle_cell_mapping_sizes(n_rad, nz, n_az, l3D, &n_cells, &ntot2, &jlo, &jn);
  int j_start = l3D ? -nz : 1;
  int istart2 = 0, iend2 = n_rad + 1, jstart2 = jlo, jend2 = nz + 1;
  int icell = 0;
  for (int q = 0; q < (n_rad + 2) * jn * n_az; ++q) cell_map[q] = 0;
  /* real cells (:90-107) */
  for (int k = 1; k <= n_az; ++k)
    for (int j = j_start; j <= nz; ++j) {
      if (j == 0) continue;
      for (int i = 1; i <= n_rad; ++i) {
        icell++;
        if (icell > n_cells) return 1;
        cell_map_i[icell - 1] = i;
        cell_map_j[icell - 1] = j;
        cell_map_k[icell - 1] = k;
        cell_map[CM_IDX(n_rad, jlo, jn, i, j, k)] = icell;
      }
    }
  if (icell != n_cells) return 2;
  for (int q = 0; q < ntot2; ++q) lexit_cell[q] = 0; /* :121 */
  /* virtual cells j = jstart2 and j = jend2 (:123-141) */
  for (int k = 1; k <= n_az; ++k)
    for (int j = jstart2; j <= jend2; j += jend2 - jstart2)
      for (int i = istart2; i <= iend2; ++i) {
        icell++;
        if (icell > ntot2) return 3;
        if (abs(j) == jend2) lexit_cell[icell - 1] = 2;
        if (i == iend2) lexit_cell[icell - 1] = 1;
        cell_map_i[icell - 1] = i;
        cell_map_j[icell - 1] = j;
        cell_map_k[icell - 1] = k;
        cell_map[CM_IDX(n_rad, jlo, jn, i, j, k)] = icell;
      }
  /* virtual cells i = 0 and i = n_rad+1 (:143-167) */
  for (int k = 1; k <= n_az; ++k)
    for (int j = j_start; j <= nz; ++j) {
      if (j == 0) continue;
      for (int i = istart2; i <= iend2; i += iend2 - istart2) {
        icell++;
        if (icell > ntot2) return 4;
        if (i == iend2) lexit_cell[icell - 1] = 1;
        cell_map_i[icell - 1] = i;
        cell_map_j[icell - 1] = j;
        cell_map_k[icell - 1] = k;
        cell_map[CM_IDX(n_rad, jlo, jn, i, j, k)] = icell;
      }
    }
  if (icell != ntot2) return 5;
  return 0;
}

static inline int cmap(const oracle_model *m, int i, int j, int k) {
  return m->cell_map[CM_IDX(m->n_rad, m->jdim_lo, m->jdim_n, i, j, k)];
}
/* lvariable_dust: the tables of the cell's class p_icell (optical_depth.f90:100-102 etc.) */
static inline double tab_kappa(const oracle_model *m, int icell, int lambda) {
  return m->p_n_cells ? m->v_kappa[(m->p_icell[icell - 1] - 1) + (size_t)m->p_n_cells * (lambda - 1)] : m->kappa[lambda - 1];
}
static inline double tab_kappa_abs(const oracle_model *m, int icell, int lambda) {
  return m->p_n_cells ? m->v_kappa_abs_LTE[(m->p_icell[icell - 1] - 1) + (size_t)m->p_n_cells * (lambda - 1)]
                      : m->kappa_abs_LTE[lambda - 1];
}
static inline float tab_albedo(const oracle_model *m, int icell, int lambda) {
  return m->p_n_cells ? m->v_albedo[(m->p_icell[icell - 1] - 1) + (size_t)m->p_n_cells * (lambda - 1)] : m->albedo[lambda - 1];
}
static inline const double *tab_log_Qcool(const oracle_model *m, int icell) {
  return m->p_n_cells ? m->v_log_Qcool + (size_t)(m->p_icell[icell - 1] - 1) * m->n_T : m->log_Qcool;
}
static inline const double *tab_cdf(const oracle_model *m, int icell) {
  return m->p_n_cells ? m->v_kdB_dT_CDF + (size_t)(m->p_icell[icell - 1] - 1) * m->n_T * m->n_lambda : m->kdB_dT_CDF;
}

static inline double zlim(const oracle_model *m, int i, int j) {
  return m->z_lim[(i - 1) + m->n_rad * (j - 1)];
}

/* ------------------------------------------------------------------------ */
/* Geometry operators                                                        */
/* ------------------------------------------------------------------------ */

/* cylindrical_grid.f90:680-704 */
int oracle_test_exit_grid_cyl(const oracle_model *m, int icell, double x,
                              double y, double z) {
  (void)x; (void)y;
  if (icell <= m->n_cells) return 0;
  int le = m->lexit_cell[icell - 1];
  if (le == 0) return 0;
  if (le == 1) return 1;
  return fabs(z) > m->zmaxmax;
}

/* zj from |z| exactly as cylindrical_grid.f90:868,1116: through default real */
static inline int zj_from_z_real(const oracle_model *m, double absz, int ri) {
  float q = (float)(absz / m->zmax[ri - 1] * (double)m->nz);
  float mi = max_int_real();
  if (!(q < mi)) q = mi; /* min(real(...), max_int) */
  return (int)floorf(q) + 1;
}

/* cylindrical_grid.f90:833-890 */
void oracle_index_cell_cyl(const oracle_model *m, double xin, double yin,
                           double zin, int *icell) {
  double r2 = xin * xin + yin * yin;
  int ri_out, zj_out, phik_out;
  if (r2 < m->r_lim_2[0]) {
    ri_out = 0; zj_out = 1; phik_out = 1;
  } else if (r2 > m->Rmax2) {
    ri_out = m->n_rad + 1; zj_out = 1; phik_out = 1;
  } else {
    int ri_min = 0, ri_max = m->n_rad;
    int ri = (ri_min + ri_max) / 2;
    while ((ri_max - ri_min) > 1) {
      if (r2 > m->r_lim_2[ri]) ri_min = ri; else ri_max = ri;
      ri = (ri_min + ri_max) / 2;
    }
    ri_out = ri + 1;
    zj_out = zj_from_z_real(m, fabs(zin), ri_out);
    if (m->l3D) {
      if (zj_out > m->nz) zj_out = m->nz + 1;
      if (zin < 0.0) zj_out = -zj_out;
      if (zin != 0.0) {
        double phi = modulo_d(atan2(yin, xin), 2 * PI);
        phik_out = (int)floor(phi / (2 * PI) * (double)(float)m->n_az) + 1;
        if (phik_out == m->n_az + 1) phik_out = m->n_az;
      } else {
        phik_out = 1;
      }
    } else {
      if (zj_out > m->nz) zj_out = m->nz + 1;
      phik_out = 1;
    }
  }
  *icell = cmap(m, ri_out, zj_out, phik_out);
}

/* cylindrical_grid.f90:918-1175 */
void oracle_cross_cylindrical_cell(const oracle_model *m, double x0, double y0,
                                   double z0, double u, double v, double w,
                                   int cell, int previous_cell, double *x1,
                                   double *y1, double *z1, int *next_cell,
                                   double *l_out, double *l_contrib,
                                   double *l_void_before) {
  (void)previous_cell;
  const int nz = m->nz, n_rad = m->n_rad, n_az = m->n_az, l3D = m->l3D;
  const double correct_moins = 1.0 - GRID_PREC; /* :938 */
  const double correct_plus = 1.0 + GRID_PREC;  /* :939 */
  double inv_a, inv_w, a, b, c, s, rac, t, t_phi, delta, r_2, den;
  double tan_angle_lim, phi, delta_vol, zl, dotprod, l;
  int ri0, zj0, k0, k0m1, delta_rad = 0, delta_zj = 0, delta_phi = 0;
  int ri1, zj1, k1;

  a = u * u + v * v;                                   /* :941-946 */
  if (a > TINY_REAL) inv_a = 1.0 / a; else inv_a = HUGE_REAL;
  if (fabs(w) > TINY_REAL) inv_w = 1.0 / w;            /* :948-952 */
  else inv_w = sign_d(HUGE_DP, w);

  ri0 = m->cell_map_i[cell - 1];                       /* :956 */
  zj0 = m->cell_map_j[cell - 1];
  k0 = m->cell_map_k[cell - 1];

  r_2 = x0 * x0 + y0 * y0;                             /* :959-960 */
  b = (x0 * u + y0 * v) * inv_a;

  if (ri0 == 0) {                                      /* :962-971 */
    c = (r_2 - m->r_lim_2[0]) * inv_a;
    delta = b * b - c;
    rac = sqrt(delta);
    s = (-b + rac) * correct_plus;
    t = HUGE_REAL;
    t_phi = HUGE_REAL;
    delta_rad = 1;
  } else {
    /* 1) radial interface (:973-1000) */
    dotprod = u * x0 + v * y0;
    if (dotprod < 0.0) {
      c = (r_2 - m->r_lim_2[ri0 - 1] * correct_moins) * inv_a;
      delta = b * b - c;
      if (delta < 0.0) {
        c = (r_2 - m->r_lim_2[ri0] * correct_plus) * inv_a;
        delta = fmax(b * b - c, 0.0);
        delta_rad = 1;
      } else {
        delta_rad = -1;
      }
    } else {
      c = (r_2 - m->r_lim_2[ri0] * correct_plus) * inv_a;
      delta = fmax(b * b - c, 0.0);
      delta_rad = 1;
    }
    rac = sqrt(delta);
    s = (-b - rac) * correct_plus;
    if (s < 0.0) s = (-b + rac) * correct_plus;
    else if (s == 0.0) s = GRID_PREC;

    /* 2) vertical interface (:1003-1055) */
    dotprod = w * z0;
    if (dotprod == 0.0) {
      t = (double)1.0e10f;
    } else {
      if (dotprod > 0.0) {
        if (abs(zj0) == nz + 1) {
          delta_zj = 0;
          zl = sign_d(1.0e10, z0);
        } else {
          zl = sign_d(zlim(m, ri0, abs(zj0) + 1) * correct_plus, z0);
          delta_zj = 1;
          if (l3D && (z0 < 0.0)) delta_zj = -1;
        }
      } else {
        if (l3D) {
          if (z0 > 0.0) {
            zl = zlim(m, ri0, abs(zj0)) * correct_moins;
            delta_zj = -1;
            if (zj0 == 1) delta_zj = -2;
          } else {
            zl = -zlim(m, ri0, abs(zj0)) * correct_moins;
            delta_zj = 1;
            if (zj0 == -1) delta_zj = 2;
          }
        } else {
          if (zj0 == 1) { /* cross the midplane, z changes sign (:1032-1040) */
            delta_zj = 1;
            if (z0 > 0.0) zl = -zlim(m, ri0, 2) * correct_moins;
            else zl = zlim(m, ri0, 2) * correct_moins;
          } else {
            if (z0 > 0.0) zl = zlim(m, ri0, zj0) * correct_moins;
            else zl = -zlim(m, ri0, zj0) * correct_moins;
            delta_zj = -1;
          }
        }
      }
      t = (zl - z0) * inv_w;
      if (t < 0.0) t = GRID_PREC;
    }

    /* 3) azimuthal interface (:1058-1094) */
    if (l3D) {
      dotprod = x0 * v - y0 * u;
      if (fabs(dotprod) < (double)1.0e-10f) {
        t_phi = (double)1.0e30f;
      } else {
        if (dotprod > 0.0) {
          tan_angle_lim = m->tan_phi_lim[k0 - 1];
          delta_phi = 1;
        } else {
          k0m1 = k0 - 1;
          if (k0m1 == 0) k0m1 = n_az;
          tan_angle_lim = m->tan_phi_lim[k0m1 - 1];
          delta_phi = -1;
        }
        if (tan_angle_lim > 1.0e299) {
          if (fabs(u) > (double)1e-6f) t_phi = -x0 / u;
          else t_phi = (double)1.0e30f;
        } else {
          den = v - u * tan_angle_lim;
          if (fabs(den) > (double)1.0e-6f)
            t_phi = -(y0 - x0 * tan_angle_lim) / den;
          else
            t_phi = (double)1.0e30f;
        }
        if (t_phi < 0.0) t_phi = (double)1.0e30f;
      }
    } else {
      t_phi = HUGE_REAL;
    }
  }

  /* 4) which interface (:1098-1156) */
  if ((s < t) && (s < t_phi)) {
    l = s;
    delta_vol = s;
    *x1 = x0 + delta_vol * u;
    *y1 = y0 + delta_vol * v;
    *z1 = z0 + delta_vol * w;
    ri1 = ri0 + delta_rad;
    if (ri1 == 0) {
      zj1 = 1;
      k1 = 1;
    } else {
      if (ri1 > n_rad) {
        zj1 = zj0;
      } else {
        zj1 = zj_from_z_real(m, fabs(*z1), ri1);
        if (zj1 > nz) zj1 = nz + 1;
        if (l3D && (*z1 < 0.0)) zj1 = -zj1;
      }
      k1 = k0;
      if ((ri0 == 0) && l3D) {
        phi = modulo_d(atan2(*y1, *x1), 2 * PI);
        k1 = (int)floor(phi * (1.0 / (2.0 * PI)) * (double)(float)n_az) + 1;
        if (k1 == n_az + 1) k1 = n_az;
      }
    }
  } else if (t < t_phi) {
    l = t;
    delta_vol = t;
    *x1 = x0 + delta_vol * u;
    *y1 = y0 + delta_vol * v;
    *z1 = z0 + delta_vol * w;
    ri1 = ri0;
    zj1 = zj0 + delta_zj;
    k1 = k0;
    if (l3D && m->midplane_snap && (delta_zj == 2 || delta_zj == -2))
      *z1 = sign_d(GRID_PREC, w); /* see mc_oracle.h: midplane_snap */
  } else {
    l = t_phi;
    delta_vol = correct_plus * t_phi;
    *x1 = x0 + delta_vol * u;
    *y1 = y0 + delta_vol * v;
    *z1 = z0 + delta_vol * w;
    ri1 = ri0;
    zj1 = (int)floor(fabs(*z1) / m->zmax[ri1 - 1] * (double)nz) + 1;
    if (zj1 > nz) zj1 = nz + 1;
    if (*z1 < 0.0) zj1 = -zj1;
    k1 = k0 + delta_phi;
    if (k1 == 0) k1 = n_az;
    if (k1 == n_az + 1) k1 = 1;
  }

  if (*z1 == 0.0) {                                    /* :1158-1165 */
    if (l3D) *z1 = sign_d(GRID_PREC, w);
    else *z1 = GRID_PREC;
  }

  *next_cell = cmap(m, ri1, zj1, k1);                  /* :1168 */
  *l_out = l;
  *l_contrib = l;
  *l_void_before = 0.0;
}

/* cylindrical_grid.f90:1284-1411 */
void oracle_move_to_grid_cyl(const oracle_model *m, double *x, double *y,
                             double *z, double u, double v, double w,
                             int *icell, int *lintersect) {
  const double correct_moins = 1.0 - 1.0e-10;
  double x0 = *x, y0 = *y, z0 = *z, z1, a, inv_a, r_2, b, c, delta, rac;
  double s1, s2, dotprod, t1, t2, zl, zl2, delta_vol, inv_w;

  a = u * u + v * v;
  if (a > TINY_REAL) inv_a = 1.0 / a; else inv_a = HUGE_REAL;
  if (fabs(w) > TINY_REAL) inv_w = 1.0 / w; else inv_w = sign_d(HUGE_DP, w);

  r_2 = x0 * x0 + y0 * y0;
  b = (x0 * u + y0 * v) * inv_a;
  c = (r_2 - m->r_lim_2[m->n_rad] * correct_moins) * inv_a;
  delta = b * b - c;
  if (delta < 0.0) {
    s1 = HUGE_REAL; s2 = HUGE_REAL;
  } else {
    rac = sqrt(delta);
    s1 = -b - rac;
    s2 = -b + rac;
  }
  dotprod = w * z0;
  if (fabs(dotprod) < TINY_REAL) {
    t1 = HUGE_REAL; t2 = HUGE_REAL;
  } else {
    if (z0 > 0.0) {
      zl = m->zmaxmax * correct_moins;
      zl2 = -m->zmaxmax * correct_moins;
    } else {
      zl = -m->zmaxmax * correct_moins;
      zl2 = m->zmaxmax * correct_moins;
    }
    t1 = (zl - z0) * inv_w;
    t2 = (zl2 - z0) * inv_w;
  }
  if (t1 > (double)1e20f) {
    if (s1 > (double)1e20f) { *lintersect = 0; return; }
  }
  if (t1 > s1) {
    if (t1 > s2) {
      delta_vol = s1;
      z1 = z0 + delta_vol * w;
      if (fabs(z1) > m->zmaxmax) { *lintersect = 0; return; }
      *lintersect = 1;
    } else {
      *lintersect = 1;
      delta_vol = t1;
    }
  } else {
    if (t2 < s1) { *lintersect = 0; return; }
    *lintersect = 1;
    delta_vol = s1;
  }
  *x = x0 + delta_vol * u;
  *y = y0 + delta_vol * v;
  *z = z0 + delta_vol * w;
  oracle_index_cell_cyl(m, *x, *y, *z, icell);
}

/* cylindrical_grid.f90:1415-1466 */
void oracle_pos_em_cell_cyl(const oracle_model *m, int icell, float rand1,
                            float rand2, float rand3, double *x, double *y,
                            double *z) {
  int ri = m->cell_map_i[icell - 1];
  int zj = m->cell_map_j[icell - 1];
  int phik = m->cell_map_k[icell - 1];
  double r = sqrt(m->r_lim_2[ri - 1] +
                  (double)rand1 * (m->r_lim_2[ri] - m->r_lim_2[ri - 1]));
  if (m->l3D) {
    if (zj > 0)
      *z = zlim(m, ri, zj) + (double)rand2 * (zlim(m, ri, zj + 1) - zlim(m, ri, zj));
    else
      *z = -(zlim(m, ri, -zj) +
             (double)rand2 * (zlim(m, ri, -zj + 1) - zlim(m, ri, -zj)));
  } else {
    if ((double)rand2 > 0.5)
      *z = zlim(m, ri, zj) + (2.0 * ((double)rand2 - 0.5)) *
                                 (zlim(m, ri, abs(zj) + 1) - zlim(m, ri, zj));
    else
      *z = -(zlim(m, ri, zj) +
             (2.0 * (double)rand2) * (zlim(m, ri, zj + 1) - zlim(m, ri, zj)));
  }
  double phi = 2.0 * PI * ((double)phik - 1.0 + (double)rand3) / (double)m->n_az;
  *x = r * cos(phi);
  *y = r * sin(phi);
}

/* ------------------------------------------------------------------------ */
/* Voronoi grid operators (Voronoi.f90).  The plane tests are in default real  */
/* exactly like the reference (n, p, r, k are `real` arrays, :859).             */
/* ------------------------------------------------------------------------ */

/* distance_to_wall (Voronoi.f90:1289-1317) */
static double voro_distance_to_wall(const oracle_model *m, double x, double y,
                                    double z, double u, double v, double w,
                                    int iwall) {
  const float *W = m->v_walls + 4 * (iwall - 1);
  double n[3] = {W[0], W[1], W[2]};
  double p[3] = {W[3] * fabs(n[0]), W[3] * fabs(n[1]), W[3] * fabs(n[2])};
  float den = (float)(n[0] * u + n[1] * v + n[2] * w);
  if (fabsf(den) > FLT_MIN)
    return (n[0] * (p[0] - x) + n[1] * (p[1] - y) + n[2] * (p[2] - z)) / (double)den;
  return (double)FLT_MAX;
}

/* distance_to_star (Voronoi.f90:1321-1375) */
static double voro_distance_to_star(const oracle_model *m, double x, double y,
                                    double z, double u, double v, double w,
                                    int *i_star) {
  double d = DBL_MAX;
  *i_star = 0;
  for (int i = 1; i <= m->n_stars; ++i) {
    const oracle_star *st = &m->stars[i - 1];
    double dx = x - st->x, dy = y - st->y, dz = z - st->z;
    double b = dx * u + dy * v + dz * w;
    double c = dx * dx + dy * dy + dz * dz - st->r * st->r;
    double delta = b * b - c;
    if (delta >= 0.) {
      double rac = sqrt(delta), s1 = -b - rac;
      if (s1 < 0) {
        double s2 = -b + rac;
        if (s2 > 0) { d = 0.0; *i_star = i; }
      } else if (s1 < d) {
        d = s1; *i_star = i;
      }
    }
  }
  return d;
}

/* is_in_volume (Voronoi.f90:1462-1478) */
static int voro_is_in_volume(const oracle_model *m, double x, double y, double z) {
  const float *W = m->v_walls;
  return (x > W[3]) && (x < W[7]) && (y > W[11]) && (y < W[15]) && (z > W[19]) && (z < W[23]);
}

/* index_cell_voronoi (Voronoi.f90:1548-1570): brute force, default-real distances */
void oracle_index_cell_voronoi(const oracle_model *m, double xin, double yin,
                               double zin, int *icell) {
  float dist2_min = FLT_MAX;
  for (int i = 1; i <= m->n_cells; ++i) {
    const double *c = m->v_xyz_dp + 3 * (size_t)(i - 1);
    float dist2 = (float)((c[0] - xin) * (c[0] - xin) + (c[1] - yin) * (c[1] - yin) +
                          (c[2] - zin) * (c[2] - zin));
    if (dist2 < dist2_min) { *icell = i; dist2_min = dist2; }
  }
}

/* cross_Voronoi_cell (Voronoi.f90:839-992) */
void oracle_cross_voronoi_cell(const oracle_model *m, double x, double y,
                               double z, double u, double v, double w,
                               int icell, int previous_cell, double *x1,
                               double *y1, double *z1, int *next_cell,
                               double *s_out, double *s_contrib,
                               double *s_void_before) {
  const double prec = (double)1e-5f;
  const float r[3] = {(float)x, (float)y, (float)z};
  const float k[3] = {(float)u, (float)v, (float)w};
  double s = (double)1e30f;
  *next_cell = 0;
  const double *rc_dp = m->v_xyz_dp + 3 * (size_t)(icell - 1);
  const float r_cell[3] = {(float)rc_dp[0], (float)rc_dp[1], (float)rc_dp[2]};
  const int ifirst = m->v_first[icell - 1], ilast = m->v_last[icell - 1];
  const int was_cut = m->v_was_cut ? m->v_was_cut[icell - 1] : 0;
  const double h = m->v_h[icell - 1];
  const int star_nb = m->v_is_star_neighbour ? m->v_is_star_neighbour[icell - 1] : 0;

  for (int i = ifirst; i <= ilast; ++i) {
    const int id_n = m->v_neigh[i - 1];
    double s_tmp;
    if (id_n == previous_cell) continue;
    if (id_n > 0) {
      const float *rn = m->v_xyz + 3 * (size_t)(id_n - 1);
      const float n[3] = {rn[0] - r_cell[0], rn[1] - r_cell[1], rn[2] - r_cell[2]};
      /* den is real(dp) in the reference: the default-real dot product, widened */
      const double den = (double)(n[0] * k[0] + n[1] * k[1] + n[2] * k[2]);
      if (den <= 0.) continue;
      const float p[3] = {0.5f * (rn[0] + r_cell[0]), 0.5f * (rn[1] + r_cell[1]), 0.5f * (rn[2] + r_cell[2])};
      s_tmp = (double)(n[0] * (p[0] - r[0]) + n[1] * (p[1] - r[1]) + n[2] * (p[2] - r[2])) / den;
      if (s_tmp < 0.) s_tmp = (double)FLT_MAX;
    } else {
      s_tmp = voro_distance_to_wall(m, x, y, z, u, v, w, -id_n);
      if (s_tmp < 0.) s_tmp = (double)FLT_MAX;
    }
    if (s_tmp < s) { s = s_tmp; *next_cell = id_n; }
  }
  s = s * (1.0 + prec);
  *x1 = x + u * s;
  *y1 = y + v * s;
  *z1 = z + w * s;
  if (*next_cell == 0) { /* rounding error somewhere (:926-937) */
    *x1 = x; *y1 = y; *z1 = z; s = 0.0;
    if (voro_is_in_volume(m, x, y, z)) {
      oracle_index_cell_voronoi(m, x, y, z, next_cell);
      if (icell == *next_cell) *next_cell = -1;
    } else {
      *next_cell = -1;
    }
  }
  if (was_cut) { /* :939-975 */
    const double dr[3] = {(double)(r[0] - r_cell[0]), (double)(r[1] - r_cell[1]), (double)(r[2] - r_cell[2])};
    const double b = dr[0] * (double)k[0] + dr[1] * (double)k[1] + dr[2] * (double)k[2];
    const double hc = h * m->v_cut_o_h;
    const double c = dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2] - hc * hc;
    const double delta = b * b - c;
    if (delta < 0.) {
      *s_void_before = s; *s_contrib = 0.0;
    } else {
      const double rac = sqrt(delta), s1 = -b - rac, s2 = -b + rac;
      if (s1 < 0) {
        if (s2 < 0) { *s_void_before = s; *s_contrib = 0.0; }
        else { *s_void_before = 0.0; *s_contrib = fmin(s2, s); }
      } else if (s1 < s) {
        *s_void_before = s1; *s_contrib = fmin(s2, s) - s1;
      } else {
        *s_void_before = s; *s_contrib = 0.0;
      }
    }
  } else {
    *s_void_before = 0.0; *s_contrib = s;
  }
  if (star_nb) { /* :977-988 */
    int i_star;
    const double d_to_star = voro_distance_to_star(m, x, y, z, u, v, w, &i_star);
    if (i_star > 0 && d_to_star < s) {
      *s_contrib = d_to_star;
      *next_cell = m->stars[i_star - 1].icell;
    }
  }
  *s_out = s;
}

/* move_to_grid_Voronoi (Voronoi.f90:1379-1442) with find_Voronoi_cell_brute_force (:1485) */
/* find_Voronoi_cell (Voronoi.f90:1625-1645): the site of wall iwall's neighbour list closest to the point, by
 * kdtree2_n_nearest with NN = 1 -- a nearest-neighbour search in kdkind = dp (kdtree2.f90:23), whose answer is the
 * minimum of the dp squared distances summed over the coordinates in order (kdtree2.f90 process_terminal_node).
 * PINNED to the reference's kdtree2 module compiled in oracle/_ref (tests/golden/kdtree_nearest.npz). */
int oracle_find_voronoi_cell(const oracle_model *m, int iwall, double x, double y, double z) {
  double dist2_min = 1.79769313486231570815e+308;
  int icell_min = 0;
  for (int q = m->v_wall_first[iwall - 1]; q < m->v_wall_first[iwall]; ++q) {
    const int ic = m->v_wall_cells[q];
    const double *c = m->v_xyz_dp + 3 * (size_t)(ic - 1);
    const double dist2 = (c[0] - x) * (c[0] - x) + (c[1] - y) * (c[1] - y) + (c[2] - z) * (c[2] - z);
    if (dist2 < dist2_min) { icell_min = ic; dist2_min = dist2; }
  }
  return icell_min;
}

void oracle_move_to_grid_voronoi(const oracle_model *m, double *x, double *y,
                                 double *z, double u, double v, double w,
                                 int *icell, int *lintersect) {
  const double prec = 1.e-6; /* module parameter (:21) */
  double s_walls[6];
  int order[6];
  for (int iw = 1; iw <= 6; ++iw) {
    double l = voro_distance_to_wall(m, *x, *y, *z, u, v, w, iw);
    s_walls[iw - 1] = (l >= 0) ? l * (1.0 + prec) : (double)FLT_MAX;
    order[iw - 1] = iw;
  }
  for (int a = 1; a < 6; ++a) /* index_quicksort: ascending */
    for (int b = a; b > 0 && s_walls[order[b] - 1] < s_walls[order[b - 1] - 1]; --b) {
      int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t;
    }
  int iwall = 0;
  double xt = 0, yt = 0, zt = 0;
  for (int i = 0; i < 6; ++i) {
    iwall = order[i];
    const double l = s_walls[iwall - 1];
    xt = *x + l * u; yt = *y + l * v; zt = *z + l * w;
    if (voro_is_in_volume(m, xt, yt, zt)) break;
    if (i == 5) { *icell = 0; *lintersect = 0; return; }
  }
  *lintersect = 1;
  *x = xt; *y = yt; *z = zt;
  *icell = oracle_find_voronoi_cell(m, iwall, xt, yt, zt);
}

/* ------------------------------------------------------------------------ */
/* Spherical grid (spherical_grid.f90)                                        */
/* ------------------------------------------------------------------------ */
static const double PREC_GRILLE_SPH = 1.0e-7; /* spherical_grid.f90:19 */

/* spherical_grid.f90:24-44: only the outer radius is an exit */
int oracle_test_exit_grid_sph(const oracle_model *m, int icell) {
  if (icell <= m->n_cells) return 0;
  return m->lexit_cell[icell - 1] == 1;
}

/* indice_cellule_sph_theta (:129-178) and the theta / phi part of index_cell_sph (:83-120) */
static void sph_theta_phi(const oracle_model *m, double xin, double yin, double zin, int *thetaj_out,
                          int *phik_out) {
  const double r02 = xin * xin + yin * yin;
  double tan_theta;
  if (r02 > TINY_DP) tan_theta = fabs(zin) / sqrt(r02);
  else tan_theta = (double)1.0e30f; /* default-real literal 1.0e30 */
  int tmin = 0, tmax = m->nz, tj = (tmin + tmax) / 2;
  while ((tmax - tmin) > 1) {
    if (tan_theta > m->tan_theta_lim[tj]) tmin = tj; else tmax = tj;
    tj = (tmin + tmax) / 2;
  }
  *thetaj_out = tj + 1;
  if (m->l3D) {
    if (zin < 0.0) *thetaj_out = -*thetaj_out;
    if (zin != 0.0) {
      double phi = modulo_d(atan2(yin, xin), 2 * PI);
      int pk = (int)floor(phi / (2 * PI) * (double)(float)m->n_az) + 1;
      if (pk == m->n_az + 1) pk = m->n_az;
      *phik_out = pk;
    } else {
      *phik_out = 1;
    }
  } else {
    *phik_out = 1;
  }
}

/* spherical_grid.f90:48-125 */
void oracle_index_cell_sph(const oracle_model *m, double xin, double yin, double zin, int *icell) {
  const double r02 = xin * xin + yin * yin;
  const double r2 = r02 + zin * zin;
  int ri_out, thetaj_out, phik_out;
  if (r2 < m->r_lim_2[0]) {
    ri_out = 0; thetaj_out = 1; phik_out = 1;
  } else if (r2 > m->Rmax2) {
    ri_out = m->n_rad + 1; thetaj_out = 1; phik_out = 1;
  } else {
    int ri_min = 0, ri_max = m->n_rad, ri = (ri_min + ri_max) / 2;
    while ((ri_max - ri_min) > 1) {
      if (r2 > m->r_lim_2[ri]) ri_min = ri; else ri_max = ri;
      ri = (ri_min + ri_max) / 2;
    }
    ri_out = ri + 1;
    sph_theta_phi(m, xin, yin, zin, &thetaj_out, &phik_out);
  }
  *icell = cmap(m, ri_out, thetaj_out, phik_out);
}

/* one theta cone (:247-276, 279-307): smallest positive root of the crossing with tan(theta) = tan_lim */
static double sph_theta_root(double x0, double y0, double z0, double u, double v, double w, double tan_lim) {
  const double precision = 1.0e-15;
  const double tan2 = tan_lim * tan_lim;
  const double a_theta = w * w - tan2 * (u * u + v * v);
  const double a_theta_m1 = 1.0 / a_theta;
  const double b_theta = w * z0 - tan2 * (x0 * u + y0 * v);
  const double c_theta = z0 * z0 - tan2 * (x0 * x0 + y0 * y0);
  const double delta = b_theta * b_theta - a_theta * c_theta;
  if (delta < 0.0) return 1.0e30;
  const double rac = sqrt(delta);
  const double t_1 = (-b_theta - rac) * a_theta_m1;
  const double t_2 = (-b_theta + rac) * a_theta_m1;
  if (t_1 <= precision) {
    if (t_2 <= precision) return 1.0e30;
    return t_2;
  }
  if (t_2 <= precision) return t_1;
  return t_1 < t_2 ? t_1 : t_2;
}

/* spherical_grid.f90:182-446 */
void oracle_cross_spherical_cell(const oracle_model *m, double x0, double y0, double z0, double u, double v,
                                 double w, int cell, int previous_cell, double *x1, double *y1, double *z1,
                                 int *next_cell, double *l, double *l_contrib, double *l_void_before) {
  (void)previous_cell;
  const double correct_moins = 1.0 - PREC_GRILLE_SPH, correct_plus = 1.0 + PREC_GRILLE_SPH;
  const int ri0 = m->cell_map_i[cell - 1], thetaj0 = m->cell_map_j[cell - 1], phik0 = m->cell_map_k[cell - 1];
  const double r0_2_cyl = x0 * x0 + y0 * y0;
  const double r0_2 = r0_2_cyl + z0 * z0;
  const double b = (x0 * u + y0 * v + z0 * w);
  double c, delta, rac, s, t, t_phi;
  int delta_rad, delta_theta = 0, delta_phi = 0;
  if (ri0 == 0) {
    c = (r0_2 - m->r_lim_2[0] * correct_plus);
    delta = b * b - c;
    rac = sqrt(delta);
    s = (-b + rac) * correct_plus;
    t = HUGE_REAL;
    delta_rad = 1;
    t_phi = HUGE_REAL;
  } else {
    /* 1) radial interface */
    if (b < 0.0) {
      c = (r0_2 - m->r_lim_2[ri0 - 1] * correct_moins);
      delta = b * b - c;
      if (delta < 0.0) {
        c = (r0_2 - m->r_lim_2[ri0] * correct_plus);
        delta = fmax(b * b - c, 0.0);
        delta_rad = 1;
      } else {
        delta_rad = -1;
      }
    } else {
      c = (r0_2 - m->r_lim_2[ri0] * correct_plus);
      delta = fmax(b * b - c, 0.0);
      delta_rad = 1;
    }
    rac = sqrt(delta);
    s = -b - rac;
    if (s < 0.0) s = -b + rac;
    else if (s == 0.0) s = GRID_PREC;
    /* 2) the two theta cones of the cell */
    const int aj = thetaj0 < 0 ? -thetaj0 : thetaj0;
    double tan_angle_lim1, tan_angle_lim2;
    if (z0 >= 0.0) {
      tan_angle_lim1 = m->tan_theta_lim[aj] * correct_plus;
      tan_angle_lim2 = m->tan_theta_lim[aj - 1] * correct_moins;
    } else {
      tan_angle_lim1 = -m->tan_theta_lim[aj] * correct_plus;
      tan_angle_lim2 = -m->tan_theta_lim[aj - 1] * correct_moins;
    }
    const double t1 = sph_theta_root(x0, y0, z0, u, v, w, tan_angle_lim1);
    const double t2 = sph_theta_root(x0, y0, z0, u, v, w, tan_angle_lim2);
    if (t1 < t2) {
      t = t1;
      delta_theta = 1;
      if (aj == m->nz) delta_theta = 0;
    } else {
      t = t2;
      delta_theta = -1;
      if (aj == 1) delta_theta = 0;
    }
    /* 3) azimuthal interface */
    if (m->l3D) {
      const double dotprod = x0 * v - y0 * u;
      if (fabs(dotprod) < (double)1.0e-10f) {
        t_phi = (double)1.0e30f;
        delta_phi = 0;
      } else {
        double tan_angle_lim;
        if (dotprod > 0.0) {
          tan_angle_lim = m->tan_phi_lim[phik0 - 1];
          delta_phi = 1;
        } else {
          int phik0m1 = phik0 - 1;
          if (phik0m1 == 0) phik0m1 = m->n_az;
          tan_angle_lim = m->tan_phi_lim[phik0m1 - 1];
          delta_phi = -1;
        }
        if (tan_angle_lim > 1.0e299) {
          t_phi = -x0 / u;
        } else {
          const double den = v - u * tan_angle_lim;
          if (fabs(den) > (double)1.0e-6f) {
            t_phi = -(y0 - x0 * tan_angle_lim) / den;
          } else {
            t_phi = (double)1.0e30f;
            delta_phi = 0;
          }
        }
        if (t_phi < 0.0) {
          t_phi = (double)1.0e30f;
          delta_phi = 0;
        }
      }
    } else {
      t_phi = HUGE_REAL;
    }
  }
  /* 4) which interface */
  int ri1, thetaj1, phik1;
  if ((s < t) && (s < t_phi)) {
    *l = s;
    *x1 = x0 + s * u; *y1 = y0 + s * v; *z1 = z0 + s * w;
    ri1 = ri0 + delta_rad;
    thetaj1 = thetaj0;
    phik1 = phik0;
    if (ri0 == 0) sph_theta_phi(m, *x1, *y1, *z1, &thetaj1, &phik1);
    if (ri1 == 0) { thetaj1 = 1; phik1 = 1; }
  } else if (t < t_phi) {
    *l = t;
    *x1 = x0 + t * u; *y1 = y0 + t * v; *z1 = z0 + t * w;
    ri1 = ri0;
    thetaj1 = (thetaj0 < 0 ? -thetaj0 : thetaj0) + delta_theta;
    if (m->l3D) {
      if (*z1 < 0) thetaj1 = -thetaj1;
    }
    phik1 = phik0;
  } else {
    *l = t_phi;
    const double delta_vol = correct_plus * t_phi;
    *x1 = x0 + delta_vol * u; *y1 = y0 + delta_vol * v; *z1 = z0 + delta_vol * w;
    ri1 = ri0;
    thetaj1 = thetaj0;
    phik1 = phik0 + delta_phi;
    if (phik1 == 0) phik1 = m->n_az;
    if (phik1 == m->n_az + 1) phik1 = 1;
  }
  if (*z1 == 0.0) *z1 = GRID_PREC;
  *next_cell = cmap(m, ri1, thetaj1, phik1);
  *l_contrib = *l;
  *l_void_before = 0.0;
}

/* spherical_grid.f90:562-615 */
void oracle_move_to_grid_sph(const oracle_model *m, double *x, double *y, double *z, double u, double v, double w,
                             int *icell, int *lintersect) {
  const double correct_moins = 1.0 - 1.0e-10;
  const double x0 = *x, y0 = *y, z0 = *z;
  const double r0_2 = x0 * x0 + y0 * y0 + z0 * z0;
  const double b = (x0 * u + y0 * v + z0 * w);
  const double c = (r0_2 - m->r_lim_2[m->n_rad] * correct_moins);
  const double delta = b * b - c;
  if (delta < 0.0) {
    *lintersect = 0;
    *icell = 0;
    return;
  }
  *lintersect = 1;
  const double rac = sqrt(delta);
  const double s1 = -b - rac;
  const double x1 = x0 + s1 * u, y1 = y0 + s1 * v, z1 = z0 + s1 * w;
  oracle_index_cell_sph(m, x1, y1, z1, icell);
  *x = x1; *y = y1; *z = z1;
}

/* spherical_grid.f90:619-699 */
void oracle_pos_em_cell_sph(const oracle_model *m, int icell, float rand1, float rand2, float rand3, double *x,
                            double *y, double *z) {
  const int ri = m->cell_map_i[icell - 1], thetaj = m->cell_map_j[icell - 1], phik = m->cell_map_k[icell - 1];
  const double one_third = 1.0 / 3.0;
  const double r = pow(m->r_lim_3[ri - 1] + (double)rand1 * (m->r_lim_3[ri] - m->r_lim_3[ri - 1]), one_third);
  double theta;
  if (m->l3D) {
    const int aj = thetaj < 0 ? -thetaj : thetaj;
    theta = m->theta_lim[aj - 1] + (double)rand2 * (m->theta_lim[aj] - m->theta_lim[aj - 1]);
  } else {
    if ((double)rand2 > 0.5)
      theta = m->theta_lim[thetaj - 1] + (2.0 * ((double)rand2 - 0.5)) * (m->theta_lim[thetaj] - m->theta_lim[thetaj - 1]);
    else
      theta = -(m->theta_lim[thetaj - 1] + (2.0 * (double)rand2) * (m->theta_lim[thetaj] - m->theta_lim[thetaj - 1]));
  }
  const double phi = 2.0 * PI * ((double)(float)phik - 1.0 + (double)rand3) / (double)(float)m->n_az;
  *z = r * sin(theta);
  const double r_cos_theta = r * cos(theta);
  *x = r_cos_theta * cos(phi);
  *y = r_cos_theta * sin(phi);
}

/* grid operator table (grid.f90:16-22, 298-357) */
static inline int grid_test_exit(const oracle_model *m, int icell, double x, double y, double z) {
  if (m->grid_type == 3) return icell < 0; /* test_exit_grid_Voronoi (:1446) */
  if (m->grid_type == 2) return oracle_test_exit_grid_sph(m, icell);
  return oracle_test_exit_grid_cyl(m, icell, x, y, z);
}
/* ... the same table for the ray tracer (cylindrical and spherical grids) */
static inline void grid_cross_cell(const oracle_model *m, double x0, double y0, double z0, double u, double v, double w, int icell,
                                   int previous_cell, double *x1, double *y1, double *z1, int *next_cell, double *l,
                                   double *l_contrib, double *l_void_before) {
  if (m->grid_type == 3) oracle_cross_voronoi_cell(m, x0, y0, z0, u, v, w, icell, previous_cell, x1, y1, z1, next_cell, l, l_contrib, l_void_before);
  else if (m->grid_type == 2) oracle_cross_spherical_cell(m, x0, y0, z0, u, v, w, icell, previous_cell, x1, y1, z1, next_cell, l, l_contrib, l_void_before);
  else oracle_cross_cylindrical_cell(m, x0, y0, z0, u, v, w, icell, previous_cell, x1, y1, z1, next_cell, l, l_contrib, l_void_before);
}
static inline void grid_move_to_grid(const oracle_model *m, double *x, double *y, double *z, double u, double v, double w, int *icell,
                                     int *lintersect) {
  if (m->grid_type == 3) oracle_move_to_grid_voronoi(m, x, y, z, u, v, w, icell, lintersect);
  else if (m->grid_type == 2) oracle_move_to_grid_sph(m, x, y, z, u, v, w, icell, lintersect);
  else oracle_move_to_grid_cyl(m, x, y, z, u, v, w, icell, lintersect);
}
static inline void grid_index_cell(const oracle_model *m, double x, double y, double z, int *icell) {
  if (m->grid_type == 3) oracle_index_cell_voronoi(m, x, y, z, icell);
  else if (m->grid_type == 2) oracle_index_cell_sph(m, x, y, z, icell);
  else oracle_index_cell_cyl(m, x, y, z, icell);
}

/* ------------------------------------------------------------------------ */
/* Direction helpers                                                         */
/* ------------------------------------------------------------------------ */

/* utils.f90:1636-1690 */
void oracle_cdapres(double cospsi, double phi, double u0, double v0, double w0,
                    double *u1, double *v1, double *w1) {
  double cpsi = cospsi;
  double spsi = sqrt(1.0 - cpsi * cpsi);
  double sphi = sin(phi);
  double cphi = cos(phi);
  double a = spsi * cphi;
  double b = spsi * sphi;
  if (fabs(w0) <= (double)0.999999f) {
    double c = sqrt(1.0 - w0 * w0);
    double cm1 = 1.0 / c;
    double aw0 = a * w0;
    *u1 = (aw0 * u0 - b * v0) * cm1 + cpsi * u0;
    *v1 = (aw0 * v0 + b * u0) * cm1 + cpsi * v0;
    *w1 = cpsi * w0 - a * c;
  } else {
    *u1 = a;
    *v1 = b;
    *w1 = cpsi;
  }
}

/* utils.f90:553-601 */
void oracle_rotation(double xinit, double yinit, double zinit, double u1,
                     double v1, double w1, double *xfin, double *yfin,
                     double *zfin) {
  double cost, sint, sing, prod, theta;
  if (w1 > 0.999999999) {
    cost = 1.0; sint = 0.0; sing = 0.0;
  } else {
    if (fabs(u1) < TINY_REAL) {
      cost = 0.0; sint = 1.0;
      sing = sqrt(1.0 - w1 * w1);
    } else {
      theta = atan2(v1, u1);
      cost = cos(theta);
      sint = sin(theta);
      sing = sqrt(1.0 - w1 * w1);
    }
  }
  prod = cost * xinit + sint * yinit;
  *xfin = sing * prod + w1 * zinit;
  *yfin = cost * yinit - sint * xinit;
  *zfin = sing * zinit - w1 * prod;
}

/* random_numbers.f90:32-51 */
static void random_isotropic_direction(rng_t *r, double *u, double *v,
                                       double *w) {
  float rand = rng_float(r);
  *w = 2.0 * (double)rand - 1.0;
  double uv = sqrt(1.0 - (*w) * (*w));
  rand = rng_float(r);
  double phi = PI * (2.0 * (double)rand - 1.0);
  *u = uv * cos(phi);
  *v = uv * sin(phi);
}

/* scattering.f90:1354-1383 */
void oracle_hg(float g, float rand, int nang_scatt, int *itheta,
               double *cospsi) {
  double rand_dp = fmin((double)rand, 1.0 - 1e-6);
  if (fabsf(g) > FLT_MIN) {
    double g1 = (double)g;
    double g2 = g1 * g1;
    double q = (1.0 - g2) / (1.0 - g1 + 2.0 * g1 * rand_dp);
    *cospsi = (1.0 + g2 - q * q) / (2.0 * g1);
  } else {
    *cospsi = 2.0 * rand_dp - 1.0;
  }
  *itheta = (int)floor(acos(*cospsi) * 180.0 / PI) + 1;
  if (*itheta > nang_scatt) *itheta = nang_scatt;
}

/* scattering.f90:1433-1475 */
void oracle_angle_diff_theta_pos(const oracle_model *m, int p_lambda,
                                 float rand, float rand2, int *itheta,
                                 double *cospsi) {
  const int na = m->nang_scatt;
  const float *prob = m->prob_s11_pos + (size_t)(na + 1) * (p_lambda - 1);
  int kmin = 0, kmax = na, k = (kmin + kmax) / 2;
  while ((kmax - kmin) > 1) {
    if (prob[k] < rand) kmin = k; else kmax = k;
    k = (kmin + kmax) / 2;
  }
  k = kmax;
  *itheta = k;
  double c0 = cos(((double)k - 1.0) * PI / (double)na);
  double c1 = cos(((double)k) * PI / (double)na);
  *cospsi = c0 + (double)rand2 * (c1 - c0);
}

/* scattering.f90:1328-1350 */
static void get_mueller_matrix_per_cell(const oracle_model *m, int lambda,
                                        int itheta, float frac, double M[16]) {
  const size_t o = (size_t)(m->nang_scatt + 1) * (lambda - 1);
  float frac_m1 = 1.0f - frac;
  memset(M, 0, 16 * sizeof(double));
  /* M is column-major M(i,j) -> M[(i-1)+4*(j-1)] */
#define MM(i, j) M[((i)-1) + 4 * ((j)-1)]
#define INTERP(t) ((t)[o + itheta] * frac + (t)[o + itheta - 1] * frac_m1)
  MM(1, 1) = 1.0;
  MM(2, 2) = (double)INTERP(m->s22_o_s11);
  MM(1, 2) = (double)INTERP(m->s12_o_s11);
  MM(2, 1) = MM(1, 2);
  MM(3, 3) = (double)INTERP(m->s33_o_s11);
  MM(4, 4) = (double)INTERP(m->s44_o_s11);
  MM(3, 4) = (double)(-m->s34_o_s11[o + itheta] * frac -
                      m->s34_o_s11[o + itheta - 1] * frac_m1);
  MM(4, 3) = -MM(3, 4);
#undef INTERP
}

/* ---- scattering method 1: the scattering grain is drawn, then its own phase function (dust_transfer.f90:1288-1316) ---- */
/* select_scattering_grain (dust_prop.f90:1292-1336), low_mem_scattering: the CDF of C_sca n over the grain sizes of the
 * cell is walked on the fly, from the small grains when rand < 0.5, else from the big ones.  (A walk that rounding lets
 * run off the end would index out of bounds in the reference: the last grain visited is returned here.) */
/* ksca_CDF (dust_prop.f90:976-994): what `opacity` stores per (cell class, wavelength) when scattering method 1 has the
 * memory for it -- the running sum of C_sca n over the grain sizes, normalised, or all ones where it is not positive */
void oracle_build_ksca_CDF(const oracle_model *m, double *ksca_CDF) {
  const int ng = m->m1_n_grains, nc = m->p_n_cells ? m->p_n_cells : 1, nl = m->n_lambda;
  for (int l = 0; l < nl; ++l)
    for (int c = 0; c < nc; ++c) {
      double *row = ksca_CDF + (size_t)(ng + 1) * ((size_t)c + (size_t)nc * l);
      row[0] = 0.0;
      for (int k = 1; k <= ng; ++k)
        row[k] = row[k - 1] + (double)m->m1_C_sca[(size_t)(k - 1) + (size_t)ng * l] * m->m1_dens[(size_t)(k - 1) + (size_t)ng * c] *
                 m->m1_nk[k - 1];
      const double last = row[ng];
      if (last > (double)1.17549435082228750797e-38f) { for (int k = 0; k <= ng; ++k) row[k] = row[k] / last; }
      else { for (int k = 0; k <= ng; ++k) row[k] = 1.0; }
    }
}

/* select_grainsize_high_mem (dust_prop.f90:1245-1288) */
static int select_grainsize_high_mem(const oracle_model *m, int lambda, int icell, float rand) {
  const int ng = m->m1_n_grains, nc = m->p_n_cells ? m->p_n_cells : 1;
  const int cls = m->p_n_cells ? m->p_icell[icell - 1] - 1 : 0;
  const double *cdf = m->m1_ksca_CDF + (size_t)(ng + 1) * ((size_t)cls + (size_t)nc * (lambda - 1));
  const float prob = rand;
  int kmin = 0, kmax = ng, k = (kmin + kmax) / 2;
  while (cdf[k] != (double)prob) {
    if (cdf[k] < (double)prob) kmin = k; else kmax = k;
    k = (kmin + kmax) / 2;
    if ((kmax - kmin) <= 1) break;
  }
  return kmax;
}

int oracle_select_scattering_grain(const oracle_model *m, int lambda, int icell, float rand) {
  if (m->m1_ksca_CDF) return select_grainsize_high_mem(m, lambda, icell, rand);   /* .not. low_mem_scattering (:1330) */
  const int ng = m->m1_n_grains;
  const double AU_to_cm = 149597870700.0 * 100.0, mum_to_cm = 1.0e-4;
  const double norm = tab_kappa(m, icell, lambda) * (double)tab_albedo(m, icell, lambda) / (AU_to_cm * (mum_to_cm * mum_to_cm));
  const int cls = m->p_n_cells ? m->p_icell[icell - 1] - 1 : 0;
  const double *d = m->m1_dens + (size_t)ng * cls;
  const float *Cs = m->m1_C_sca + (size_t)ng * (lambda - 1);
  double CDF = 0.0;
  int k;
  if (rand < 0.5f) {
    const double prob = (double)rand * norm;
    for (k = 1; k <= ng; ++k) {
      const double density = d[k - 1] * m->m1_nk[k - 1];
      CDF = CDF + (double)Cs[k - 1] * density;
      if (CDF > prob) break;
    }
    if (k > ng) k = ng;
  } else {
    const double prob = (double)(1.0f - rand) * norm;
    for (k = ng; k >= 1; --k) {
      const double density = d[k - 1] * m->m1_nk[k - 1];
      CDF = CDF + (double)Cs[k - 1] * density;
      if (CDF > prob) break;
    }
    if (k < 1) k = 1;
  }
  return k;
}

/* angle_diff_theta (scattering.f90:1387-1429): prob_s11(lambda, igrain, 0:nang) */
static void angle_diff_theta_grain(const oracle_model *m, int lambda, int igrain, float rand, float rand2, int *itheta,
                                   double *cospsi) {
  const int na = m->nang_scatt;
  const size_t st = (size_t)m->n_lambda * m->m1_n_grains;   /* stride of the angle */
  const float *prob = m->m1_prob_s11 + (size_t)(lambda - 1) + (size_t)m->n_lambda * (igrain - 1);
  int kmin = 0, kmax = na, k = (kmin + kmax) / 2;
  while ((kmax - kmin) > 1) {
    if (prob[st * k] < rand) kmin = k; else kmax = k;
    k = (kmin + kmax) / 2;
  }
  k = kmax;
  *itheta = k;
  const double c0 = cos(((double)k - 1.0) * PI / (double)na), c1 = cos(((double)k) * PI / (double)na);
  *cospsi = c0 + (double)rand2 * (c1 - c0);
}

/* get_Mueller_matrix_per_grain (scattering.f90:1302-1324): tab_s1x(0:nang, n_grains, n_lambda), s11 included */
static void get_mueller_matrix_per_grain(const oracle_model *m, int lambda, int itheta, float frac, int igrain, double M[16]) {
  const size_t o = (size_t)(m->nang_scatt + 1) * ((size_t)(igrain - 1) + (size_t)m->m1_n_grains * (lambda - 1));
  float frac_m1 = 1.0f - frac;
  memset(M, 0, 16 * sizeof(double));
#define MM(i, j) M[((i)-1) + 4 * ((j)-1)]
#define INTERP(t) ((t)[o + itheta] * frac + (t)[o + itheta - 1] * frac_m1)
  MM(1, 1) = (double)INTERP(m->m1_s11);
  MM(2, 2) = (double)INTERP(m->m1_s22);
  MM(1, 2) = (double)INTERP(m->m1_s12);
  MM(2, 1) = MM(1, 2);
  MM(3, 3) = (double)INTERP(m->m1_s33);
  MM(4, 4) = (double)INTERP(m->m1_s44);
  MM(3, 4) = (double)(-m->m1_s34[o + itheta] * frac - m->m1_s34[o + itheta - 1] * frac_m1);
  MM(4, 3) = -MM(3, 4);
#undef INTERP
#undef MM
}

/* scattering.f90:1187-1298 */
void oracle_update_stokes(double S[4], double u0, double v0, double w0,
                          double u1, double v1, double w1, const double M[16]) {
  float sinw, cosw, omega, theta, costhet, xnyp;
  double v1pi, v1pj, v1pk, S1_0;
  double C[4], D[4];
  oracle_rotation(u0, v0, w0, u1, v1, w1, &v1pi, &v1pj, &v1pk);
  xnyp = (float)sqrt(v1pk * v1pk + v1pj * v1pj);
  if (xnyp < 1e-10f) {
    xnyp = 0.0f;
    costhet = 1.0f;
  } else {
    costhet = (float)(-1.0 * v1pj / (double)xnyp);
  }
  theta = acosf(costhet);
  if ((double)theta >= PI) theta = 0.0f;
  theta = (float)((double)theta + 0.5 * PI);
  omega = 2.0f * theta;
  if (v1pk < 0.0) omega = -1.0f * omega;
  cosw = cosf(omega);
  sinw = sinf(omega);
  if (fabsf(cosw) < 1e-06f) cosw = 0.0f;
  if (fabsf(sinw) < 1e-06f) sinw = 0.0f;
  /* ROP: (2,2)=cosw (3,2)=sinw (2,3)=-sinw (3,3)=cosw ; C = ROP*S */
  C[0] = S[0];
  C[1] = (double)cosw * S[1] - (double)sinw * S[2];
  C[2] = (double)sinw * S[1] + (double)cosw * S[2];
  C[3] = S[3];
  for (int i = 0; i < 4; ++i) {
    D[i] = 0.0;
    for (int j = 0; j < 4; ++j) D[i] += M[i + 4 * j] * C[j];
  }
  S1_0 = S[0];
  /* RPO: (2,2)=cosw (2,3)=sinw (3,2)=-sinw (3,3)=cosw ; S = RPO*D */
  S[0] = D[0];
  S[1] = (double)cosw * D[1] + (double)sinw * D[2];
  S[2] = -(double)sinw * D[1] + (double)cosw * D[2];
  S[3] = D[3];
  if (S[0] > TINY_REAL) {
    double f = M[0] * S1_0 / S[0];
    for (int i = 0; i < 4; ++i) S[i] *= f;
  }
}

/* thermal_emission.f90:364-400 */
void oracle_select_wl_em(const oracle_model *m, float rand, int *lambda) {
  const double *cum = m->spectre_emission_cumul;
  int kmin = 0, kmax = m->n_lambda, k = (kmin + kmax) / 2;
  while (cum[k] != (double)rand) {
    if (cum[k] < (double)rand) kmin = k; else kmax = k;
    k = (kmin + kmax) / 2;
    if ((kmax - kmin) <= 1) break;
  }
  *lambda = kmax;
}

/* stars.f90:75-104 */
static int select_star(const oracle_model *m, int lambda, float rand) {
  int kmin = 0, kmax = m->n_stars, k = (kmax - kmin) / 2;
  while ((kmax - kmin) > 1) {
    /* CDF_E_star(lambda,k), k = 0..n_stars */
    if (m->CDF_E_star[(lambda - 1) + (size_t)m->n_lambda * k] < (double)rand)
      kmin = k;
    else
      kmax = k;
    k = (kmin + kmax) / 2;
  }
  return kmax;
}

/* thermal_emission.f90:2044-2073 */
static int select_cellule(const oracle_model *m, int lambda, float rand) {
  const double *p = m->prob_E_cell + (size_t)(m->n_cells + 1) * (lambda - 1);
  int kmin = 0, kmax = m->n_cells, k = (kmin + kmax) / 2;
  while ((kmax - kmin) > 1) {
    if (p[k] < (double)rand) kmin = k; else kmax = k;
    k = (kmin + kmax) / 2;
  }
  return kmax;
}

/* stars.f90:812-884 */
void oracle_intersect_stars(const oracle_model *m, double x, double y,
                            double z, double u, double v, double w,
                            int *lintersect, int *i_star, int *icell_star) {
  double d_to_star = DBL_MAX;
  *i_star = 0;
  for (int i = 1; i <= m->n_stars; ++i) {
    const oracle_star *st = &m->stars[i - 1];
    double dx = x - st->x, dy = y - st->y, dz = z - st->z;
    double b = dx * u + dy * v + dz * w;
    double c = dx * dx + dy * dy + dz * dz - st->r * st->r;
    double delta = b * b - c;
    if (delta >= 0.0) {
      double rac = sqrt(delta);
      double s1 = -b - rac;
      if (s1 < 0) {
        double s2 = -b + rac;
        if (s2 > 0) {
          d_to_star = 0.0;
          *i_star = i;
        }
      } else {
        if (s1 < d_to_star) {
          d_to_star = s1;
          *i_star = i;
        }
      }
    }
  }
  *lintersect = (*i_star > 0);
  *icell_star = *lintersect ? m->stars[*i_star - 1].icell : 0;
}

/* stars.f90:108-169 */
static void emit_packet_uniform_sphere(const oracle_model *m, int i_star,
                                       float rand1, float rand2, float rand3,
                                       float rand4, int *icell, double *x,
                                       double *y, double *z, double *u,
                                       double *v, double *w, int *lintersect) {
  const oracle_star *st = &m->stars[i_star - 1];
  *z = 2.0 * (double)rand1 - 1.0;
  double srw02 = sqrt(1.0 - (*z) * (*z));
  double argmt = PI * (2.0 * (double)rand2 - 1.0);
  *x = srw02 * cos(argmt);
  *y = srw02 * sin(argmt);
  double cospsi = sqrt((double)rand3);
  double phi = 2.0 * PI * (double)rand4;
  oracle_cdapres(cospsi, phi, *x, *y, *z, u, v, w);
  double r_star = st->r * (1.0 + 1e-6);
  *x = *x * r_star + st->x;
  *y = *y * r_star + st->y;
  *z = *z * r_star + st->z;
  if (m->grid_type == 3) *icell = st->icell; /* stars.f90:155-156 */
  else if (m->grid_type == 2) oracle_index_cell_sph(m, *x, *y, *z, icell);
  else oracle_index_cell_cyl(m, *x, *y, *z, icell);
  if (st->out_model) {
    if (m->grid_type == 3) oracle_move_to_grid_voronoi(m, x, y, z, *u, *v, *w, icell, lintersect);
    else if (m->grid_type == 2) oracle_move_to_grid_sph(m, x, y, z, *u, *v, *w, icell, lintersect);
    else oracle_move_to_grid_cyl(m, x, y, z, *u, *v, *w, icell, lintersect);
  } else {
    *lintersect = 1;
  }
}

/* ------------------------------------------------------------------------ */
/* Temperature                                                               */
/* ------------------------------------------------------------------------ */

/* Temp_LTE (thermal_emission.f90:649-706) given the cell's heating integral
 * sum_k kappa_abs*l (Qheat = E*L_packet_th/volume is formed here).  Ti_start
 * is the cached xT_ech entry (>= 2).  When the cell is at T_min the reference
 * leaves `frac` unassigned (:674,:679); the oracle returns frac = 0, i.e. the
 * T_1 row of the CDF. */
static void temp_lte_tab(const oracle_model *m, const double *lq, double E_scaled, double volume,
                         int Ti_start, int *Ti_out, float *Temp, double *frac);
void oracle_temp_lte(const oracle_model *m, double E_scaled, double volume,
                     int Ti_start, int *Ti_out, float *Temp, double *frac) {
  temp_lte_tab(m, m->log_Qcool, E_scaled, volume, Ti_start, Ti_out, Temp, frac);
}
/* lq = log_Qcool_minus_extra_heating(:, p_icell), 1-based: lq[T-1] */
static void temp_lte_tab(const oracle_model *m, const double *lq, double E_scaled, double volume,
                         int Ti_start, int *Ti_out, float *Temp, double *frac) {
  double Qheat = E_scaled * m->L_packet_th / volume;
  int Ti;
  *frac = 0.0;
  if (Qheat < TINY_DP) {
    *Temp = m->T_min; Ti = 2;
  } else {
    double log_Qheat = log(Qheat);
    if (log_Qheat < lq[0]) {
      *Temp = m->T_min; Ti = 2;
    } else {
      Ti = Ti_start;
      while ((lq[Ti - 1] < log_Qheat) && (Ti < m->n_T)) Ti++;
      *frac = (log_Qheat - lq[Ti - 2]) / (lq[Ti - 1] - lq[Ti - 2]);
      /* log of a default real is a default-real log (:697) */
      *Temp = (float)exp((double)logf(m->tab_Temp[Ti - 1]) * (*frac) +
                         (double)logf(m->tab_Temp[Ti - 2]) * (1.0 - *frac));
    }
  }
  *Ti_out = Ti;
}

/* Temp_finale (thermal_emission.f90:870-906): id = 0, E_abs already summed */
void oracle_temp_finale(const oracle_model *m, const double *E_abs,
                        float *Tdust) {
  for (int icell = 1; icell <= m->n_cells; ++icell) {
    int Ti; double frac;
    temp_lte_tab(m, tab_log_Qcool(m, icell), E_abs[icell - 1], m->volume[icell - 1], 2, &Ti,
                 &Tdust[icell - 1], &frac);
  }
}

/* ------------------------------------------------------------------------ */
/* Packet loop                                                               */
/* ------------------------------------------------------------------------ */
#define ORACLE_MAX_RT 64
typedef struct {
  const oracle_model *m;
  const oracle_opts *o;
  const double *E_prior;
  double *E_abs;   /* this thread's xKJ_abs(:,id) */
  int *xT_ech;     /* this thread's xT_ech(:,id) */
  double *sed;     /* this thread's sed arrays */
  double *n_sent;
  uint64_t cnt[ORACLE_N_COUNTERS];
  double qscale;   /* nb_proc * n_replicas */
  rng_t rng;
  /* SED mode (lmono): NULL in the thermal step */
  const oracle_mono_opts *mono;
  double *xI;          /* shared xI_scatt (atomic adds) */
  int flag_direct_star; /* starlight that has not interacted yet (dust_transfer.f90:1189-1193, 1262) */
  int itheta_rt1[ORACLE_MAX_RT];      /* dust_ray_tracing.f90:39 */
  double cos_omega_rt1[ORACLE_MAX_RT], sin_omega_rt1[ORACLE_MAX_RT]; /* :40 */
} __attribute__((aligned(256))) worker_t; /* one cache-line group per thread: no false sharing */

/* angles_scatt_rt1 (dust_ray_tracing.f90:409-476) */
static void angles_scatt_rt1(worker_t *W, double u, double v, double w) {
  const oracle_model *m = W->m;
  for (int ibin = 1; ibin <= m->RT_n_incl; ++ibin)
    for (int iaz = 1; iaz <= m->RT_n_az; ++iaz) {
      const int q = (ibin - 1) + m->RT_n_incl * (iaz - 1);
      const double ur = m->tab_u_rt[q], vr = m->tab_v_rt[q], wr = m->tab_w_rt[ibin - 1];
      const float cos_scatt = (float)(ur * u + vr * v + wr * w);
      /* nint(acos(cos_scatt) * real(nang_scatt)/pi): default-real acos and product, then / pi in
       * dp.  The default-real acos is taken as the correctly rounded one ((float)acos(double)) so
       * that every libm gives the same index; |cos_scatt| > 1 by rounding gives NaN, which the
       * reference's nint turns into a value < 1, i.e. k = 1. */
      const float ac = (float)acos((double)cos_scatt);
      int k;
      if (ac != ac) k = 1;
      else k = (int)lround((double)(ac * (float)m->nang_scatt) / PI);
      if (k > m->nang_scatt) k = m->nang_scatt;
      if (k < 1) k = 1;
      W->itheta_rt1[q] = k;
      if (m->lsepar_pola) {
        double v1pi, v1pj, v1pk;
        oracle_rotation(u, v, w, -ur, -vr, -wr, &v1pi, &v1pj, &v1pk);
        double xnyp = sqrt(v1pk * v1pk + v1pj * v1pj), costhet;
        if (xnyp < 1e-10) { xnyp = 0.0; costhet = 1.0; }
        else costhet = -1.0 * v1pj / xnyp;
        double theta = acos(costhet);
        if (theta >= PI) theta = 0.0;
        theta = theta + PI / 2;
        double omega = 2.0 * theta;
        if (v1pk < 0.0) omega = -1.0 * omega;
        double cosw = cos(omega), sinw = sin(omega);
        if (fabs(cosw) < 1e-06) cosw = 0.0;
        if (fabs(sinw) < 1e-06) sinw = 0.0;
        W->cos_omega_rt1[q] = cosw;
        W->sin_omega_rt1[q] = sinw;
      }
    }
}

static inline void xI_add(worker_t *W, int phik, int psup, int type, int iRT, int icell, double v) {
  const oracle_model *m = W->m;
  const size_t idx = (size_t)(phik - 1) + (size_t)m->n_az_rt * ((psup - 1) + (size_t)m->n_theta_rt * ((type - 1) +
                     (size_t)m->N_type_flux * ((iRT - 1) + (size_t)(m->RT_n_incl * m->RT_n_az) * (size_t)(icell - 1))));
#ifdef _OPENMP
#pragma omp atomic
#endif
  W->xI[idx] += v;
}

/* save_radiation_field, lscatt_ray_tracing1 branch (radiation_field.f90:63-89) with
 * calc_xI_scatt (dust_ray_tracing.f90:480-529) / calc_xI_scatt_pola (:533-632) */
static void save_radiation_field_rt1(worker_t *W, int icell, const double Stokes[4], double l,
                                     double x0, double y0, double z0, double x1, double y1, double z1,
                                     int flag_star) {
  const oracle_model *m = W->m;
  const int p_lambda = W->mono->p_lambda;
  const double xm = 0.5 * (x0 + x1), ym = 0.5 * (y0 + y1), zm = 0.5 * (z0 + z1);
  int phik, psup;
  if (m->l3D) { phik = 1; psup = 1; }
  else {
    const double phi_pos = atan2(xm, ym);
    phik = (int)floor(modulo_d(phi_pos, 2 * PI) / (2 * PI) * (double)m->n_az_rt) + 1;
    if (phik > m->n_az_rt) phik = m->n_az_rt;
    psup = (zm > 0.0) ? 1 : 2;
  }
  const size_t na1 = (size_t)m->nang_scatt + 1;
  /* tab_s11_pos(it, p_icell, p_lambda) (dust_ray_tracing.f90:503-512): lvariable_dust reads the crossed cell's own tables */
  const int vd = m->p_n_cells != 0;
  const size_t col = vd ? na1 * ((size_t)(p_lambda - 1) * m->p_n_cells + (size_t)(m->p_icell[icell - 1] - 1)) : na1 * (size_t)(p_lambda - 1);
  const float *t_s11 = vd ? m->v_tab_s11_pos : m->tab_s11_pos;
  const float *t_s12 = vd ? m->v_s12_o_s11 : m->s12_o_s11, *t_s22 = vd ? m->v_s22_o_s11 : m->s22_o_s11;
  const float *t_s33 = vd ? m->v_s33_o_s11 : m->s33_o_s11, *t_s34 = vd ? m->v_s34_o_s11 : m->s34_o_s11;
  const float *t_s44 = vd ? m->v_s44_o_s11 : m->s44_o_s11;
  for (int ibin = 1; ibin <= m->RT_n_incl; ++ibin)
    for (int iaz = 1; iaz <= m->RT_n_az; ++iaz) {
      const int q = (ibin - 1) + m->RT_n_incl * (iaz - 1), iRT = q + 1; /* RT2d_to_RT1d (:66-76) */
      const int it = W->itheta_rt1[q];
      const float s11 = t_s11[col + it];
      if (!m->lsepar_pola) {
        const double flux = l * Stokes[0] * (double)s11;
        xI_add(W, phik, psup, 1, iRT, icell, flux);
        if (m->lsepar_contrib) xI_add(W, phik, psup, flag_star ? 3 : 5, iRT, icell, flux); /* n_Stokes = 1 */
        continue;
      }
      const float s12 = -s11 * t_s12[col + it], s22 = s11 * t_s22[col + it];
      const float s33 = -s11 * t_s33[col + it], s34 = -s11 * t_s34[col + it];
      const float s44 = -s11 * t_s44[col + it];
      const double cosw = W->cos_omega_rt1[q], sinw = W->sin_omega_rt1[q];
      /* C = ROP * Stokes, ROP(2:3,2:3) = [[cosw, -sinw], [sinw, cosw]] */
      const double C1 = Stokes[0], C4 = Stokes[3];
      const double C2 = cosw * Stokes[1] + (-sinw) * Stokes[2];
      const double C3 = sinw * Stokes[1] + cosw * Stokes[2];
      /* D = M * C */
      const double D1 = (double)s11 * C1 + (double)s12 * C2;
      const double D2 = (double)s12 * C1 + (double)s22 * C2;
      const double D3 = (double)s33 * C3 + (double)(-s34) * C4;
      const double D4 = (double)s34 * C3 + (double)s44 * C4;
      /* S = RPO * D, RPO(2:3,2:3) = [[-cosw, -sinw], [-sinw, cosw]] */
      const double S1 = D1, S4 = D4;
      const double S2 = (-cosw) * D2 + (-sinw) * D3;
      const double S3 = (-sinw) * D2 + cosw * D3;
      xI_add(W, phik, psup, 1, iRT, icell, l * S1);
      xI_add(W, phik, psup, 2, iRT, icell, l * S2);
      xI_add(W, phik, psup, 3, iRT, icell, l * S3);
      xI_add(W, phik, psup, 4, iRT, icell, l * S4);
      if (m->lsepar_contrib) xI_add(W, phik, psup, flag_star ? 6 : 8, iRT, icell, l * S1);
    }
}

/* save_radiation_field, lscatt_ray_tracing2 branch (radiation_field.f90:91-129; 2D): I_spec(N_type_flux, n_theta_I,
 * n_phi_I, n_cells) and I_spec_star(n_cells), summed over the threads (atomic adds into the shared arrays) */
static void save_radiation_field_rt2(worker_t *W, int icell, const double Stokes[4], double l, double x0, double y0,
                                     double z0, double x1, double y1, double z1, double u, double v, double w,
                                     int flag_star) {
  const oracle_model *m = W->m;
  const oracle_mono_opts *o = W->mono;
  const int n_Stokes = m->lsepar_pola ? 4 : 1, ntf = n_Stokes + (m->lsepar_contrib ? 4 : 0);
  if (W->flag_direct_star) {
#ifdef _OPENMP
#pragma omp atomic
#endif
    o->I_spec_star[icell - 1] += l * Stokes[0];
    return;
  }
  const double xm = 0.5 * (x0 + x1), ym = 0.5 * (y0 + y1), zm = 0.5 * (z0 + z1);
  const double phi_pos = atan2(xm, ym);
  const double phi_vol = atan2(-u, -v) + 2 * PI;
  int phi_I = (int)floor(modulo_d(phi_vol - phi_pos, 2 * PI) / (2 * PI) * (double)o->n_phi_I) + 1;
  if (phi_I > o->n_phi_I) phi_I = 1;
  int theta_I;
  if (zm > 0.0) theta_I = (int)floor(0.5 * (w + 1.0) * (double)o->n_theta_I) + 1;
  else theta_I = (int)floor(0.5 * (-w + 1.0) * (double)o->n_theta_I) + 1;
  if (theta_I > o->n_theta_I) theta_I = o->n_theta_I;
  double *rec = o->I_spec + (size_t)ntf * ((size_t)(theta_I - 1) + (size_t)o->n_theta_I * ((size_t)(phi_I - 1) + (size_t)o->n_phi_I * (icell - 1)));
  for (int t = 0; t < n_Stokes; ++t) {
#ifdef _OPENMP
#pragma omp atomic
#endif
    rec[t] += l * Stokes[t];
  }
  if (m->lsepar_contrib) {
    double *c = rec + (flag_star ? n_Stokes + 1 : n_Stokes + 3); /* n_Stokes + 2 / + 4, 1-based */
#ifdef _OPENMP
#pragma omp atomic
#endif
    *c += l * Stokes[0];
  }
}

/* physical_length (optical_depth.f90:21-182), letape_th branch only */
/* optional accumulators of save_radiation_field's thermal branch (radiation_field.f90:54-55), shared by the threads */
static double *g_xN_abs = NULL, *g_xJ_abs = NULL;
void oracle_set_radiation_field_outputs(double *xN_abs, double *xJ_abs) { g_xN_abs = xN_abs; g_xJ_abs = xJ_abs; }

static void physical_length(worker_t *W, int lambda, const double Stokes[4],
                            int *icell, double *xio, double *yio, double *zio,
                            double *u, double *v, double *w, int flag_star, double extrin,
                            int *flag_sortie, int *lpacket_alive) {
  const oracle_model *m = W->m;
  double x0 = *xio, y0 = *yio, z0 = *zio, x1 = *xio, y1 = *yio, z1 = *zio;
  double x_old, y_old, z_old, extr = extrin, l, tau, opacity, l_contrib,
         l_void_before;
  int icell_old, next_cell = *icell, previous_cell, icell0 = 0;
  int lintersect_stars, i_star, icell_star, lstop = 0, lcell_not_empty;
  *flag_sortie = 0;
  W->cnt[ORC_CNT_FLIGHTS]++;
  if (W->mono && W->mono->rt1 == 1) angles_scatt_rt1(W, *u, *v, *w);   /* :65 */

  oracle_intersect_stars(m, x0, y0, z0, *u, *v, *w, &lintersect_stars, &i_star,
                         &icell_star);                              /* :68 */
  for (;;) {
    icell_old = icell0;                                             /* :79 */
    x_old = x0; y_old = y0; z_old = z0;
    x0 = x1; y0 = y1; z0 = z1;
    previous_cell = icell0;
    icell0 = next_cell;

    if (grid_test_exit(m, icell0, x0, y0, z0)) {                    /* :87 */
      *flag_sortie = 1;
      return;
    }
    if (lintersect_stars && icell0 == icell_star) {                 /* :91 */
      *lpacket_alive = 0;
      *flag_sortie = 1;
      W->cnt[ORC_CNT_KILLED_STAR]++;
      return;
    }
    if (icell0 <= m->n_cells && icell0 >= 1) {                      /* :100 */
      lcell_not_empty = 1;
      opacity = tab_kappa(m, icell0, lambda) * m->kappa_factor[icell0 - 1];
      if (m->l_dark_zone && m->l_dark_zone[icell0 - 1]) {           /* :104 */
        *u = -*u; *v = -*v; *w = -*w;
        *icell = icell_old;
        *xio = x_old; *yio = y_old; *zio = z_old;
        *flag_sortie = 0;
        W->cnt[ORC_CNT_DARK]++;
        return;
      }
    } else {
      lcell_not_empty = 0;
      opacity = 0.0;
    }
    if (m->grid_type == 3)
      oracle_cross_voronoi_cell(m, x0, y0, z0, *u, *v, *w, icell0, previous_cell, &x1, &y1, &z1,
                                &next_cell, &l, &l_contrib, &l_void_before);
    else if (m->grid_type == 2)
      oracle_cross_spherical_cell(m, x0, y0, z0, *u, *v, *w, icell0, previous_cell, &x1, &y1, &z1,
                                  &next_cell, &l, &l_contrib, &l_void_before);
    else
      oracle_cross_cylindrical_cell(m, x0, y0, z0, *u, *v, *w, icell0,
                                    previous_cell, &x1, &y1, &z1, &next_cell, &l,
                                    &l_contrib, &l_void_before);    /* :119 */
    W->cnt[ORC_CNT_CROSSINGS]++;
    tau = l_contrib * opacity;                                      /* :134 */
    if (tau > extr) {                                               /* :138 */
      lstop = 1;
      l_contrib = l_contrib * (extr / tau);
      l = l_void_before + l_contrib;
    } else {
      extr = extr - tau;
    }
    /* save_radiation_field (radiation_field.f90:31-135): thermal step :53, SED mode :63-89 */
    if (lcell_not_empty) {
      if (!W->mono) {
        W->E_abs[icell0 - 1] += tab_kappa_abs(m, icell0, lambda) * l_contrib * Stokes[0];
        if (g_xN_abs) { /* :55 (lmcfost_lib) */
#pragma omp atomic
          g_xN_abs[icell0 - 1] += 1.0;
        }
        if (g_xJ_abs) { /* :54 (lxJ_abs_step1) */
#pragma omp atomic
          g_xJ_abs[(size_t)(icell0 - 1) + (size_t)m->n_cells * (size_t)(lambda - 1)] += l_contrib * Stokes[0];
        }
      }
      else if (W->mono->rt1 == 1)
        save_radiation_field_rt1(W, icell0, Stokes, l_contrib, x0, y0, z0, x1, y1, z1, flag_star);
      else if (W->mono->rt1 == 2)
        save_radiation_field_rt2(W, icell0, Stokes, l_contrib, x0, y0, z0, x1, y1, z1, *u, *v, *w, flag_star);
    }
    if (lstop) {                                                    /* :153 */
      *flag_sortie = 0;
      *xio = x0 + l * (*u);
      *yio = y0 + l * (*v);
      *zio = z0 + l * (*w);
      *icell = icell0;
      if (m->l3D && m->grid_type != 3 && m->grid_type != 2) oracle_index_cell_cyl(m, *xio, *yio, *zio, icell); /* :162: lcylindrical only */
      return;
    }
  }
}

/* im_reemission_LTE (thermal_emission.f90:710-771) */
/* the cell's temperature as im_reemission_LTE sees it (thermal_emission.f90:660-708) */
static void cell_temperature(worker_t *W, int icell, int *Ti, float *Temp, double *frac_T2) {
  const oracle_model *m = W->m;
  const double *lq = tab_log_Qcool(m, icell);
  if (W->o->frozen) {
    temp_lte_tab(m, lq, W->E_prior[icell - 1], m->volume[icell - 1], 2, Ti,
                 Temp, frac_T2);
  } else {
    /* id > 0 branch (:670): partial sum * nb_proc, cached xT_ech (:685,:702) */
    temp_lte_tab(m, lq, W->E_abs[icell - 1] * W->qscale, m->volume[icell - 1],
                 W->xT_ech[icell - 1], Ti, Temp, frac_T2);
    W->xT_ech[icell - 1] = *Ti;
  }
}

static void im_reemission_LTE(worker_t *W, int icell, float rand1, float rand2,
                              int *lambda) {
  const oracle_model *m = W->m;
  (void)rand1;
  int Ti; float Temp; double frac_T2;
  cell_temperature(W, icell, &Ti, &Temp, &frac_T2);
  int T2 = Ti, T1 = Ti - 1;
  double frac_T1 = 1.0 - frac_T2;
  int l1 = 0, l2 = m->n_lambda, l = (l1 + l2) / 2;
  const double *cdf1 = tab_cdf(m, icell) + (size_t)m->n_lambda * (T1 - 1);
  const double *cdf2 = tab_cdf(m, icell) + (size_t)m->n_lambda * (T2 - 1);
  while ((l2 - l1) > 1) {
    double proba = frac_T1 * cdf1[l - 1] + frac_T2 * cdf2[l - 1];
    if ((double)rand2 > proba) l1 = l; else l2 = l;
    l = (l1 + l2) / 2;
  }
  *lambda = l + 1;
}

/* emit_packet (dust_transfer.f90:1047-1151) */
static int emit_packet(worker_t *W, int lambda, int *icell, double *x,
                       double *y, double *z, double *u, double *v, double *w,
                       double Stokes[4], int *flag_star, int *flag_ISM,
                       int *lintersect) {
  const oracle_model *m = W->m;
  *lintersect = 1;
  float rand = rng_float(&W->rng);
  if ((double)rand <= m->frac_E_stars[lambda - 1]) {
    *flag_star = 1; *flag_ISM = 0;
    rand = rng_float(&W->rng);
    int i_star = select_star(m, lambda, rand);
    float r1 = rng_float(&W->rng), r2 = rng_float(&W->rng);
    float r3 = rng_float(&W->rng), r4 = rng_float(&W->rng);
    emit_packet_uniform_sphere(m, i_star, r1, r2, r3, r4, icell, x, y, z, u, v,
                               w, lintersect);
    Stokes[0] = 1.0; Stokes[1] = 0.0; Stokes[2] = 0.0; Stokes[3] = 0.0;
  } else if ((double)rand <= m->frac_E_disk[lambda - 1]) {
    *flag_star = 0; *flag_ISM = 0;
    if (!m->prob_E_cell) return 11;
    rand = rng_float(&W->rng);
    *icell = select_cellule(m, lambda, rand);
    float r1 = rng_float(&W->rng), r2 = rng_float(&W->rng),
          r3 = rng_float(&W->rng);
    if (m->grid_type == 3) { /* pos_em_cell_voronoi (Voronoi.f90:1510-1542): the cell centre */
      const double *c = m->v_xyz_dp + 3 * (size_t)(*icell - 1);
      *x = c[0]; *y = c[1]; *z = c[2];
    } else if (m->grid_type == 2) {
      oracle_pos_em_cell_sph(m, *icell, r1, r2, r3, x, y, z);
    } else {
      oracle_pos_em_cell_cyl(m, *icell, r1, r2, r3, x, y, z);
    }
    random_isotropic_direction(&W->rng, u, v, w);
    Stokes[0] = 1.0; Stokes[1] = 0.0; Stokes[2] = 0.0; Stokes[3] = 0.0;
  } else { /* emit_packet_ISM (stars.f90:728-785) */
    *flag_star = 0; *flag_ISM = 1;
    if (!(m->R_ISM > 0.0)) return 12;
    float r1 = rng_float(&W->rng), r2 = rng_float(&W->rng);
    *z = 2.0 * (double)r1 - 1.0;
    const double srw02 = sqrt(1.0 - (*z) * (*z));
    const double argmt = PI * (2.0 * (double)r2 - 1.0);
    *x = srw02 * cos(argmt);
    *y = srw02 * sin(argmt);
    float r3 = rng_float(&W->rng), r4 = rng_float(&W->rng);
    const double cospsi = -sqrt((double)r3); /* towards the interior */
    const double phi = 2.0 * PI * (double)r4;
    oracle_cdapres(cospsi, phi, *x, *y, *z, u, v, w);
    *x = m->centre_ISM[0] + *x * m->R_ISM;
    *y = m->centre_ISM[1] + *y * m->R_ISM;
    *z = m->centre_ISM[2] + *z * m->R_ISM;
    Stokes[0] = 1.0; Stokes[1] = 0.0; Stokes[2] = 0.0; Stokes[3] = 0.0;
    if (m->grid_type == 3) oracle_move_to_grid_voronoi(m, x, y, z, *u, *v, *w, icell, lintersect);
    else if (m->grid_type == 2) oracle_move_to_grid_sph(m, x, y, z, *u, *v, *w, icell, lintersect);
    else oracle_move_to_grid_cyl(m, x, y, z, *u, *v, *w, icell, lintersect);
  }
  return 0;
}

/* propagate_packet (dust_transfer.f90:1155-1409), .not.lmono, lonly_LTE,
 * scattering method 2 */
/* ------------------------------------------------------------------------ */
/* Modified random walk (Min et al. 2009; Robitaille 2010).  The reference carries the pieces -- the zeta table     */
/* (MRW.f90:16-53), gamma_MRW = 2 (:11), cst_ct = 3/pi^2 (:12), the step (:74-115), distance_to_closest_wall_cyl     */
/* (cylindrical_grid.f90:1179), the trigger n_interactions_in_cell > 5 and the loop (dust_transfer.f90:1222-1239) --  */
/* but its step is an unfinished stub behind a commented-out call.  What follows is the algorithm those pieces and    */
/* their TODO comments describe, made to work: PARITY UNPINNED, checked against the brute-force loop.                 */
/* ------------------------------------------------------------------------ */
/* distance_to_closest_wall_sph (spherical_grid.f90:451-499): the shells and -- the working form of what the reference
 * sketches: it multiplies z0 with cos_phi_lim(thetaj0), the table of the AZIMUTHAL walls, where the cosine of the polar
 * wall is meant -- the cones |rcyl sin(a) - z0 cos(a)|, a the elevation of the wall (tan_theta_lim = tan a, 1e30 at the
 * pole); the azimuthal walls of a 3D grid like in the cylindrical routine. */
static double distance_to_closest_wall_sph(const oracle_model *m, int icell, double x, double y, double z) {
  const int ri0 = m->cell_map_i[icell - 1];
  int tj0 = m->cell_map_j[icell - 1];
  if (tj0 < 0) tj0 = -tj0;
  const double r2_cyl = x * x + y * y, rcyl = sqrt(r2_cyl), r = sqrt(r2_cyl + z * z), z0 = fabs(z);
  double s = m->r_lim[ri0] - r;
  const double s2 = r - m->r_lim[ri0 - 1];
  if (s2 < s) s = s2;
  for (int j = tj0 - 1; j <= tj0; ++j) {
    const double t = m->tan_theta_lim[j];
    const double c = 1.0 / sqrt(1.0 + t * t), sn = t * c;
    const double d = fabs(rcyl * sn - z0 * c);
    if (d < s) s = d;
  }
  if (m->l3D && m->n_az > 1 && m->sin_phi_lim) {
    const int k0 = m->cell_map_k[icell - 1], km = k0 > 1 ? k0 - 1 : m->n_az;
    const double s5 = fabs(x * m->sin_phi_lim[k0 - 1] - y * m->cos_phi_lim[k0 - 1]);
    const double s6 = fabs(x * m->sin_phi_lim[km - 1] - y * m->cos_phi_lim[km - 1]);
    if (s5 < s) s = s5;
    if (s6 < s) s = s6;
  }
  return s;
}

/* distance_to_closest_wall_Voronoi (Voronoi.f90:996-1061) in its working form: n . (p - r) / |n| to the closest face (the
 * reference divides by n . n: the distance in units of the neighbour separation); 0 in a cut cell and next to the box */
static double distance_to_closest_wall_voro(const oracle_model *m, int icell, double x, double y, double z) {
  if (m->v_was_cut && m->v_was_cut[icell - 1]) return 0.0;
  const float *c = m->v_xyz + 3 * (size_t)(icell - 1);
  double s = 1.0e30;
  for (int i = m->v_first[icell - 1]; i <= m->v_last[icell - 1]; ++i) {
    const int id = m->v_neigh[i - 1];
    if (id <= 0) return 0.0;
    const float *nbx = m->v_xyz + 3 * (size_t)(id - 1);
    const double n0 = (double)nbx[0] - (double)c[0], n1 = (double)nbx[1] - (double)c[1], n2 = (double)nbx[2] - (double)c[2];
    const double p0 = 0.5 * ((double)nbx[0] + (double)c[0]), p1 = 0.5 * ((double)nbx[1] + (double)c[1]), p2 = 0.5 * ((double)nbx[2] + (double)c[2]);
    double d = (n0 * (p0 - x) + n1 * (p1 - y) + n2 * (p2 - z)) / sqrt(n0 * n0 + n1 * n1 + n2 * n2);
    if (d < 0.0) d = 0.0;
    if (d < s) s = d;
  }
  return s;
}

double oracle_distance_to_closest_wall_cyl(const oracle_model *m, int icell, double x, double y, double z) {
  if (m->grid_type == 3) return distance_to_closest_wall_voro(m, icell, x, y, z);
  if (m->grid_type == 2) return distance_to_closest_wall_sph(m, icell, x, y, z);
  const int ri0 = m->cell_map_i[icell - 1];
  int zj0 = m->cell_map_j[icell - 1];
  const double r = sqrt(x * x + y * y);
  const double s1 = m->r_lim[ri0] - r;
  const double s2 = r - m->r_lim[ri0 - 1];
  const double z0 = fabs(z);
  if (zj0 < 0) zj0 = -zj0;
  const double s3 = zlim(m, ri0, zj0 + 1) - z0;
  const double s4 = z0 - zlim(m, ri0, zj0);
  double s = s1 < s2 ? s1 : s2;
  if (s3 < s) s = s3;
  if (s4 < s) s = s4;
  if (m->l3D && m->n_az > 1 && m->sin_phi_lim) { /* phi walls (:1198-1218): |x sin(phi) - y cos(phi)| */
    const int k0 = m->cell_map_k[icell - 1];
    const int km = k0 > 1 ? k0 - 1 : m->n_az; /* (the reference reads sin_phi_lim(0), out of bounds, for k0 = 1) */
    const double s5 = fabs(x * m->sin_phi_lim[k0 - 1] - y * m->cos_phi_lim[k0 - 1]);
    const double s6 = fabs(x * m->sin_phi_lim[km - 1] - y * m->cos_phi_lim[km - 1]);
    if (s5 < s) s = s5;
    if (s6 < s) s = s6;
  }
  return s;
}

void oracle_mrw_zeta_table(int n, double *zeta) {
  for (int i = 1; i <= n; ++i) {
    const double yv = (double)(i - 1) / (double)(n - 1);
    double zt = 0.0;
    if (i == n) {
      zt = 0.5;
    } else {
      for (int j = 1;; ++j) {
        const double term = pow(yv, (double)(j * j));
        if (term == 0.0) break;
        if (j % 2 == 0) zt -= term; else zt += term;
      }
    }
    zeta[i - 1] = zt * 2.0;
  }
}

double oracle_mrw_sample_y(const oracle_model *m, float xi) {
  const double *zt = m->mrw_zeta;
  const int n = m->mrw_n_zeta;
  const double x = xi > 0.0f ? (double)xi : 2.9802322387695312e-08; /* a zero draw: half the smallest one */
  int lo = 0, hi = n - 1; /* zt[lo] <= x < zt[hi] */
  while (hi - lo > 1) {
    const int mid = (lo + hi) / 2;
    if (zt[mid] <= x) lo = mid; else hi = mid;
  }
  const double f = (x - zt[lo]) / (zt[hi] - zt[lo]);
  return ((double)lo + f) / (double)(n - 1);
}

/* One walk: returns 0 when the criterion d * chi > gamma does not hold at the start (nothing is drawn). */
static int mrw_walk(worker_t *W, int *lambda, int icell, double *x, double *y, double *z, double *u, double *v,
                    double *w, double Stokes[4]) {
  const oracle_model *m = W->m;
  double d = oracle_distance_to_closest_wall_cyl(m, icell, *x, *y, *z);
  int Ti; float Temp; double frac;
  cell_temperature(W, icell, &Ti, &Temp, &frac);
  const double kf = m->kappa_factor[icell - 1];
  /* lvariable_dust: the mean opacities of the cell's class ([p_n_cells][n_T]) */
  const size_t co = m->p_n_cells ? (size_t)(m->p_icell[icell - 1] - 1) * m->n_T : 0;
  const double *t_chi = m->mrw_chi + co, *t_kdep = m->mrw_kappa_dep + co, *t_ext = m->mrw_ext + co;
  const double chi = (t_chi[Ti - 2] * (1.0 - frac) + t_chi[Ti - 1] * frac) * kf;
  if (!(d * chi > (double)m->mrw_gamma)) return 0;
  const double kdep = t_kdep[Ti - 2] * (1.0 - frac) + t_kdep[Ti - 1] * frac;
  /* the radius the diffusion solution is extrapolated to (the packets' mean free paths are not small against a
   * sphere of a few of them): d + ext, ext at the reference cell's density */
  const double ext = (t_ext[Ti - 2] * (1.0 - frac) + t_ext[Ti - 1] * frac) / kf;
  const double cst_ct = 3.0 / (PI * PI);
  float r4[4];
  double su, sv, sw;
  uint32_t blk = 0;
  do {
    rng_mrw_block(&W->rng, blk++, r4);
    /* the packet leaves the sphere of radius d through a uniformly distributed point (MRW.f90:85-89) */
    sw = 2.0 * (double)r4[0] - 1.0;
    const double uv = sqrt(1.0 - sw * sw), phi = PI * (2.0 * (double)r4[1] - 1.0);
    su = uv * cos(phi);
    sv = uv * sin(phi);
    *x += su * d;
    *y += sv * d;
    *z += sw * d;
    /* the path it travelled inside: Min et al. eq. 8 (MRW.f90:93-99) */
    const double yv = oracle_mrw_sample_y(m, r4[2]);
    const double de = d + ext;
    const double ct = -log(yv) * cst_ct * chi * (de * de);
    /* "Deposit energy using Planck mean opacity" (MRW.f90:102-103): the Lucy estimator over that path */
    W->E_abs[icell - 1] += kdep * ct * Stokes[0];
    W->cnt[ORC_CNT_MRW_STEPS]++;
    d = oracle_distance_to_closest_wall_cyl(m, icell, *x, *y, *z);
  } while (d * chi > (double)m->mrw_gamma);
  /* "Only at end of MRW, to switch to MC: select new wavelength, select new photon direction" (MRW.f90:108-112):
   * the packet was absorbed and re-emitted inside, so it leaves as a thermal packet of this cell -- through the
   * sphere's surface, i.e. outwards: cosine law about the last step's direction */
  rng_mrw_block(&W->rng, blk, r4);
  if (m->mrw_exit_cdf) {
    /* the packet crosses the sphere in the middle of a flight: its wavelength is that of the packets in flight (the
     * thick cell's radiation field), not of a packet that has just been emitted -- the latter prefers the wavelengths of
     * high opacity and would be re-absorbed next to the sphere, where the former flies on */
    const size_t eo = (m->p_n_cells ? (size_t)(m->p_icell[icell - 1] - 1) * m->n_T : 0) * (size_t)m->n_lambda;
    int Te; float Tf; double fe;
    cell_temperature(W, icell, &Te, &Tf, &fe);   /* (the walk's deposits included, as im_reemission_LTE has it) */
    const double *c1 = m->mrw_exit_cdf + eo + (size_t)m->n_lambda * (Te - 2), *c2 = m->mrw_exit_cdf + eo + (size_t)m->n_lambda * (Te - 1);
    int l1 = 0, l2 = m->n_lambda, l = (l1 + l2) / 2;
    while ((l2 - l1) > 1) {
      const double proba = (1.0 - fe) * c1[l - 1] + fe * c2[l - 1];
      if ((double)r4[0] > proba) l1 = l; else l2 = l;
      l = (l1 + l2) / 2;
    }
    *lambda = l + 1;
  } else im_reemission_LTE(W, icell, r4[3], r4[0], lambda);
  oracle_cdapres(sqrt((double)r4[1]), PI * (2.0 * (double)r4[2] - 1.0), su, sv, sw, u, v, w);
  Stokes[1] = 0.0; Stokes[2] = 0.0; Stokes[3] = 0.0;
  W->cnt[ORC_CNT_MRW_WALKS]++;
  return 1;
}

static void propagate_packet(worker_t *W, int *lambda, int p_lambda,
                             int *icell, double *x, double *y, double *z,
                             double *u, double *v, double *w, double Stokes[4],
                             int *flag_star, int *flag_ISM, int *flag_scatt,
                             int *lpacket_alive) {
  const oracle_model *m = W->m;
  int flag_sortie = 0;
  int n_interactions_in_cell = 0;                             /* :1204 */
  *flag_scatt = 0;
  for (;;) {
    float rand = rng_tau(&W->rng);                            /* :1208 */
    double tau;
    if (W->o->tau_fp32) {
      float tf;
      if (rand == 1.0f) tf = 1.0e30f;
      else if (rand > 1.0e-6f) tf = -logf(1.0f - rand);
      else tf = rand;
      tau = (double)tf;
    } else {
      if (rand > 1.0e-6f) tau = -log(1.0 - (double)rand);
      else tau = (double)rand;
    }
    /* :1222-1239.  A walk starts from a packet the cell has just re-emitted (it is what the walk's diffusion
     * describes: a packet that still carries starlight it has only scattered has its first absorption ahead) */
    if (m->mrw && !W->mono && n_interactions_in_cell > m->mrw_n_inter && !*flag_scatt && !*flag_star) {
      if (mrw_walk(W, lambda, *icell, x, y, z, u, v, w, Stokes)) {
        *flag_star = 0; *flag_scatt = 0; *flag_ISM = 0;
      }
    }
    const uint64_t cross_before = W->cnt[ORC_CNT_CROSSINGS], dark_before = W->cnt[ORC_CNT_DARK];
    W->flag_direct_star = *flag_star && !*flag_scatt;
    physical_length(W, *lambda, Stokes, icell, x, y, z, u, v, w, *flag_star, tau,
                    &flag_sortie, lpacket_alive);             /* :1243 */
    /* :1244-1249 counts the flights that end in the cell they started in; here: that never left it (in a 2D grid
     * a flight can come back to its ring on the other side of the star) */
    if (W->cnt[ORC_CNT_CROSSINGS] - cross_before == 1 && W->cnt[ORC_CNT_DARK] == dark_before) {
      if (n_interactions_in_cell < 7) n_interactions_in_cell++;
    } else n_interactions_in_cell = 0;
    if (flag_sortie) return;                                  /* :1251 */

    rng_begin_event(&W->rng);
    rand = rng_float(&W->rng);                                /* :1280 */
    if (W->mono) { /* forced scattering (:1263-1278); the draw above is not used */
      if (m->l_dark_zone && m->l_dark_zone[*icell - 1]) { W->cnt[ORC_CNT_ABS]++; *lpacket_alive = 0; return; }
      const double alb = (double)tab_albedo(m, *icell, *lambda);
      Stokes[0] *= alb; Stokes[1] *= alb; Stokes[2] *= alb; Stokes[3] *= alb;
      if (Stokes[0] < (double)(FLT_MIN * 1.0e6f)) { /* tiny_real_x1e6 */
        W->cnt[ORC_CNT_ABS]++; /* SED mode: "absorptions" counts the packets dropped here */
        *lpacket_alive = 0;
        return;
      }
      rand = -1.0f;
    }
    if (rand < tab_albedo(m, *icell, *lambda)) {              /* :1284 */
      *flag_scatt = 1;
      W->cnt[ORC_CNT_SCATT]++;
      int igrain = 0;
      if (m->scattering_method1) {                            /* :1288-1290: the grain that scatters */
        rand = rng_float(&W->rng);
        igrain = oracle_select_scattering_grain(m, *lambda, *icell, rand);
      }
      rand = rng_float(&W->rng);                              /* :1319 */
      float rand2 = rng_float(&W->rng);
      int itheta; double cospsi, u1, v1, w1;
      /* lvariable_dust with per-class scattering tables: (0:nang, p_n_cells, n_lambda) slices of the cell's class */
      const int vsc = m->p_n_cells && m->v_prob_s11_pos;
      const size_t vrow = vsc ? (size_t)(m->p_icell[*icell - 1] - 1) : 0;
      if (m->scattering_method1 && m->aniso_method == 1) {    /* :1294-1305: the grain's own Mie phase function */
        angle_diff_theta_grain(m, *lambda, igrain, rand, rand2, &itheta, &cospsi);
        rand = rng_float(&W->rng);
        double phi = PI * (2.0 * (double)rand - 1.0);
        oracle_cdapres(cospsi, phi, *u, *v, *w, &u1, &v1, &w1);
        if (m->lsepar_pola) {
          double M[16];
          get_mueller_matrix_per_grain(m, *lambda, itheta, rand2, igrain, M);
          oracle_update_stokes(Stokes, *u, *v, *w, u1, v1, w1, M);
        }
      } else if (m->aniso_method == 1) {
        int pl = (W->mono || m->p_lambda_fixed) ? p_lambda : *lambda;
        if (vsc) {
          oracle_model mv = *m; /* the class's column (p_icell, pl) as a one-column table */
          mv.prob_s11_pos = m->v_prob_s11_pos + ((size_t)(pl - 1) * m->p_n_cells + vrow) * (m->nang_scatt + 1);
          oracle_angle_diff_theta_pos(&mv, 1, rand, rand2, &itheta, &cospsi);
        } else
        oracle_angle_diff_theta_pos(m, pl, rand, rand2, &itheta, &cospsi);
        if (m->lisotropic) { itheta = 1; cospsi = 2.0 * (double)rand - 1.0; }
        rand = rng_float(&W->rng);
        double phi = PI * (2.0 * (double)rand - 1.0);
        oracle_cdapres(cospsi, phi, *u, *v, *w, &u1, &v1, &w1);
        if (m->lsepar_pola) {
          double M[16];
          if (vsc) {
            oracle_model mv = *m;
            const size_t o = ((size_t)(*lambda - 1) * m->p_n_cells + vrow) * (m->nang_scatt + 1);
            mv.s12_o_s11 = m->v_s12_o_s11 + o; mv.s22_o_s11 = m->v_s22_o_s11 + o; mv.s33_o_s11 = m->v_s33_o_s11 + o;
            mv.s34_o_s11 = m->v_s34_o_s11 + o; mv.s44_o_s11 = m->v_s44_o_s11 + o;
            get_mueller_matrix_per_cell(&mv, 1, itheta, rand2, M);
          } else
          get_mueller_matrix_per_cell(m, *lambda, itheta, rand2, M);
          oracle_update_stokes(Stokes, *u, *v, *w, u1, v1, w1, M);
        }
      } else {
        oracle_hg(m->scattering_method1 ? m->m1_tab_g[(size_t)(igrain - 1) + (size_t)m->m1_n_grains * (*lambda - 1)] /* :1307 */
                  : vsc ? m->v_tab_g_pos[vrow + (size_t)m->p_n_cells * (*lambda - 1)] : m->tab_g_pos[*lambda - 1], rand, m->nang_scatt, &itheta,
                  &cospsi);
        if (m->lisotropic) { itheta = 1; cospsi = 2.0 * (double)rand - 1.0; }
        rand = rng_float(&W->rng);
        double phi = PI * (2.0 * (double)rand - 1.0);
        oracle_cdapres(cospsi, phi, *u, *v, *w, &u1, &v1, &w1);
      }
      *u = u1; *v = v1; *w = w1;                              /* :1351 */
    } else {
      W->cnt[ORC_CNT_ABS]++;
      *flag_star = 0; *flag_scatt = 0; *flag_ISM = 0;         /* :1367 */
      rand = rng_float(&W->rng);                              /* :1374 */
      float rand2 = rng_float(&W->rng);
      im_reemission_LTE(W, *icell, rand, rand2, lambda);
      random_isotropic_direction(&W->rng, u, v, w);           /* :1398 */
      Stokes[1] = 0.0; Stokes[2] = 0.0; Stokes[3] = 0.0;
    }
  }
}

/* capteur, SED branch (output.f90:294-397, 572-592) */
static int capteur(worker_t *W, int lambda, double uin, double vin,
                   double win, const double stokin[4], int flag_star,
                   int flag_scatt) {
  const oracle_model *m = W->m;
  double u1 = uin, v1 = vin, w1 = win;
  double stok[4] = {stokin[0], stokin[1], stokin[2], stokin[3]};
  if (w1 < 0.0) {
    if (m->l_sym_centrale) {
      u1 = -u1; v1 = -v1; w1 = -w1;
      stok[2] = -stok[2];
    } else {
      return 0; /* capt is left undefined by the reference here (output.f90:339) */
    }
  }
  int capt = (int)((-1.0 * w1 + 1.0) * (double)m->N_thet) + 1;
  if (capt == m->N_thet + 1) capt = m->N_thet;
  int c_phi;
  if (m->l_sym_axiale) {
    if (v1 < 0.0) { v1 = -v1; stok[2] = -stok[2]; }
    if (w1 == 1.0) c_phi = 1;
    else c_phi = (int)(atan2(v1, u1) / PI * (double)m->N_phi) + 1;
  } else {
    if (w1 == 1.0) c_phi = 1;
    else
      c_phi = (int)(modulo_d(atan2(u1, v1) + PI / 2, 2 * PI) / (2 * PI) *
                    (double)m->N_phi) + 1;
  }
  if (c_phi == m->N_phi + 1) c_phi = m->N_phi;
  else if (c_phi == 0) c_phi = 1;

  size_t plane = (size_t)m->n_lambda * m->N_thet * m->N_phi;
  size_t idx = (size_t)(lambda - 1) +
               (size_t)m->n_lambda * ((capt - 1) + (size_t)m->N_thet * (c_phi - 1));
  W->sed[0 * plane + idx] += stok[0];
  W->sed[1 * plane + idx] += stok[1];
  W->sed[2 * plane + idx] += stok[2];
  W->sed[3 * plane + idx] += stok[3];
  W->sed[4 * plane + idx] += 1.0;
  if (flag_star) {
    if (flag_scatt) W->sed[6 * plane + idx] += stok[0];
    else W->sed[5 * plane + idx] += stok[0];
  } else {
    if (flag_scatt) W->sed[8 * plane + idx] += stok[0];
    else W->sed[7 * plane + idx] += stok[0];
  }
  W->cnt[ORC_CNT_ESCAPED]++;
  return capt;
}

/* one packet of mc_photon_loop's body (dust_transfer.f90:529-552) */
static int one_packet(worker_t *W, uint64_t packet) {
  const oracle_model *m = W->m;
  rng_init(&W->rng, W->o->seed, packet);
  W->rng.two_blocks = W->m->scattering_method1 != 0;
  rng_begin_event(&W->rng);
  W->cnt[ORC_CNT_PACKETS]++;
  int lambda, icell = 0, lintersect, flag_star, flag_ISM, flag_scatt = 0;
  int alive = 1;
  double x, y, z, u, v, w, Stokes[4];
  float rand = rng_float(&W->rng);
  oracle_select_wl_em(m, rand, &lambda);                      /* :536-537 */
  W->n_sent[lambda - 1] += 1.0;                               /* :531 */
  int rc = emit_packet(W, lambda, &icell, &x, &y, &z, &u, &v, &w, Stokes,
                       &flag_star, &flag_ISM, &lintersect);
  if (rc) return rc;
  if (lintersect)
    propagate_packet(W, &lambda, 1, &icell, &x, &y, &z, &u, &v, &w, Stokes,
                     &flag_star, &flag_ISM, &flag_scatt, &alive);
  if (alive && !flag_ISM)
    capteur(W, lambda, u, v, w, Stokes, flag_star, flag_scatt);
  return 0;
}

int oracle_run_thermal(const oracle_model *m, const oracle_opts *o,
                       const double *E_prior, double *E_abs, double *sed,
                       double *n_sent, uint64_t *counters) {
  int nth = o->n_threads > 0 ? o->n_threads : 1;
  if (o->frozen && !E_prior) return 21;
  const size_t nc = (size_t)m->n_cells;
  const size_t nsed =
      (size_t)ORACLE_N_SED_TYPES * m->n_lambda * m->N_thet * m->N_phi;
  double *E_t = (double *)calloc(nc * nth, sizeof(double));
  int *xT_t = (int *)malloc(nc * nth * sizeof(int));
  double *sed_t = (double *)calloc(nsed * nth, sizeof(double));
  double *ns_t = (double *)calloc((size_t)m->n_lambda * nth, sizeof(double));
  worker_t *Ws = NULL;
  if (posix_memalign((void **)&Ws, 256, (size_t)nth * sizeof(worker_t))) Ws = NULL;
  if (Ws) memset(Ws, 0, (size_t)nth * sizeof(worker_t));
  if (!E_t || !xT_t || !sed_t || !ns_t || !Ws) return 22;
  for (size_t q = 0; q < nc * nth; ++q) xT_t[q] = 2; /* thermal_emission.f90:119 */
  int err = 0;
  const uint64_t chunk = 1024;
  const uint64_t nchunks = (o->n_packets + chunk - 1) / chunk;
#ifdef _OPENMP
#pragma omp parallel num_threads(nth)
#endif
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    worker_t *W = &Ws[tid];
    W->m = m; W->o = o; W->E_prior = E_prior;
    W->E_abs = E_t + nc * tid;
    W->xT_ech = xT_t + nc * tid;
    W->sed = sed_t + nsed * tid;
    W->n_sent = ns_t + (size_t)m->n_lambda * tid;
    W->qscale = (double)nth * (o->n_replicas >= 1.0 ? o->n_replicas : 1.0);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (uint64_t ch = 0; ch < nchunks; ++ch) {
      uint64_t p0 = ch * chunk;
      uint64_t p1 = p0 + chunk < o->n_packets ? p0 + chunk : o->n_packets;
      for (uint64_t p = p0; p < p1; ++p) {
        int rc = one_packet(W, o->first_packet + p);
        if (rc) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
          err = rc;
        }
      }
    }
  }
  memset(E_abs, 0, nc * sizeof(double));
  memset(sed, 0, nsed * sizeof(double));
  memset(n_sent, 0, (size_t)m->n_lambda * sizeof(double));
  memset(counters, 0, ORACLE_N_COUNTERS * sizeof(uint64_t));
  for (int t = 0; t < nth; ++t) {
    for (size_t q = 0; q < nc; ++q) E_abs[q] += E_t[nc * t + q];
    for (size_t q = 0; q < nsed; ++q) sed[q] += sed_t[nsed * t + q];
    for (int q = 0; q < m->n_lambda; ++q)
      n_sent[q] += ns_t[(size_t)m->n_lambda * t + q];
    for (int q = 0; q < ORACLE_N_COUNTERS; ++q) counters[q] += Ws[t].cnt[q];
  }
  free(E_t); free(xT_t); free(sed_t); free(ns_t); free(Ws);
  return err;
}

/* ------------------------------------------------------------------------ */
/* SED mode: one wavelength of run_sed_mc (dust_transfer.f90:828-1042)        */
/* ------------------------------------------------------------------------ */
/* one packet of mc_photon_loop's body with lmono (dust_transfer.f90:529-552);
 * returns 1 when the packet was binned in capt_sup (:551) */
static int one_packet_mono(worker_t *W, uint64_t packet, int *err) {
  rng_init(&W->rng, W->o->seed, packet);
  W->rng.two_blocks = W->m->scattering_method1 != 0;
  rng_begin_event(&W->rng);
  W->cnt[ORC_CNT_PACKETS]++;
  int lambda = W->mono->lambda, icell = 0, lintersect, flag_star, flag_ISM, flag_scatt = 0;
  int alive = 1;
  double x, y, z, u, v, w, Stokes[4];
  (void)rng_float(&W->rng); /* the wavelength draw of the thermal step keeps its slot (lmono: :535) */
  W->n_sent[lambda - 1] += 1.0;
  int rc = emit_packet(W, lambda, &icell, &x, &y, &z, &u, &v, &w, Stokes, &flag_star, &flag_ISM, &lintersect);
  if (rc) { *err = rc; return 0; }
  if (lintersect)
    propagate_packet(W, &lambda, W->mono->p_lambda, &icell, &x, &y, &z, &u, &v, &w, Stokes, &flag_star,
                     &flag_ISM, &flag_scatt, &alive);
  if (alive && !flag_ISM) {
    const int capt = capteur(W, lambda, u, v, w, Stokes, flag_star, flag_scatt);
    return capt == W->mono->capt_sup;
  }
  return 0;
}

int oracle_run_mono(const oracle_model *m, const oracle_mono_opts *o, double *xI_scatt,
                    double *sed, double *n_sent, uint64_t *n_sent_chunk,
                    uint64_t *counters) {
  int nth = o->n_threads > 0 ? o->n_threads : 1;
  if (m->p_n_cells && (!m->v_prob_s11_pos || (o->rt1 == 1 && !m->v_tab_s11_pos))) return 32; /* variable dust: the classes' scattering tables */
  if (o->lambda < 1 || o->lambda > m->n_lambda || o->p_lambda < 1 || o->n_chunks < 1) return 23;
  if (o->rt1 == 1 && (m->RT_n_incl * m->RT_n_az > ORACLE_MAX_RT || m->RT_n_incl < 1 || (!m->p_n_cells && !m->tab_s11_pos))) return 24;
  const size_t nsed = (size_t)ORACLE_N_SED_TYPES * m->n_lambda * m->N_thet * m->N_phi;
  if (o->rt1 == 2 && (m->l3D || m->grid_type == 3 || !o->I_spec || !o->I_spec_star || o->n_theta_I < 1 || o->n_phi_I < 1)) return 25; /* rt2: 2D only */
  const size_t nxI = o->rt1 == 1 ? (size_t)m->n_az_rt * m->n_theta_rt * m->N_type_flux * m->RT_n_incl * m->RT_n_az * (size_t)m->n_cells : 0;
  double *sed_t = (double *)calloc(nsed * nth, sizeof(double));
  double *ns_t = (double *)calloc((size_t)m->n_lambda * nth, sizeof(double));
  worker_t *Ws = NULL;
  if (posix_memalign((void **)&Ws, 256, (size_t)nth * sizeof(worker_t))) Ws = NULL;
  if (!sed_t || !ns_t || !Ws) return 22;
  memset(Ws, 0, (size_t)nth * sizeof(worker_t));
  if (nxI) memset(xI_scatt, 0, nxI * sizeof(double));
  oracle_opts base;
  memset(&base, 0, sizeof(base));
  base.seed = o->seed; base.n_threads = nth;
  int err = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nth)
#endif
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    worker_t *W = &Ws[tid];
    W->m = m; W->o = &base; W->mono = o; W->xI = xI_scatt;
    W->sed = sed_t + nsed * tid;
    W->n_sent = ns_t + (size_t)m->n_lambda * tid;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int ch = 0; ch < o->n_chunks; ++ch) { /* nnfot1 (:525) */
      double n_phot_sed2 = 0.0, n_in_loop = 0.0;
      uint64_t seq = 0;
      while (n_phot_sed2 < o->n_photons2 && n_in_loop < o->n_phot_lim) { /* :530 */
        n_in_loop += 1.0;
        int e = 0;
        if (one_packet_mono(W, ((uint64_t)(ch + o->first_chunk) << 40) | seq, &e)) n_phot_sed2 += 1.0;
        if (e) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
          err = e;
          break;
        }
        ++seq;
      }
      n_sent_chunk[ch] = seq;
    }
  }
  memset(sed, 0, nsed * sizeof(double));
  memset(n_sent, 0, (size_t)m->n_lambda * sizeof(double));
  memset(counters, 0, ORACLE_N_COUNTERS * sizeof(uint64_t));
  for (int t = 0; t < nth; ++t) {
    for (size_t q = 0; q < nsed; ++q) sed[q] += sed_t[nsed * t + q];
    for (int q = 0; q < m->n_lambda; ++q) n_sent[q] += ns_t[(size_t)m->n_lambda * t + q];
    for (int q = 0; q < ORACLE_N_COUNTERS; ++q) counters[q] += Ws[t].cnt[q];
  }
  free(sed_t); free(ns_t); free(Ws);
  return err;
}

/* ------------------------------------------------------------------------ */
/* define_dark_zone (optical_depth.f90:1425-1651), 2D cylindrical grids       */
/* ------------------------------------------------------------------------ */
int oracle_define_dark_zone(const oracle_model *m, int lambda, double tau_max_in, const double *r_lim,
                            const double *r_grid, const double *z_grid, unsigned char *dz) {
  if (m->l3D || m->grid_type == 3) return 31;
  const int n_rad = m->n_rad, nz = m->nz;
  const float tau_max = (float)tau_max_in; /* real, intent(in) */
  int ri_in = n_rad, ri_out = 1;  /* (kappa(p_icell, lambda): the cell's class with lvariable_dust, optical_depth.f90:1454-1458) */
  int *zj_sup = (int *)calloc((size_t)n_rad + 1, sizeof(int));
  if (!zj_sup) return 22;
  memset(dz, 0, (size_t)m->n_cells);
  float total; /* real :: total_sum */
  /* step 1: radially from the centre (:1460-1470); cell_map(i,1,1) = i in 2D */
  total = 0.0f;
  for (int i = 1; i <= n_rad; ++i) {
    total = (float)((double)total + tab_kappa(m, i, lambda) * m->kappa_factor[i - 1] * (r_lim[i] - r_lim[i - 1]));
    if (total > tau_max) { ri_in = i; break; }
  }
  /* step 2: radially from the outer edge (:1473-1482) */
  total = 0.0f;
  for (int i = n_rad; i >= 1; --i) {
    total = (float)((double)total + tab_kappa(m, i, lambda) * m->kappa_factor[i - 1] * (r_lim[i] - r_lim[i - 1]));
    if (total > tau_max) { ri_out = i; break; }
  }
  if (ri_out == n_rad) ri_out = n_rad - 1;
  /* step 3: vertically from the top (:1485-1497) */
  for (int i = ri_in; i <= ri_out; ++i) {
    total = 0.0f;
    for (int j = nz; j >= 1; --j) {
      const int icell = i + n_rad * (j - 1);
      const double dzl = m->z_lim[(i - 1) + (size_t)n_rad * j] - m->z_lim[(i - 1) + (size_t)n_rad * (j - 1)];
      total = (float)((double)total + tab_kappa(m, icell, lambda) * m->kappa_factor[icell - 1] * dzl);
      if (total > tau_max) { zj_sup[i] = j; break; }
    }
  }
  /* step 4: test rays in 11 directions from the cell centres (:1522-1551) */
  worker_t W;
  memset(&W, 0, sizeof(W));
  oracle_opts o;
  memset(&o, 0, sizeof(o));
  double *E_dummy = (double *)calloc((size_t)m->n_cells, sizeof(double));
  if (!E_dummy) { free(zj_sup); return 22; }
  oracle_model mm = *m;
  /* l_dark_zone(:) = .false. (:1517), then set column by column while step 4 runs: physical_length reads the flags of the
   * columns already decided (its mirror, :104-112, returns flag_sortie = .false.), so a ray that enters a cell flagged
   * earlier "does not leave".  MCGPU_ORACLE_DARK_NO_MIRROR=1: the flags ignored while probing (a test's comparison). */
  mm.l_dark_zone = getenv("MCGPU_ORACLE_DARK_NO_MIRROR") ? NULL : dz;
  W.m = &mm; W.o = &o; W.E_abs = E_dummy;
  const double Stokes[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = (ri_in > 2 ? ri_in : 2); i <= ri_out; ++i) {
    int done = 0;
    for (int j = zj_sup[i]; j >= 1 && !done; --j) {
      int icell = i + n_rad * (j - 1);
      for (int n = 1; n <= 11; ++n) {
        const float angle = (float)(PI * (double)((float)n / (float)12)); /* pi * real(n)/real(nbre_angle+1) */
        double x0 = r_grid[icell - 1], y0 = 0.0, z0 = z_grid[icell - 1];
        double u0 = (double)cosf(angle), v0 = 0.0, w0 = (double)sinf(angle);
        int ic = icell, flag_sortie = 0, alive = 1;
        physical_length(&W, lambda, Stokes, &ic, &x0, &y0, &z0, &u0, &v0, &w0, 0, (double)tau_max, &flag_sortie, &alive);
        if (!flag_sortie) { /* the ray does not leave: this cell and those below are dark */
          for (int jj = 1; jj <= j; ++jj) dz[i + n_rad * (jj - 1) - 1] = 1;
          done = 1;
          break;
        }
      }
    }
  }
  free(E_dummy); free(zj_sup);
  return 0;
}

#define ORC_PC_TO_AU (648000.0 / PI)           /* constants.f90:91 */
#define ORC_AU_TO_CM (149597870700.0 * 100.0)  /* constants.f90:62-65 */
#define ORC_HP 6.626070040e-34
#define ORC_C_LIGHT 299792458.0
#define ORC_KB 1.38064852e-23

/* Steps 1-3 of define_dark_zone (optical_depth.f90:1459-1500) and the extension of zj_sup to the radii outside
 * [ri_in, ri_out] (:1579-1586): the extent of the zone the diffusion approximation refills (2D cylindrical).
 * zj_sup[i-1], i = 1..n_rad (0 where the reference leaves it: mem.f90:169). */
int oracle_dark_zone_extent(const oracle_model *m, int lambda, double tau_max_in, const double *r_lim, int *ri_in_out,
                            int *ri_out_out, int *zj_sup) {
  if (m->l3D || m->grid_type != 1) return 31;
  const int n_rad = m->n_rad, nz = m->nz;
  const float tau_max = (float)tau_max_in;
  int ri_in = n_rad, ri_out = 1;  /* (kappa(p_icell, lambda): the cell's class with lvariable_dust, optical_depth.f90:1454-1458) */
  float total = 0.0f;
  for (int i = 0; i < n_rad; ++i) zj_sup[i] = 0;
  for (int i = 1; i <= n_rad; ++i) {
    total = (float)((double)total + tab_kappa(m, i, lambda) * m->kappa_factor[i - 1] * (r_lim[i] - r_lim[i - 1]));
    if (total > tau_max) { ri_in = i; break; }
  }
  total = 0.0f;
  for (int i = n_rad; i >= 1; --i) {
    total = (float)((double)total + tab_kappa(m, i, lambda) * m->kappa_factor[i - 1] * (r_lim[i] - r_lim[i - 1]));
    if (total > tau_max) { ri_out = i; break; }
  }
  if (ri_out == n_rad) ri_out = n_rad - 1;
  for (int i = ri_in; i <= ri_out; ++i) {
    total = 0.0f;
    for (int j = nz; j >= 1; --j) {
      const int icell = i + n_rad * (j - 1);
      const double dzl = m->z_lim[(i - 1) + (size_t)n_rad * j] - m->z_lim[(i - 1) + (size_t)n_rad * (j - 1)];
      total = (float)((double)total + tab_kappa(m, icell, lambda) * m->kappa_factor[icell - 1] * dzl);
      if (total > tau_max) { zj_sup[i - 1] = j; break; }
    }
  }
  if (ri_in <= ri_out) {
    for (int i = 1; i < ri_in; ++i) zj_sup[i - 1] = zj_sup[ri_in - 1];
    for (int i = ri_out + 1; i <= n_rad; ++i) zj_sup[i - 1] = zj_sup[ri_out - 1];
  }
  *ri_in_out = ri_in; *ri_out_out = ri_out;
  return 0;
}

/* ------------------------------------------------------------------------ */
/* Temp_approx_diffusion_vertical (diffusion.f90:292-374) with clean_temperature (:183), temperature_to_DensE        */
/* (:131), setDiffusion_coeff0 (:78), iter_Temp_approx_diffusion_vertical (:504), setDiffusion_coeff (:17) and        */
/* DensE_to_temperature (:162): the 1+1D diffusion fill of the dark zone, 2D cylindrical grids.  PARITY UNPINNED      */
/* (module diffusion uses dust_prop / thermal_emission, unbuildable here); known-answer tests in tests/.               */
/* ------------------------------------------------------------------------ */
#define ORC_DELTA_CELL_DARK_ZONE 3 /* cylindrical_grid.f90:39 */

/* the Rosseland-type sum of setDiffusion_coeff[0] for one cell at temperature Temp */
static double diffusion_coeff(const oracle_model *m, const double *tab_lambda, const double *tab_delta_lambda,
                              int icell, double Temp) {
  const float thermal_const = (float)(ORC_C_LIGHT * ORC_HP / ORC_KB);            /* constants.f90:24, default real */
  const double cst_Dcoeff = PI / (double)(12.0f * 5.670367e-8f);                  /* pi/(12.*sigma), sigma default real */
  const double cst = (double)thermal_const / Temp;
  double total_sum = 0.0;
  for (int l = 0; l < m->n_lambda; ++l) {
    const double wl = tab_lambda[l] * (double)1.e-6f;
    const double delta_wl = tab_delta_lambda[l] * (double)1.e-6f;
    const double cst_wl = cst / wl;
    double dB_dT;
    if (cst_wl < 200.0) {
      const double coeff_exp = exp(cst_wl);
      const double wl2 = wl * wl, wl5 = (wl2 * wl2) * wl; /* wl**5 */
      dB_dT = cst_wl * coeff_exp / (wl5 * ((coeff_exp - 1.0) * (coeff_exp - 1.0)));
    } else {
      dB_dT = 0.0;
    }
    total_sum = total_sum + dB_dT / (tab_kappa(m, icell, l + 1) * m->kappa_factor[icell - 1]) * delta_wl; /* kappa(p_icell, lambda), diffusion.f90:60 */
  }
  return cst_Dcoeff * total_sum / (Temp * Temp * Temp);
}

int oracle_temp_approx_diffusion_vertical(const oracle_model *m, const double *tab_lambda,
                                          const double *tab_delta_lambda, int ri_in, int ri_out, const int *zj_sup,
                                          float *Tdust, int *n_iter_out) {
  if (m->l3D || m->grid_type != 1) return 31;
  const int n_rad = m->n_rad, nz = m->nz, dcz = ORC_DELTA_CELL_DARK_ZONE;
  /* clean_temperature (:183-199) */
  for (int i = ri_in; i <= ri_out; ++i)
    for (int j = 1; j <= zj_sup[i - 1]; ++j) Tdust[i + n_rad * (j - 1) - 1] = m->T_min;
  double *DensE = (double *)calloc((size_t)nz + 2, sizeof(double));
  double *DensE_m1 = (double *)calloc((size_t)nz + 2, sizeof(double));
  double *Dcoeff = (double *)calloc((size_t)nz + 2, sizeof(double));
  if (!DensE || !DensE_m1 || !Dcoeff) { free(DensE); free(DensE_m1); free(Dcoeff); return 22; }
  const int i_lo = (ri_in - dcz > 3) ? ri_in - dcz : 3, i_hi = (ri_out + dcz < n_rad - 2) ? ri_out + dcz : n_rad - 2;
  int n_iter_total = 0;
  for (int i = i_lo; i <= i_hi; ++i) {
    const float precision = 1.0e-6f;
    const float stabilite = 2.0f;
    int jtop = zj_sup[i - 1] + dcz;
    if (jtop > nz - 1) jtop = nz - 1; /* the stencil reads j+1 */
    /* temperature_to_DensE (:131-158): Tdust**4 in default real */
    for (int j = 1; j <= nz; ++j) {
      const float T = Tdust[i + n_rad * (j - 1) - 1];
      DensE[j] = (double)((T * T) * (T * T));
    }
    DensE[0] = DensE[1];
    memcpy(DensE_m1, DensE, ((size_t)nz + 1) * sizeof(double));
    /* setDiffusion_coeff0 (:78-127) */
    for (int j = 1; j <= nz; ++j)
      Dcoeff[j] = diffusion_coeff(m, tab_lambda, tab_delta_lambda, i + n_rad * (j - 1), (double)Tdust[i + n_rad * (j - 1) - 1]);
    int n_iter = 0;
    for (;;) {
      n_iter++;
      /* iter_Temp_approx_diffusion_vertical (:504-594) */
      double dt_min = 1.79769313486231570815e+308;
      for (int j = 1; j <= jtop; ++j) {
        const double dz = m->z_lim[(i - 1) + (size_t)n_rad]; /* cell_height(i,j) = zmax(i)/nz = z_lim(i,2), cylindrical_grid.f90:459-463 */
        const double t = dz * dz / Dcoeff[j];
        if (t < dt_min) dt_min = t;
      }
      const double dt = (double)(stabilite * 0.5f) * dt_min;
      memcpy(DensE_m1, DensE, ((size_t)nz + 1) * sizeof(double));
      float max_delta_E_r = 0.0f;
      for (int j = 1; j <= jtop; ++j) {
        const double dz = m->z_lim[(i - 1) + (size_t)n_rad]; /* cell_height(i,j) = zmax(i)/nz = z_lim(i,2), cylindrical_grid.f90:459-463 */
        const double dE_dz_p1 = DensE_m1[j + 1] - DensE_m1[j];
        const double dE_dz_m1 = DensE_m1[j] - DensE_m1[j - 1];
        const double d2E_dz2 = Dcoeff[j] * (dE_dz_p1 - dE_dz_m1) / (2.0 * (dz * dz));
        const double delta_E = d2E_dz2 * dt;
        DensE[j] = DensE_m1[j] + delta_E;
        const double delta_E_r = delta_E / DensE[j];
        if (delta_E_r > (double)max_delta_E_r) max_delta_E_r = (float)delta_E_r;
      }
      DensE[0] = DensE[1];
      if (max_delta_E_r < precision) break;
      if (n_iter > 50000000) { free(DensE); free(DensE_m1); free(Dcoeff); return 23; }
      /* setDiffusion_coeff (:17-74): refreshed where the energy density moved by more than 10 % in this step */
      for (int j = 1; j <= nz; ++j) {
        if (fabs(DensE[j] - DensE_m1[j]) > 1.0e-1 * DensE_m1[j])
          Dcoeff[j] = diffusion_coeff(m, tab_lambda, tab_delta_lambda, i + n_rad * (j - 1), pow(DensE[j], (double)0.25f));
      }
      Dcoeff[0] = Dcoeff[1];
    }
    n_iter_total += n_iter;
    /* DensE_to_temperature (:162-179) */
    for (int j = 1; j <= jtop; ++j) Tdust[i + n_rad * (j - 1) - 1] = (float)pow(DensE[j], (double)0.25f);
  }
  if (n_iter_out) *n_iter_out = n_iter_total;
  free(DensE); free(DensE_m1); free(Dcoeff);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* Ray-traced SED of the dust, RT method 1 (dust_transfer.f90:1413-1600)      */
/* ------------------------------------------------------------------------ */

/* rotation_3d (utils.f90:1545-1589) */
static void rotation_3d(const double axis[3], double angle_deg, const double v[3], double out[3]) {
  const double d = v[0] * axis[0] + v[1] * axis[1] + v[2] * axis[2];
  double vp[3] = {d * axis[0], d * axis[1], d * axis[2]};
  double vn[3] = {v[0] - vp[0], v[1] - vp[1], v[2] - vp[2]};
  const double norm = sqrt(vn[0] * vn[0] + vn[1] * vn[1] + vn[2] * vn[2]);
  if (norm < DBL_MIN) { out[0] = vp[0]; out[1] = vp[1]; out[2] = vp[2]; return; }
  vn[0] /= norm; vn[1] /= norm; vn[2] /= norm;
  const double vn2[3] = {axis[1] * vn[2] - axis[2] * vn[1], axis[2] * vn[0] - axis[0] * vn[2],
                         axis[0] * vn[1] - axis[1] * vn[0]};
  const double ca = cos(angle_deg * (PI / 180.0)), sa = sin(angle_deg * (PI / 180.0));
  for (int q = 0; q < 3; ++q) out[q] = vp[q] + norm * (ca * vn[q] + sa * vn2[q]);
}

/* calc_Jth (dust_ray_tracing.f90:810-846), LTE grains */
static double *rt_calc_Jth(const oracle_model *m, int lam, double wl, const float *Tdust) {
  double *J_th = (double *)calloc((size_t)m->n_cells, sizeof(double));
  if (!J_th) return NULL;
  const double cst_E = 2.0 * ORC_HP * ORC_C_LIGHT * ORC_C_LIGHT;
  const float thermal_const = (float)(ORC_C_LIGHT * ORC_HP / ORC_KB);
  for (int ic = 0; ic < m->n_cells; ++ic) {
    const double Temp = (double)Tdust[ic];
    if (Temp * wl > 3.e-4) {
      const double cst_wl = (double)thermal_const / (Temp * wl);
      const double coeff_exp = exp(cst_wl);
      J_th[ic] = cst_E / (pow(wl, 5) * (coeff_exp - 1.0)) * wl * tab_kappa_abs(m, ic + 1, lam) * m->kappa_factor[ic];
    }
  }
  return J_th;
}

/* ---- ray tracing method 2: dust_source_fct (dust_ray_tracing.f90:1478-1700) on the eps_dust2 / eps_dust2_star of one
 * inclination (oracle_init_dust_source_fct2).  While a source is set, the dust maps below ray-trace that inclination
 * (observer q = ibin - 1; method 2 is 2D and knows no observer azimuth) with it instead of method 1's eps_dust1. */
typedef struct { const float *eps2, *eps2_star; int nang_rt, nang_star, q; const double *z_grid; } rt2_source_t;
static rt2_source_t g_rt2_store;
static const rt2_source_t *g_rt2 = NULL;
void oracle_set_rt2_source(const float *eps_dust2, const float *eps_dust2_star, int nang_rt, int nang_star, int ibin,
                           const double *z_grid) {
  if (!eps_dust2) { g_rt2 = NULL; return; }
  g_rt2_store.eps2 = eps_dust2; g_rt2_store.eps2_star = eps_dust2_star; g_rt2_store.nang_rt = nang_rt;
  g_rt2_store.nang_star = nang_star; g_rt2_store.q = ibin - 1; g_rt2_store.z_grid = z_grid;
  g_rt2 = &g_rt2_store;
}

/* interpolate_Stokes_QU (:1705-1742): between two (P I, 2 theta) pairs, back to (Q, U) */
static void interpolate_stokes_qu(const float a[2], const float b[2], double frac1, float out[2]) {
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

/* one corner of dust_source_fct's interpolation: the cell icell_tmp, between the directions iscatt1 / iscatt2 */
static void rt2_corner(const oracle_model *m, const rt2_source_t *R, int icell_tmp, int dir, double phi_pos, double SF[8]) {
  const int n_Stokes = m->lsepar_pola ? 4 : 1, ntf = m->N_type_flux;
  for (int t = 0; t < 8; ++t) SF[t] = 0.0;
  {
    const int N = R->nang_rt;
    const double xiscatt = fmax(phi_pos / (2 * PI) * (double)N, 0.0);
    int iscatt1 = (int)floor(xiscatt);
    const double frac = xiscatt - iscatt1, un_m_frac = 1.0 - frac;
    int iscatt2 = iscatt1 + 1;
    iscatt1 = ((iscatt1 % N) + N) % N; if (iscatt1 == 0) iscatt1 = N;
    iscatt2 = ((iscatt2 % N) + N) % N; if (iscatt2 == 0) iscatt2 = N;
    const float *e1 = R->eps2 + (size_t)ntf * ((size_t)(iscatt1 - 1) + (size_t)N * (dir + 2 * (size_t)(icell_tmp - 1)));
    const float *e2 = R->eps2 + (size_t)ntf * ((size_t)(iscatt2 - 1) + (size_t)N * (dir + 2 * (size_t)(icell_tmp - 1)));
    SF[0] = (double)e2[0] * frac + (double)e1[0] * un_m_frac;
    if (m->lsepar_pola) {
      float qu[2];
      interpolate_stokes_qu(e1 + 1, e2 + 1, un_m_frac, qu);
      SF[1] = (double)qu[0]; SF[2] = (double)qu[1];
    }
    if (m->lsepar_contrib)
      for (int t = n_Stokes; t < ntf; ++t) SF[t] = (double)e2[t] * frac + (double)e1[t] * un_m_frac;
  }
  {
    const int N = R->nang_star;
    const double xiscatt = fmax(phi_pos / (2 * PI) * (double)N, 0.0);
    int iscatt1 = (int)floor(xiscatt);
    const double frac = xiscatt - iscatt1, un_m_frac = 1.0 - frac;
    int iscatt2 = iscatt1 + 1;
    iscatt1 = ((iscatt1 % N) + N) % N; if (iscatt1 == 0) iscatt1 = N;
    iscatt2 = ((iscatt2 % N) + N) % N; if (iscatt2 == 0) iscatt2 = N;
    const float *e1 = R->eps2_star + (size_t)n_Stokes * ((size_t)(iscatt1 - 1) + (size_t)N * (dir + 2 * (size_t)(icell_tmp - 1)));
    const float *e2 = R->eps2_star + (size_t)n_Stokes * ((size_t)(iscatt2 - 1) + (size_t)N * (dir + 2 * (size_t)(icell_tmp - 1)));
    SF[0] = (SF[0] + (double)e2[0] * frac) + (double)e1[0] * un_m_frac;
    if (m->lsepar_pola) {
      float qu[2];
      interpolate_stokes_qu(e1 + 1, e2 + 1, un_m_frac, qu);
      SF[1] = SF[1] + (double)qu[0]; SF[2] = SF[2] + (double)qu[1];
    }
    if (m->lsepar_contrib) SF[n_Stokes + 1] = (SF[n_Stokes + 1] + (double)e2[0] * frac) + (double)e1[0] * un_m_frac;
  }
}

/* dust_source_fct, method 2 (:1478-1700): linear in z between the cell and its vertical neighbour on the side of the
 * point (the radial interpolation is switched off in the reference: ri1 = ri, frac_r = 1), linear in the azimuth between
 * the tabulated directions */
static void dust_source_fct2(const oracle_model *m, const rt2_source_t *R, int icell, double x, double y, double z, double SF[8]) {
  const int n_rad = m->n_rad, nz = m->nz, ntf = m->N_type_flux;
  const int ri = m->cell_map_i[icell - 1], zj = m->cell_map_j[icell - 1];
  int zj1, zj2;
  double frac_z;
  if (fabs(z) > R->z_grid[icell - 1]) { zj1 = zj; zj2 = zj + 1; } else { zj1 = zj - 1; zj2 = zj; }
  if (zj2 > nz) { zj2 = nz; frac_z = 1.0; }
  else if (zj1 < 1) { zj1 = 1; frac_z = 1.0; }
  else {
    const double za = R->z_grid[(ri - 1) + n_rad * (zj2 - 1)], zb = R->z_grid[(ri - 1) + n_rad * (zj1 - 1)];
    frac_z = (za - fabs(z)) / (za - zb);
  }
  frac_z = fmax(fmin(1.0, frac_z), 0.0);
  const double phi_pos = modulo_d(atan2(x, y) + 2 * PI, 2 * PI);
  const int dir = z > 0.0 ? 1 : 0;
  double SF1[8], SF3[8];
  rt2_corner(m, R, (ri - 1) + n_rad * (zj1 - 1) + 1, dir, phi_pos, SF1);
  rt2_corner(m, R, (ri - 1) + n_rad * (zj2 - 1) + 1, dir, phi_pos, SF3);
  const double frac_r = 1.0;
  for (int t = 0; t < ntf; ++t) SF[t] = frac_r * frac_z * SF1[t] + frac_r * (1.0 - frac_z) * SF3[t];
}

/* integ_ray_dust (optical_depth.f90:1327-1421) from (x0,y0,z0) on the grid edge in cell icell, direction
 * (u0,v0,w0), with dust_source_fct of RT method 1 (dust_ray_tracing.f90:1455-1475) = eps_dust1(k,psup,:,icell)
 * / kappa_ext, eps_dust1 built here from xI_scatt like init_dust_source_fct1 (:676-703) for observer q. */
static void rt1_integ_ray_dust(const oracle_model *m, int lam, double tau_dark_zone_obs, const double *xI,
                               const double *J_th, double photon_energy, int q, double x0, double y0, double z0,
                               double u0, double v0, double w0, int icell, double S[8]) {
  const int ntf = m->N_type_flux, nRT = m->RT_n_incl * m->RT_n_az, nc = m->n_cells;
  const int n_Stokes = m->lsepar_pola ? 4 : 1; /* init_mcfost.f90:1603-1616 */
  const size_t st_type = (size_t)m->n_az_rt * m->n_theta_rt, st_rt = st_type * ntf;
  for (int t = 0; t < 8; ++t) S[t] = 0.0;
  double x1 = x0, y1 = y0, z1 = z0, tau = 0.0;
  int next_cell = icell, lis, i_star, icell_star;
  oracle_intersect_stars(m, x0, y0, z0, u0, v0, w0, &lis, &i_star, &icell_star);
  for (long guard = 0; guard < 100000000L; ++guard) {
    const int ic = next_cell;
    const double xa = x1, ya = y1, za = z1;
    if (grid_test_exit(m, ic, xa, ya, za)) break;
    if (lis && ic == icell_star) break;
    double l, lc, lv;
    grid_cross_cell(m, xa, ya, za, u0, v0, w0, ic, 0, &x1, &y1, &z1, &next_cell, &l, &lc, &lv);
    if (ic <= nc) {
      const double kappa_ext = tab_kappa(m, ic, lam) * m->kappa_factor[ic - 1];
      const double dtau = lc * kappa_ext;
      const double xm = 0.5 * (xa + x1), ym = 0.5 * (ya + y1), zm = 0.5 * (za + z1);
      int k = 1, psup = 1;
      if (!m->l3D) {
        psup = (zm > 0.0) ? 1 : 2;
        const double phi_pos = atan2(xm, ym);
        k = (int)floor(modulo_d(phi_pos, 2 * PI) / (2 * PI) * (double)m->n_az_rt) + 1;
        if (k > m->n_az_rt) k = m->n_az_rt;
      }
      if (g_rt2) { /* method 2: the interpolated source function at the middle of the path (optical_depth.f90:1396-1404) */
        double SF[8];
        dust_source_fct2(m, g_rt2, ic, xm, ym, zm, SF);
        const double wgt = exp(-tau) * (1.0 - exp(-dtau));
        for (int t = 0; t < ntf; ++t) S[t] += wgt * SF[t];
      } else if (kappa_ext > DBL_MIN) {
        const double factor = photon_energy / m->volume[ic - 1] * m->n_az_rt * m->n_theta_rt;
        const double kappa_sca = kappa_ext * (double)tab_albedo(m, ic, lam);
        const double *px = xI + (size_t)(k - 1) + (size_t)m->n_az_rt * (psup - 1) + st_rt * ((size_t)q + (size_t)nRT * (ic - 1));
        const double wgt = exp(-tau) * (1.0 - exp(-dtau));
        double eps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = 0; t < ntf; ++t) eps[t] = px[st_type * t] * factor * kappa_sca; /* I_scatt(:,:,itype) */
        double src[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        src[0] = (eps[0] + J_th[ic - 1]) / kappa_ext;
        if (m->lsepar_pola) { src[1] = eps[1] / kappa_ext; src[2] = eps[2] / kappa_ext; src[3] = eps[3] / kappa_ext; }
        if (m->lsepar_contrib) {
          src[n_Stokes + 1] = eps[n_Stokes + 1] / kappa_ext; /* n_Stokes+2 */
          src[n_Stokes + 2] = J_th[ic - 1] / kappa_ext;      /* n_Stokes+3 */
          src[n_Stokes + 3] = eps[n_Stokes + 3] / kappa_ext; /* n_Stokes+4 */
        }
        for (int t = 0; t < ntf; ++t) S[t] += wgt * src[t];
      }
      tau += dtau;
      if (tau > tau_dark_zone_obs) break;
    }
  }
}

/* image-plane basis of dust_map (dust_transfer.f90:1440-1455) */
static void rt_image_plane(const oracle_model *m, const oracle_rt_opts *o, int ibin, int iaz, double uvw[3],
                           double xpi[3], double ypi[3], double center[3]) {
  const int q = (ibin - 1) + m->RT_n_incl * (iaz - 1);
  uvw[0] = m->tab_u_rt[q]; uvw[1] = m->tab_v_rt[q]; uvw[2] = m->tab_w_rt[ibin - 1];
  const double az = (double)o->tab_RT_az[iaz - 1] * (PI / 180.0);
  const double x[3] = {cos(az), sin(az), 0.0};
  if (fabs(o->ang_disque) > (double)FLT_MIN) rotation_3d(uvw, o->ang_disque, x, xpi);
  else { xpi[0] = x[0]; xpi[1] = x[1]; xpi[2] = x[2]; }
  /* y_plan_image = -cross_product(x_plan_image, uvw) */
  ypi[0] = -(xpi[1] * uvw[2] - xpi[2] * uvw[1]);
  ypi[1] = -(xpi[2] * uvw[0] - xpi[0] * uvw[2]);
  ypi[2] = -(xpi[0] * uvw[1] - xpi[1] * uvw[0]);
  const double lfar = 10. * o->Rmax;
  for (int c = 0; c < 3; ++c) center[c] = uvw[c] * lfar;
}

static double rt_photon_energy(const oracle_rt_opts *o) { /* (:661-663), SED / image branch alike */
  return o->E_src * o->wl_um * 1.0e-6 / (o->n_sent_photons * ORC_AU_TO_CM * PI);
}

/* ------------------------------------------------------------------------ */
/* compute_stars_map for the SED (dust_transfer.f90:1604-1854, lresolved = .false., no limb darkening)               */
/* ------------------------------------------------------------------------ */
/* optical_length_tot (optical_depth.f90:248-324), cylindrical grids */
static float optical_length_tot_from(const oracle_model *m, int lambda, int icell, double x, double y, double z, double u, double v,
                                     double w);
static float optical_length_tot(const oracle_model *m, int lambda, double x, double y, double z, double u, double v,
                                double w) {
  int icell;
  grid_index_cell(m, x, y, z, &icell);
  return optical_length_tot_from(m, lambda, icell, x, y, z, u, v, w);
}
/* ... from a known cell (the callers at dust_transfer.f90:2196 and :1680 hand it in) */
static float optical_length_tot_from(const oracle_model *m, int lambda, int icell, double x, double y, double z, double u, double v,
                                     double w) {
  int next_cell, previous_cell = 0;
  next_cell = icell;
  double x1 = x, y1 = y, z1 = z, tau_tot = 0.0;
  int icell0 = 0;
  for (;;) {
    previous_cell = icell0;
    icell0 = next_cell;
    const double x0 = x1, y0 = y1, z0 = z1;
    if (grid_test_exit(m, icell0, x0, y0, z0)) return (float)tau_tot;
    const double opacity = (icell0 <= m->n_cells && icell0 >= 1) ? tab_kappa(m, icell0, lambda) * m->kappa_factor[icell0 - 1] : 0.0;
    double l, l_contrib, l_void;
    grid_cross_cell(m, x0, y0, z0, u, v, w, icell0, previous_cell, &x1, &y1, &z1, &next_cell, &l, &l_contrib, &l_void);
    tau_tot += l_contrib * opacity;
  }
}

int oracle_stars_map_sed(const oracle_model *m, const oracle_rt_opts *o, uint64_t seed, const double *star_flux,
                         double *out) {
  enum { NXS = 10, NS = 21, N_RAY_SED = 1024 };
  const int nRT = m->RT_n_incl * m->RT_n_az;
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (int q = 0; q < nRT; ++q) out[q] = 0.0;
  for (int q = 0; q < nRT; ++q)
    for (int istar = 0; istar < m->n_stars; ++istar) {
      double uvw[3], xpi[3], ypi[3], center[3];
      rt_image_plane(m, o, q % m->RT_n_incl + 1, q / m->RT_n_incl + 1, uvw, xpi, ypi, center);
      const oracle_star *st = &m->stars[istar];
      const double delta = st->r / (double)NXS;
      const double nx = sqrt(xpi[0] * xpi[0] + xpi[1] * xpi[1] + xpi[2] * xpi[2]);
      const double ny = sqrt(ypi[0] * ypi[0] + ypi[1] * ypi[1] + ypi[2] * ypi[2]);
      const double dxs[3] = {delta * xpi[0] / nx, delta * xpi[1] / nx, delta * xpi[2] / nx};
      const double dys[3] = {delta * ypi[0] / ny, delta * ypi[1] / ny, delta * ypi[2] / ny};
      float tau_screen[NS * NS];
      for (int j = -NXS; j <= NXS; ++j)
        for (int i = -NXS; i <= NXS; ++i)
          tau_screen[(i + NXS) + NS * (j + NXS)] =
              optical_length_tot(m, o->lambda, st->x + dxs[0] * i + dys[0] * j, st->y + dxs[1] * i + dys[1] * j,
                                 st->z + dxs[2] * i + dys[2] * j, uvw[0], uvw[1], uvw[2]);
      const int n_ray = N_RAY_SED / m->n_stars > 1 ? N_RAY_SED / m->n_stars : 1;
      const double norm_screen2 = 1.0 / (delta * delta);
      double sum_f = 0.0, sum_n = 0.0;
      for (int iray = 0; iray < n_ray; ++iray) {
        uint32_t ctr[4] = {(uint32_t)iray, 2u, (uint32_t)(q * m->n_stars + istar), 0u}, r4[4];
        oracle_philox4x32_10(ctr, key, r4);
        const float rand = u32_to_real(r4[0]), rand2 = u32_to_real(r4[1]);
        const double z = 2.0 * (double)rand - 1.0;
        const double srw02 = sqrt(1.0 - z * z), argmt = PI * (2.0 * (double)rand2 - 1.0);
        const double x = srw02 * cos(argmt), y = srw02 * sin(argmt);
        const float cos_thet = (float)fabs(x * uvw[0] + y * uvw[1] + z * uvw[2]);
        const double vec[3] = {x * st->r, y * st->r, z * st->r};
        const double offset_x = (vec[0] * dxs[0] + vec[1] * dxs[1] + vec[2] * dxs[2]) * norm_screen2;
        const double offset_y = (vec[0] * dys[0] + vec[1] * dys[1] + vec[2] * dys[2]) * norm_screen2;
        const int i = (int)floor(offset_x), j = (int)floor(offset_y);
        const double fx = offset_x - i, fy = offset_y - j;
        float tau = 0.0f;
        if (i >= -NXS && i < NXS && j >= -NXS && j < NXS) {
          const int p = (i + NXS) + NS * (j + NXS);
          tau = (float)((double)tau_screen[p] * (1 - fx) * (1 - fy) + (double)tau_screen[p + 1] * fx * (1 - fy) +
                        (double)tau_screen[p + NS] * (1 - fx) * fy + (double)tau_screen[p + NS + 1] * fx * fy);
        }
        sum_f += (double)(expf(-tau) * cos_thet);
        sum_n += (double)cos_thet;
      }
      out[q] += star_flux[istar] * sum_f / sum_n;
    }
  return 0;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* Ray tracing method 2: init_dust_source_fct2 (dust_ray_tracing.f90:717-806) with calc_Isca_rt2 (:907-1240),           */
/* calc_Isca_rt2_star (:1245-1440), angles_scatt_rt2 (:304-405) and calc_Jth -- the source function of one inclination  */
/* from the specific intensity the packet loop stored (I_spec, I_spec_star): 2D grids.                                  */
/* ------------------------------------------------------------------------------------------------------------------ */
#define RT2_N_SUPER 5

/* nint(acos(cos_scatt) * real(nang_scatt) / pi) with cos_scatt a default real (the correctly rounded default-real acos) */
static int rt2_angle_index(float cos_scatt, int nang) {
  const float ac = (float)acos((double)cos_scatt);
  if (ac != ac) return nang;
  return (int)llrint(floor((double)(ac * (float)nang) / PI + 0.5));
}

/* the rotation angle's cosine and sine between the scattering plane and the meridian of the ray-tracing direction
 * (:1030-1065, :369-398): omega = 2 acos(v1pj / |(v1pj, v1pk)|), negative with v1pk */
static double rt2_omega(double u, double v, double w, double ur, double vr, double wr) {
  double v1pi, v1pj, v1pk;
  oracle_rotation(u, v, w, -ur, -vr, -wr, &v1pi, &v1pj, &v1pk);
  double xnyp = sqrt(v1pk * v1pk + v1pj * v1pj), costhet;
  if (xnyp < 1e-10) costhet = 1.0; else costhet = v1pj / xnyp;
  double theta = acos(costhet);
  if (theta >= PI) theta = 0.0;
  double omega = 2.0 * theta;
  if (v1pk < 0.0) omega = -1.0 * omega;
  return omega;
}

int oracle_init_dust_source_fct2(const oracle_model *m, const oracle_rt_opts *o, int p_lambda, int ibin, int n_theta_I,
                                 int n_phi_I, int nang_rt, int nang_star, const double *I_spec, const double *I_spec_star,
                                 const float *Tdust, const double *r_grid, const double *z_grid, float *eps_dust2,
                                 float *eps_dust2_star) {
  if (m->l3D || m->grid_type == 3) return 31;
  const int lam = o->lambda, nang = m->nang_scatt, na1 = nang + 1, nc = m->n_cells;
  const int n_Stokes = m->lsepar_pola ? 4 : 1, ntf = n_Stokes + (m->lsepar_contrib ? 4 : 0);
  const int vd = m->p_n_cells != 0;
  if (vd ? !m->v_tab_s11_pos : !m->tab_s11_pos) return 24;
  const double w0 = m->tab_w_rt[ibin - 1], uv0 = sqrt(1.0 - w0 * w0); /* tab_uv_rt(ibin) = sin(incl), :280-289 */
  const double photon_energy = rt_photon_energy(o);
  double *J_th = rt_calc_Jth(m, lam, o->wl_um * 1.e-6, Tdust);
  const size_t nbin = (size_t)n_theta_I * n_phi_I, nsup = RT2_N_SUPER * RT2_N_SUPER;
  const size_t ntab = nbin * nang_rt * 2;
  int *tab_k = (int *)malloc(sizeof(int) * ntab * nsup);
  float *tab_sin = (float *)malloc(sizeof(float) * ntab * nsup);
  double *tab_cosw = (double *)calloc(ntab, sizeof(double)), *tab_sinw = (double *)calloc(ntab, sizeof(double));
  int *tab_kc = (int *)malloc(sizeof(int) * ntab);
  if (!J_th || !tab_k || !tab_sin || !tab_cosw || !tab_sinw || !tab_kc) return 22;
  /* ---- the directions where Inu * s11 is evaluated (:973-1072); index ((dir * nang_rt + iscatt) * n_phi_I + phi_I) * n_theta_I + theta_I */
  for (int dir = 0; dir <= 1; ++dir)
    for (int iscatt = 1; iscatt <= nang_rt; ++iscatt) {
      const float phi_scatt = (float)(2 * PI * (double)((float)iscatt / (float)nang_rt)); /* two_pi * real / real -> default real */
      const double ur = uv0 * sin((double)phi_scatt), vr = -uv0 * cos((double)phi_scatt), wr = w0;
      for (int phi_I = 1; phi_I <= n_phi_I; ++phi_I)
        for (int theta_I = 1; theta_I <= n_theta_I; ++theta_I) {
          const size_t b = (((size_t)dir * nang_rt + (iscatt - 1)) * n_phi_I + (phi_I - 1)) * n_theta_I + (theta_I - 1);
          float sum_sin = 0.f;
          for (int i2 = 1; i2 <= RT2_N_SUPER; ++i2)
            for (int i1 = 1; i1 <= RT2_N_SUPER; ++i1) {
              const float f1 = (float)i1 / (float)(RT2_N_SUPER + 1), f2 = (float)i2 / (float)(RT2_N_SUPER + 1);
              const double w = (2.0 * (((double)theta_I - (double)f1) / (double)n_theta_I) - 1.0) * (double)(2 * dir - 1);
              const double phi = 2 * PI * ((double)phi_I - (double)f2) / (double)n_phi_I;
              const double w02 = sqrt(1.0 - w * w), u = w02 * sin(phi), v = -w02 * cos(phi);
              const float cos_scatt = (float)(ur * u + vr * v + wr * w);
              int k = rt2_angle_index(cos_scatt, nang);
              if (k > nang) k = nang;
              if (k < 0) k = 0;
              const float sin_scatt = (float)sqrt(1.0 - (double)cos_scatt * (double)cos_scatt);
              tab_k[b * nsup + (i1 - 1) + RT2_N_SUPER * (i2 - 1)] = k;
              tab_sin[b * nsup + (i1 - 1) + RT2_N_SUPER * (i2 - 1)] = sin_scatt;
              sum_sin = sum_sin + sin_scatt;
            }
          for (size_t t = 0; t < nsup; ++t) tab_sin[b * nsup + t] = tab_sin[b * nsup + t] / sum_sin;
          if (m->lsepar_pola) { /* the bin's centre for the polarisation (:1021-1070) */
            const double w = (2.0 * (((double)theta_I - (double)0.5f) / (double)n_theta_I) - 1.0) * (double)(2 * dir - 1);
            const double phi = 2 * PI * ((double)phi_I - (double)0.5f) / (double)n_phi_I;
            const double w02 = sqrt(1.0 - w * w), u = w02 * sin(phi), v = -w02 * cos(phi);
            const double omega = rt2_omega(u, v, w, ur, vr, wr);
            double cosw = cos(omega), sinw = sin(omega);
            if (fabs(cosw) < 1e-06) cosw = 0.0;
            if (fabs(sinw) < 1e-06) sinw = 0.0;
            tab_cosw[b] = cosw; tab_sinw[b] = sinw;
          }
          tab_kc[b] = tab_k[b * nsup + (RT2_N_SUPER / 2) + RT2_N_SUPER * (RT2_N_SUPER / 2)]; /* i1 = i2 = N_super/2 + 1 */
        }
    }
  /* ---- per cell: the scattered field towards the nang_rt directions and both hemispheres (:1130-1232) */
  memset(eps_dust2, 0, sizeof(float) * (size_t)ntf * nang_rt * 2 * nc);
  memset(eps_dust2_star, 0, sizeof(float) * (size_t)n_Stokes * nang_star * 2 * nc);
  for (int icell = 1; icell <= nc; ++icell) {
    const size_t cls = vd ? (size_t)(m->p_icell[icell - 1] - 1) : 0;
    const size_t col = vd ? (size_t)na1 * ((size_t)(p_lambda - 1) * m->p_n_cells + cls) : (size_t)na1 * (p_lambda - 1);
    const float *t11 = (vd ? m->v_tab_s11_pos : m->tab_s11_pos) + col;
    const float *t12 = (vd ? m->v_s12_o_s11 : m->s12_o_s11), *t22 = (vd ? m->v_s22_o_s11 : m->s22_o_s11);
    const float *t33 = (vd ? m->v_s33_o_s11 : m->s33_o_s11), *t34 = (vd ? m->v_s34_o_s11 : m->s34_o_s11);
    const float *t44 = (vd ? m->v_s44_o_s11 : m->s44_o_s11);
    const double kappa_ext = tab_kappa(m, icell, lam) * m->kappa_factor[icell - 1];
    const double factor = photon_energy / m->volume[icell - 1];
    const double kappa_sca = tab_kappa(m, icell, lam) * m->kappa_factor[icell - 1] * (double)tab_albedo(m, icell, lam);
    const double *Inu = I_spec + (size_t)ntf * nbin * (icell - 1); /* (N_type_flux, n_theta_I, n_phi_I, icell) */
    float *eps = eps_dust2 + (size_t)ntf * nang_rt * 2 * (icell - 1); /* (N_type_flux, nang_rt, 0:1, icell) */
    for (int dir = 0; dir <= 1; ++dir)
      for (int iscatt = 1; iscatt <= nang_rt; ++iscatt) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int phi_I = 1; phi_I <= n_phi_I; ++phi_I)
          for (int theta_I = 1; theta_I <= n_theta_I; ++theta_I) {
            const size_t b = (((size_t)dir * nang_rt + (iscatt - 1)) * n_phi_I + (phi_I - 1)) * n_theta_I + (theta_I - 1);
            float s11 = 0.f;
            for (size_t t = 0; t < nsup; ++t) s11 = s11 + t11[tab_k[b * nsup + t]] * tab_sin[b * nsup + t];
            const double *st = Inu + (size_t)ntf * ((size_t)(theta_I - 1) + (size_t)n_theta_I * (phi_I - 1));
            if (m->lsepar_pola) {
              const int k = tab_kc[b];
              const double cosw = tab_cosw[b], sinw = tab_sinw[b];
              const float s12 = -s11 * t12[col + k], s22 = s11 * t22[col + k], s33 = -s11 * t33[col + k];
              const float s34 = -s11 * t34[col + k], s44 = -s11 * t44[col + k];
              /* C = ROP S, ROP(2:3,2:3) = [[cosw, -sinw], [sinw, cosw]] */
              const double C1 = st[0], C4 = st[3];
              const double C2 = cosw * st[1] + (-1.0 * sinw) * st[2];
              const double C3 = sinw * st[1] + cosw * st[2];
              /* D = M C */
              const double D1 = (double)s11 * C1 + (double)s12 * C2;
              const double D2 = (double)s12 * C1 + (double)s22 * C2;
              const double D3 = (double)s33 * C3 + (double)(-s34) * C4;
              const double D4 = (double)s34 * C3 + (double)s44 * C4;
              /* S = RPO D, RPO(2:3,2:3) = [[cosw, sinw], [-sinw, cosw]]; S(3) = -S(3) */
              const double S2 = cosw * D2 + sinw * D3;
              const double S3 = -((-1.0 * sinw) * D2 + cosw * D3);
              acc[0] = (float)((double)acc[0] + D1);
              acc[1] = (float)((double)acc[1] + S2);
              acc[2] = (float)((double)acc[2] + S3);
              acc[3] = (float)((double)acc[3] + D4);
            } else {
              acc[0] = (float)((double)acc[0] + (double)s11 * st[0]);
            }
            if (m->lsepar_contrib) {
              acc[n_Stokes + 1] = (float)((double)acc[n_Stokes + 1] + (double)s11 * st[n_Stokes + 1]);
              acc[n_Stokes + 3] = (float)((double)acc[n_Stokes + 3] + (double)s11 * st[n_Stokes + 3]);
            }
          }
        float *e = eps + (size_t)ntf * ((size_t)(iscatt - 1) + (size_t)nang_rt * dir);
        for (int t = 0; t < ntf; ++t) e[t] = (float)(((double)acc[t] * factor) * kappa_sca); /* I_sca2 (:1228) */
      }
    /* ---- calc_Isca_rt2_star (:1292-1432): unscattered starlight scattered once towards the observer */
    float *es = eps_dust2_star + (size_t)n_Stokes * nang_star * 2 * (icell - 1);
    const double Istar = I_spec_star[icell - 1];
    if (!(Istar < 1.e-30)) {
      const double yy = r_grid[icell - 1], zz = z_grid[icell - 1];
      const double norm = sqrt(yy * yy + zz * zz), u = 0.0, v = yy / norm, w = zz / norm;
      const double phi_pos = modulo_d(atan2(0.0, yy) + 2 * PI, 2 * PI); /* angles_scatt_rt2 (:327) at x = 0 */
      for (int iscatt = 1; iscatt <= nang_star; ++iscatt) {
        double phi = 2 * PI * (double)((float)iscatt / (float)nang_star); /* two_pi * real(iscatt) / real(N) */
        phi = phi - phi_pos;
        const double ur = uv0 * sin(phi), vr = -uv0 * cos(phi), wr = w0;
        const double prod1 = ur * u + vr * v;
        for (int dir = 0; dir <= 1; ++dir) {
          const double w2 = dir == 1 ? w : -w;
          const float cos_scatt = (float)(prod1 + wr * w2); /* cos_thet_ray_tracing_star is a default real */
          int k = rt2_angle_index(cos_scatt, nang);
          if (k > nang) k = nang;
          if (k < 1) k = 1;
          const float s11 = t11[k];
          float *e = es + (size_t)n_Stokes * ((size_t)(iscatt - 1) + (size_t)nang_star * dir);
          if (m->lsepar_pola) {
            const double omega = (double)(float)rt2_omega(u, v, w2, ur, vr, wr); /* omega_ray_tracing_star: default real */
            double cosw = cos(omega), sinw = sin(omega);
            if (fabs(cosw) < 1e-06) cosw = 0.0;
            if (fabs(sinw) < 1e-06) sinw = 0.0;
            const float s12 = -s11 * t12[col + k], s22 = s11 * t22[col + k];
            /* Stokes = (I*, 0, 0, 0): C = Stokes; D = M C; S(2:3) = RPO(2:3,2:3) D(2:3), RPO = [[cosw, sinw], [sinw, -cosw]] */
            const double D1 = (double)s11 * Istar, D2 = (double)s12 * Istar;
            (void)s22;
            e[0] = (float)((D1 * factor) * kappa_sca);
            e[1] = (float)(((cosw * D2) * factor) * kappa_sca);
            e[2] = (float)(((sinw * D2) * factor) * kappa_sca);
            e[3] = 0.f;
          } else {
            e[0] = (float)((((double)s11 * Istar) * factor) * kappa_sca);
          }
        }
      }
    }
    /* ---- init_dust_source_fct2 (:749-799) */
    if (kappa_ext > DBL_MIN) {
      for (int dir = 0; dir <= 1; ++dir) {
        for (int iscatt = 1; iscatt <= nang_rt; ++iscatt) {
          float *e = eps + (size_t)ntf * ((size_t)(iscatt - 1) + (size_t)nang_rt * dir);
          float I2[8];
          for (int t = 0; t < ntf; ++t) I2[t] = e[t];
          e[0] = (float)(((double)I2[0] + J_th[icell - 1]) / kappa_ext);
          if (m->lsepar_pola) {
            for (int t = 1; t < 4; ++t) e[t] = (float)((double)I2[t] / kappa_ext);
            const float Q = e[1], U = e[2];
            e[1] = sqrtf(Q * Q + U * U);
            e[2] = atan2f(U, Q);
          }
          if (m->lsepar_contrib) {
            e[n_Stokes + 1] = (float)((double)I2[n_Stokes + 1] / kappa_ext);
            e[n_Stokes + 2] = (float)(J_th[icell - 1] / kappa_ext);
            e[n_Stokes + 3] = (float)((double)I2[n_Stokes + 3] / kappa_ext);
            e[n_Stokes] = 0.f; /* (n_Stokes + 1: direct starlight, filled by the ray tracer) */
          }
        }
        for (int iscatt = 1; iscatt <= nang_star; ++iscatt) {
          float *e = es + (size_t)n_Stokes * ((size_t)(iscatt - 1) + (size_t)nang_star * dir);
          for (int t = 0; t < n_Stokes; ++t) e[t] = (float)((double)e[t] / kappa_ext);
          if (m->lsepar_pola) {
            const float Q = e[1], U = e[2];
            e[1] = sqrtf(Q * Q + U * U);
            e[2] = atan2f(U, Q);
          }
        }
      }
    } else {
      memset(eps, 0, sizeof(float) * (size_t)ntf * nang_rt * 2);
      memset(es, 0, sizeof(float) * (size_t)n_Stokes * nang_star * 2);
    }
  }
  free(J_th); free(tab_k); free(tab_sin); free(tab_cosw); free(tab_sinw); free(tab_kc);
  return 0;
}

/* interp (utils.f90:130-175, default real): linear interpolation in a table, the end values outside it */
static float interp_sp(const float *y, const float *x, int n, float xp) {
  float xmin = x[0], xmax = x[0];
  for (int i = 1; i < n; ++i) { if (x[i] < xmin) xmin = x[i]; if (x[i] > xmax) xmax = x[i]; }
  const int inc = x[n - 1] > x[0];
  if (xp < xmin) return inc ? y[0] : y[n - 1];
  if (xp > xmax) return inc ? y[n - 1] : y[0];
  int j; /* 1-based index of the upper point */
  if (inc) { for (j = 2; j <= n - 1; ++j) if (x[j - 1] > xp) break; }
  else { for (j = 2; j <= n - 1; ++j) if (x[j - 1] < xp) break; }
  const float frac = (xp - x[j - 2]) / (x[j - 1] - x[j - 2]);
  return y[j - 2] * (1.f - frac) + y[j - 1] * frac;
}

/* compute_stars_map for images (dust_transfer.f90:1604-1854 with lresolved = .true.): the stars' discs in the pixel map
 * of every observer.  Per (observer, star): the 21 x 21 screen of optical depths as for the SED; n_ray_star random points
 * of the stellar sphere -- 1024 / n_stars, or 100 per pixel of the disc when the star is wider than a pixel (:1655-1667) --
 * each placed in its pixel (find_pixel :1858-1893) with weight exp(-tau) cos_thet LimbDarkening(cos_thet), normalised so
 * that the star's map sums to star_flux (when every ray falls inside the map).  Limb darkening (n_mu > 0): interp of
 * limb_darkening(mu); with pola_ld the polarised maps Q = P cos 2 phi, U = P sin 2 phi, phi the pixel's position angle
 * about the map centre (:1817-1823: "only works for a star centered").
 * map (npix_x, npix_y, n_maps, nRT) column-major in double (the reference sums default reals per thread), n_maps = 3 with
 * pola_ld else 1; star_position (n_stars, nRT, 2) in arcsec (:1847-1848).  Ray k of (observer q, star s): Philox block
 * (k, 2, q n_stars + s) of the seed, as in oracle_stars_map_sed. */
int oracle_stars_map_image(const oracle_model *m, const oracle_rt_opts *o, uint64_t seed, const double *star_flux,
                           int npix_x, int npix_y, double map_size, double zoom, int n_mu, const float *mu_ld,
                           const float *ld, const float *pola_ld, double *map, double *star_position) {
  enum { NXS = 10, NS = 21, N_RAY_SED = 1024 };
  const int nRT = m->RT_n_incl * m->RT_n_az, n_maps = (n_mu > 0 && pola_ld) ? 3 : 1;
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  const size_t n_pix = (size_t)npix_x * npix_y;
  memset(map, 0, sizeof(double) * n_pix * n_maps * nRT);
  const double taille_pix = (map_size / zoom) / (double)(npix_x > npix_y ? npix_x : npix_y);
  const float pix_size = (float)(map_size / zoom / (double)(npix_x > npix_y ? npix_x : npix_y));
  const int x_center = npix_x / 2 + 1, y_center = npix_y / 2 + 1;
  const double factor_pix = 1.0 / (taille_pix * o->distance);
  for (int q = 0; q < nRT; ++q)
    for (int istar = 0; istar < m->n_stars; ++istar) {
      double uvw[3], xpi[3], ypi[3], center[3];
      rt_image_plane(m, o, q % m->RT_n_incl + 1, q / m->RT_n_incl + 1, uvw, xpi, ypi, center);
      const oracle_star *st = &m->stars[istar];
      const double dx_map[3] = {xpi[0] * taille_pix, xpi[1] * taille_pix, xpi[2] * taille_pix};
      const double dy_map[3] = {ypi[0] * taille_pix, ypi[1] * taille_pix, ypi[2] * taille_pix};
      const double delta = st->r / (double)NXS;
      const double nx = sqrt(xpi[0] * xpi[0] + xpi[1] * xpi[1] + xpi[2] * xpi[2]);
      const double ny = sqrt(ypi[0] * ypi[0] + ypi[1] * ypi[1] + ypi[2] * ypi[2]);
      const double dxs[3] = {delta * xpi[0] / nx, delta * xpi[1] / nx, delta * xpi[2] / nx};
      const double dys[3] = {delta * ypi[0] / ny, delta * ypi[1] / ny, delta * ypi[2] / ny};
      float tau_screen[NS * NS];
      for (int j = -NXS; j <= NXS; ++j)
        for (int i = -NXS; i <= NXS; ++i)
          tau_screen[(i + NXS) + NS * (j + NXS)] =
              optical_length_tot(m, o->lambda, st->x + dxs[0] * i + dys[0] * j, st->y + dxs[1] * i + dys[1] * j,
                                 st->z + dxs[2] * i + dys[2] * j, uvw[0], uvw[1], uvw[2]);
      int n_ray = N_RAY_SED / m->n_stars > 1 ? N_RAY_SED / m->n_stars : 1;
      if (2.0 * st->r > (double)pix_size) { /* resolved: on average 100 rays per pixel (:1659-1662) */
        const float ratio = (float)(st->r / (double)pix_size);
        const int n_res = 100 * (int)(4.0 * PI * (double)(ratio * ratio));
        n_ray = n_res > N_RAY_SED ? n_res : N_RAY_SED;
      }
      const double norm_screen2 = 1.0 / (delta * delta);
      double norm = 0.0;
      double *mp = map + n_pix * n_maps * (size_t)q;
      double *tmp = (double *)calloc(n_pix * n_maps, sizeof(double));
      if (!tmp) return 22;
      for (int iray = 0; iray < n_ray; ++iray) {
        uint32_t ctr[4] = {(uint32_t)iray, 2u, (uint32_t)(q * m->n_stars + istar), 0u}, r4[4];
        oracle_philox4x32_10(ctr, key, r4);
        const float rand = u32_to_real(r4[0]), rand2 = u32_to_real(r4[1]);
        const double z = 2.0 * (double)rand - 1.0;
        const double srw02 = sqrt(1.0 - z * z), argmt = PI * (2.0 * (double)rand2 - 1.0);
        const double x = srw02 * cos(argmt), y = srw02 * sin(argmt);
        const float cos_thet = (float)fabs(x * uvw[0] + y * uvw[1] + z * uvw[2]);
        float LimbDarkening = 1.0f, Pola_LD = 0.0f;
        if (n_mu > 0) {
          LimbDarkening = interp_sp(ld, mu_ld, n_mu, cos_thet);
          if (pola_ld) Pola_LD = interp_sp(pola_ld, mu_ld, n_mu, cos_thet);
        }
        const double vec[3] = {x * st->r, y * st->r, z * st->r};
        const double px = st->x + vec[0], py = st->y + vec[1], pz = st->z + vec[2];
        const double offset_x = (vec[0] * dxs[0] + vec[1] * dxs[1] + vec[2] * dxs[2]) * norm_screen2;
        const double offset_y = (vec[0] * dys[0] + vec[1] * dys[1] + vec[2] * dys[2]) * norm_screen2;
        const int i = (int)floor(offset_x), j = (int)floor(offset_y);
        const double fx = offset_x - i, fy = offset_y - j;
        float tau = 0.0f;
        if (i >= -NXS && i < NXS && j >= -NXS && j < NXS) {
          const int p = (i + NXS) + NS * (j + NXS);
          tau = (float)((double)tau_screen[p] * (1 - fx) * (1 - fy) + (double)tau_screen[p + 1] * fx * (1 - fy) +
                        (double)tau_screen[p + NS] * (1 - fx) * fy + (double)tau_screen[p + NS + 1] * fx * fy);
        }
        /* find_pixel (:1858-1893) */
        const double factor = 1.0 / (taille_pix * taille_pix);
        const double x_map = (px * dx_map[0] + py * dx_map[1] + pz * dx_map[2]) * factor;
        const double y_map = (px * dy_map[0] + py * dy_map[1] + pz * dy_map[2]) * factor;
        const int ip = (npix_x % 2 == 1) ? (int)llround(x_map) + npix_x / 2 + 1 : (int)llround(x_map + 0.5) + npix_x / 2;
        const int jp = (npix_y % 2 == 1) ? (int)llround(y_map) + npix_y / 2 + 1 : (int)llround(y_map + 0.5) + npix_y / 2;
        if (ip >= 1 && ip <= npix_x && jp >= 1 && jp <= npix_y) {
          const float wgt = expf(-tau) * cos_thet * LimbDarkening;
          const size_t pp = (size_t)(ip - 1) + (size_t)npix_x * (jp - 1);
          tmp[pp] += (double)wgt;
          if (n_maps == 3) {
            const float P = wgt * Pola_LD;
            const float phi = atan2f((float)(jp - y_center) * 1.0f, (float)(ip - x_center) * 1.0f);
            tmp[pp + n_pix] += (double)(P * cosf(2.0f * phi));
            tmp[pp + 2 * n_pix] += (double)(P * sinf(2.0f * phi));
          }
        }
        norm += (double)(cos_thet * LimbDarkening);
      }
      const double factor2 = star_flux[istar] / norm;
      for (size_t t = 0; t < n_pix * n_maps; ++t) mp[t] += tmp[t] * factor2;
      free(tmp);
      if (star_position) {
        const double xyz[3] = {st->x, st->y, st->z};
        star_position[(size_t)istar + (size_t)m->n_stars * q] =
            -(xyz[0] * dx_map[0] + xyz[1] * dx_map[1] + xyz[2] * dx_map[2]) * factor_pix;
        star_position[(size_t)istar + (size_t)m->n_stars * (q + (size_t)nRT)] =
            (xyz[0] * dy_map[0] + xyz[1] * dy_map[1] + xyz[2] * dy_map[2]) * factor_pix;
      }
    }
  return 0;
}

int oracle_dust_map_sed(const oracle_model *m, const oracle_rt_opts *o, const double *xI, const float *Tdust,
                        double *out) {
  if (m->grid_type == 3 && g_rt2) return 31; /* Voronoi: method 1 */
  const int ntf = m->N_type_flux, nRT = m->RT_n_incl * m->RT_n_az;
  const int lam = o->lambda;
  memset(out, 0, sizeof(double) * (size_t)ntf * nRT);
  double *J_th = rt_calc_Jth(m, lam, o->wl_um * 1.e-6, Tdust);
  if (!J_th) return 22;
  const double photon_energy = rt_photon_energy(o);

  /* image-plane sampling of dust_map, method 1 (:1481-1535) */
  enum { n_rad_RT = 128, n_phi_RT = 30 };
  double tab_r[n_rad_RT];
  const double rmin_RT = 0.01 * o->Rmin, rmax_RT = 2.0 * o->Rmax;
  tab_r[0] = rmin_RT;
  const double fact_r = exp((1.0 / ((double)n_rad_RT - 1)) * log(rmax_RT / rmin_RT));
  for (int i = 1; i < n_rad_RT; ++i) tab_r[i] = tab_r[i - 1] * fact_r;
  const double fact_A = sqrt(PI * (fact_r - 1.0 / fact_r) / n_phi_RT);
  const double cst_phi = (o->l_sym_ima ? PI : 2 * PI) / (double)n_phi_RT;

  for (int ibin = 1; ibin <= m->RT_n_incl; ++ibin)
    for (int iaz = 1; iaz <= m->RT_n_az; ++iaz) {
      const int q = (ibin - 1) + m->RT_n_incl * (iaz - 1);
      if (g_rt2 && q != g_rt2->q) continue; /* method 2: the inclination its source function was built for */
      double uvw[3], xpi[3], ypi[3], center[3];
      rt_image_plane(m, o, ibin, iaz, uvw, xpi, ypi, center);
      const double u0 = -uvw[0], v0 = -uvw[1], w0 = -uvw[2]; /* reverse propagation */
      double *acc = out + (size_t)q * ntf;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(o->n_threads > 0 ? o->n_threads : 1)
#endif
      for (int ri = 0; ri < n_rad_RT; ++ri) {
        double loc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const double r = tab_r[ri], taille_pix = fact_A * r;
        for (int ph = 1; ph <= n_phi_RT; ++ph) {
          const double phi = cst_phi * ((double)ph - 0.5);
          /* intensite_pixel_dust (:1899-2004) with one sub-pixel: the pixel centre */
          double x0 = center[0] + r * sin(phi) * xpi[0] + r * cos(phi) * ypi[0];
          double y0 = center[1] + r * sin(phi) * xpi[1] + r * cos(phi) * ypi[1];
          double z0 = center[2] + r * sin(phi) * xpi[2] + r * cos(phi) * ypi[2];
          int icell, lintersect;
          grid_move_to_grid(m, &x0, &y0, &z0, u0, v0, w0, &icell, &lintersect);
          if (!lintersect) continue;
          double S[8];
          rt1_integ_ray_dust(m, lam, o->tau_dark_zone_obs, xI, J_th, photon_energy, q, x0, y0, z0, u0, v0, w0, icell, S);
          const double pix = (taille_pix / (o->distance * ORC_PC_TO_AU));
          for (int t = 0; t < ntf; ++t) loc[t] += S[t] * pix * pix; /* (:1989, :1993) */
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        for (int t = 0; t < ntf; ++t) acc[t] += loc[t];
      }
    }
  free(J_th);
  return 0;
}

/* dust_map, method 2 (images, dust_transfer.f90:1537-1577): square pixels of map_size/zoom / max(npix_x,npix_y) AU,
 * each refined by intensite_pixel_dust (:1899-2004) -- 1, 2x2, 4x4 ... sub-pixels, at least n_iter_min = 2
 * iterations, at most n_iter_max = 6, until Stokes I changes by less than 1 %.
 * image[(((type * RT_n_az + iaz) * RT_n_incl + ibin) * npix_y + j) * npix_x + i] = Stokes_ray_tracing(lambda,i,j,ibin,
 * iaz,type); with l_sym_ima only i <= npix_x/2 + mod(npix_x,2) is computed (the writer mirrors, output.f90:1007). */
int oracle_dust_map_image(const oracle_model *m, const oracle_rt_opts *o, int npix_x, int npix_y, double map_size,
                          double zoom, const double *xI, const float *Tdust, double *image, int *n_rays) {
  if (m->grid_type == 3 && g_rt2) return 31;
  if (npix_x < 1 || npix_y < 1 || !(map_size > 0.0) || !(zoom > 0.0)) return 11;
  const int ntf = m->N_type_flux, nRT = m->RT_n_incl * m->RT_n_az;
  const int lam = o->lambda;
  memset(image, 0, sizeof(double) * (size_t)ntf * nRT * npix_x * npix_y);
  double *J_th = rt_calc_Jth(m, lam, o->wl_um * 1.e-6, Tdust);
  if (!J_th) return 22;
  const double photon_energy = rt_photon_energy(o);
  const int n_iter_min = 2, n_iter_max = 6;
  const double precision = 1.e-2;
  const double taille_pix = (map_size / zoom) / (double)(npix_x > npix_y ? npix_x : npix_y);
  const int npix_x_max = o->l_sym_ima ? npix_x / 2 + npix_x % 2 : npix_x;
  long rays = 0;

  for (int ibin = 1; ibin <= m->RT_n_incl; ++ibin)
    for (int iaz = 1; iaz <= m->RT_n_az; ++iaz) {
      const int q = (ibin - 1) + m->RT_n_incl * (iaz - 1);
      if (g_rt2 && q != g_rt2->q) continue; /* method 2: the inclination its source function was built for */
      double uvw[3], xpi[3], ypi[3], center[3], dx[3], dy[3], Icorner[3];
      rt_image_plane(m, o, ibin, iaz, uvw, xpi, ypi, center);
      const double u0 = -uvw[0], v0 = -uvw[1], w0 = -uvw[2];
      for (int c = 0; c < 3; ++c) { dx[c] = xpi[c] * taille_pix; dy[c] = ypi[c] * taille_pix; }
      for (int c = 0; c < 3; ++c) Icorner[c] = center[c] - (0.5 * npix_x * dx[c] + 0.5 * npix_y * dy[c]);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(o->n_threads > 0 ? o->n_threads : 1) reduction(+ : rays)
#endif
      for (int i = 1; i <= npix_x_max; ++i)
        for (int j = 1; j <= npix_y; ++j) {
          double corner[3];
          for (int c = 0; c < 3; ++c) corner[c] = Icorner[c] + (i - 1) * dx[c] + (j - 1) * dy[c];
          double S[8] = {0, 0, 0, 0, 0, 0, 0, 0}, S_old[8];
          int subpixels = 1, iter = 1;
          for (;;) {
            const double npix2 = (double)subpixels * (double)subpixels;
            for (int t = 0; t < 8; ++t) { S_old[t] = S[t]; S[t] = 0.0; }
            double sdx[3], sdy[3];
            for (int c = 0; c < 3; ++c) { sdx[c] = dx[c] / (double)subpixels; sdy[c] = dy[c] / (double)subpixels; }
            for (int si = 1; si <= subpixels; ++si)
              for (int sj = 1; sj <= subpixels; ++sj) {
                double x0 = corner[0] + (si - 0.5) * sdx[0] + (sj - 0.5) * sdy[0];
                double y0 = corner[1] + (si - 0.5) * sdx[1] + (sj - 0.5) * sdy[1];
                double z0 = corner[2] + (si - 0.5) * sdx[2] + (sj - 0.5) * sdy[2];
                int icell, lintersect;
                grid_move_to_grid(m, &x0, &y0, &z0, u0, v0, w0, &icell, &lintersect);
                ++rays;
                if (!lintersect) continue;
                double R[8];
                rt1_integ_ray_dust(m, lam, o->tau_dark_zone_obs, xI, J_th, photon_energy, q, x0, y0, z0, u0, v0, w0, icell, R);
                for (int t = 0; t < ntf; ++t) S[t] += R[t];
              }
            for (int t = 0; t < ntf; ++t) S[t] = S[t] / npix2;
            if (iter < n_iter_min) subpixels *= 2;
            else if (iter >= n_iter_max) break;
            else if (fabs(S[0] - S_old[0]) > precision * S_old[0]) subpixels *= 2;
            else break;
            ++iter;
          }
          const double pix = taille_pix / (o->distance * ORC_PC_TO_AU);
          for (int t = 0; t < ntf; ++t)
            image[((((size_t)t * m->RT_n_az + (iaz - 1)) * m->RT_n_incl + (ibin - 1)) * npix_y + (j - 1)) * npix_x + (i - 1)] =
                S[t] * (pix * pix);
        }
    }
  if (n_rays) *n_rays = (int)(rays > 2147483647L ? 2147483647L : rays);
  free(J_th);
  return 0;
}

/* ---------------------------------------------------------------------------
 * Optical-depth maps of the ray tracer: compute_tau_map (dust_transfer.f90:2114-2210, option -tau_map) and
 * compute_tau_surface_map (:2006-2110, option -tau_surface).  One ray per pixel CENTRE of the observer's image, sent
 * backwards from 10 Rmax: move_to_grid, then
 *   tau_map(i,j,ibin,iaz)           = optical_length_tot from the entry point to the far edge of the grid (default real)
 *   tau_surface_map(i,j,ibin,iaz,:) = the point where physical_length has used up the optical depth `tau` (default reals);
 *                                     zeros when the ray leaves the grid or ends on a star first (flag_sortie).  A ray that
 *                                     meets a cell of the dark zone stops at the entry point of the cell BEFORE it, which is
 *                                     what physical_length's mirror (:104-112) hands back.
 * physical_length also deposits into the radiation field in the reference (a side effect of reusing the packets' routine
 * after the Monte Carlo); nothing is deposited here.  Either output may be NULL.
 * ------------------------------------------------------------------------- */
int oracle_tau_maps(const oracle_model *m, const oracle_rt_opts *o, int npix_x, int npix_y, double map_size, double zoom,
                    float tau, float *tau_map, float *tau_surface_map) {
  if (npix_x < 1 || npix_y < 1 || !(map_size > 0.0) || !(zoom > 0.0)) return 11;
  const int nRT = m->RT_n_incl * m->RT_n_az, lam = o->lambda;
  const size_t npix = (size_t)npix_x * npix_y;
  const double taille_pix = (map_size / zoom) / (double)(npix_x > npix_y ? npix_x : npix_y);
  worker_t W;
  memset(&W, 0, sizeof(W));
  oracle_opts oo;
  memset(&oo, 0, sizeof(oo));
  double *E_dummy = (double *)calloc((size_t)m->n_cells, sizeof(double));
  if (!E_dummy) return 22;
  W.m = m; W.o = &oo; W.E_abs = E_dummy;
  const double Stokes[4] = {0.0, 0.0, 0.0, 0.0};
  for (int ibin = 1; ibin <= m->RT_n_incl; ++ibin)
    for (int iaz = 1; iaz <= m->RT_n_az; ++iaz) {
      const int q = (ibin - 1) + m->RT_n_incl * (iaz - 1);
      double uvw[3], xpi[3], ypi[3], center[3], dx[3], dy[3], Icorner[3];
      rt_image_plane(m, o, ibin, iaz, uvw, xpi, ypi, center);
      for (int c = 0; c < 3; ++c) { dx[c] = xpi[c] * taille_pix; dy[c] = ypi[c] * taille_pix; }
      for (int c = 0; c < 3; ++c) Icorner[c] = center[c] - (0.5 * npix_x * dx[c] + 0.5 * npix_y * dy[c]);
      for (int i = 1; i <= npix_x; ++i)
        for (int j = 1; j <= npix_y; ++j) {
          const size_t at = (size_t)(i - 1) + (size_t)npix_x * ((size_t)(j - 1) + (size_t)npix_y * q);
          double pc[3];
          for (int c = 0; c < 3; ++c) pc[c] = Icorner[c] + (i - 0.5) * dx[c] + (j - 0.5) * dy[c];
          if (tau_map) {
            double x0 = pc[0], y0 = pc[1], z0 = pc[2];
            int icell, lintersect;
            grid_move_to_grid(m, &x0, &y0, &z0, -uvw[0], -uvw[1], -uvw[2], &icell, &lintersect);
            tau_map[at] = lintersect ? optical_length_tot_from(m, lam, icell, x0, y0, z0, -uvw[0], -uvw[1], -uvw[2]) : 0.0f;
          }
          if (tau_surface_map) {
            double x0 = pc[0], y0 = pc[1], z0 = pc[2], u0 = -uvw[0], v0 = -uvw[1], w0 = -uvw[2];
            int icell, lintersect, flag_sortie = 1, alive = 1;
            grid_move_to_grid(m, &x0, &y0, &z0, u0, v0, w0, &icell, &lintersect);
            if (lintersect) physical_length(&W, lam, Stokes, &icell, &x0, &y0, &z0, &u0, &v0, &w0, 0, (double)tau, &flag_sortie, &alive);
            const int hit = lintersect && !flag_sortie;
            tau_surface_map[at] = hit ? (float)x0 : 0.0f;
            tau_surface_map[at + npix * nRT] = hit ? (float)y0 : 0.0f;
            tau_surface_map[at + 2 * npix * nRT] = hit ? (float)z0 : 0.0f;
          }
        }
    }
  free(E_dummy);
  return 0;
}

/* ---------------------------------------------------------------------------
 * init_reemission (thermal_emission.f90:404-550): the LTE tables, lextra_heating off.
 * Arrays in the reference's layouts: kappa_abs_LTE(p_n_cells, n_lambda),
 * log_Qcool_minus_extra_heating(n_T, p_n_cells), kdB_dT_CDF(n_lambda, n_T, p_n_cells).
 * ------------------------------------------------------------------------- */
/* -------------------------------------------------------------------------
 * repartition_energie (thermal_emission.f90:1771-1949), the LTE branch (:1814-1831) with the normalisation that
 * follows it (:1893-1942).
 * ------------------------------------------------------------------------- */
int oracle_repartition_energie(const oracle_model *m, int lambda, double wl_um, double E_star, double E_ISM,
                               const float *Tdust, const float *weight_proba_emission, double *frac_E_stars,
                               double *frac_E_disk, double *E_disk, double *prob_E_cell) {
  const double hp = 6.626070040e-34, c_light = 299792458.0, kb = 1.38064852e-23; /* constants.f90:21-23 */
  const float thermal_const = (float)(c_light * hp / kb);                         /* real, constants.f90:24 */
  const double tiny_dp = 2.2250738585072014e-308;
  const double cst_wl_max = log(HUGE_REAL) - (double)1.0e-4f;                     /* :1802 */
  const double wl = wl_um * (double)1.e-6f;                                       /* :1804, default-real literal */
  const int n_cells = m->n_cells;
  double sum_E = 0.0;
  prob_E_cell[0] = 0.0;                                                           /* :1923 */
  for (int icell = 1; icell <= n_cells; ++icell) {
    double E_cell = 0.0;
    if (!(m->l_dark_zone && m->l_dark_zone[icell - 1])) {                         /* :1816 */
      const double Temp = (double)Tdust[icell - 1];
      if (!(Temp < TINY_REAL)) {                                                  /* :1818 */
        const double cst_wl = (double)thermal_const / (Temp * wl);
        if (cst_wl < cst_wl_max) {
          const double kabs = m->p_n_cells > 0
                                  ? m->v_kappa_abs_LTE[(size_t)(m->p_icell[icell - 1] - 1) + (size_t)m->p_n_cells * (lambda - 1)]
                                  : m->kappa_abs_LTE[lambda - 1];
          const double wl2 = wl * wl, wl5 = (wl2 * wl2) * wl;                     /* wl**5 */
          E_cell = 4.0 * kabs * m->kappa_factor[icell - 1] * m->volume[icell - 1] / (wl5 * (exp(cst_wl) - 1.0)); /* :1823 */
        }
      }
    }
    sum_E = sum_E + E_cell;                                                       /* E_disk(lambda) = sum(E_cell), :1897 */
    const double corr = weight_proba_emission ? E_cell * (double)weight_proba_emission[icell - 1] : E_cell; /* :1888-1894 */
    prob_E_cell[icell] = prob_E_cell[icell - 1] + corr;                           /* :1924-1926 */
  }
  *E_disk = sum_E;
  if (E_star + sum_E + E_ISM < tiny_dp) return 1;                                 /* :1899-1903 */
  *frac_E_stars = E_star / (E_star + sum_E + E_ISM);                              /* :1905 */
  *frac_E_disk = (E_star + sum_E) / (E_star + sum_E + E_ISM);                     /* :1906 */
  const double last = prob_E_cell[n_cells];
  if (last > tiny_dp) for (int i = 0; i <= n_cells; ++i) prob_E_cell[i] = prob_E_cell[i] / last; /* :1933-1937 */
  else for (int i = 0; i <= n_cells; ++i) prob_E_cell[i] = 0.0;
  return 0;
}

int oracle_init_reemission(int p_n_cells, int n_T, int n_lambda, const float *tab_Temp, const double *tab_lambda,
                           const double *tab_delta_lambda, const double *kappa_abs_LTE, double *log_Qcool,
                           double *kdB_dT_CDF) {
  return oracle_init_reemission_ex(p_n_cells, n_T, n_lambda, tab_Temp, tab_lambda, tab_delta_lambda, kappa_abs_LTE, NULL,
                                   NULL, 0.0, log_Qcool, kdB_dT_CDF);
}

/* ... with lextra_heating (:486-494): dudt(icell), heating_norm(icell) = AU_to_m**2 * volume(icell) * kappa_factor(icell),
 * ufac_implicit > 0: ldudt_implicit */
int oracle_init_reemission_ex(int p_n_cells, int n_T, int n_lambda, const float *tab_Temp, const double *tab_lambda,
                              const double *tab_delta_lambda, const double *kappa_abs_LTE, const double *dudt,
                              const double *heating_norm, double ufac_implicit, double *log_Qcool, double *kdB_dT_CDF) {
  const double hp = 6.626070040e-34, c_light = 299792458.0, kb = 1.38064852e-23; /* constants.f90:21-23 */
  const float thermal_const = (float)(c_light * hp / kb);                         /* real, constants.f90:24 */
  const double cst_E = 2.0 * hp * (c_light * c_light) * (4.0 * M_PI);             /* :427 */
  const double tiny_dp = 2.2250738585072014e-308;
  double *B = (double *)calloc((size_t)n_lambda * n_T, sizeof(double));
  double *dB_dT = (double *)calloc((size_t)n_lambda * n_T, sizeof(double));
  double *integ3 = (double *)calloc((size_t)n_lambda + 1, sizeof(double));
  if (!B || !dB_dT || !integ3) { free(B); free(dB_dT); free(integ3); return 1; }
  /* the black body and its temperature derivative, bin width included (:431-452) */
  for (int t = 1; t <= n_T; ++t) {
    const double Temp = (double)tab_Temp[t - 1];
    const double cst = (double)thermal_const / Temp;
    for (int lambda = 1; lambda <= n_lambda; ++lambda) {
      const double wl = tab_lambda[lambda - 1] * (double)1.e-6f; /* default-real literal */
      const double delta_wl = tab_delta_lambda[lambda - 1] * (double)1.e-6f;
      const double cst_wl = cst / wl;
      const size_t k = (size_t)(lambda - 1) + (size_t)n_lambda * (t - 1);
      if (cst_wl < 500.0) {
        const double coeff_exp = exp(cst_wl);
        const double wl2 = wl * wl, wl5 = (wl2 * wl2) * wl; /* wl**5 */
        B[k] = 1.0 / (wl5 * (coeff_exp - 1.0)) * delta_wl;
        dB_dT[k] = B[k] * cst_wl * coeff_exp / (coeff_exp - 1.0);
      } else {
        B[k] = 0.0;
        dB_dT[k] = 0.0;
      }
    }
  }
  /* the cooling rate above the one at tab_Temp(1) (:464-513) */
  for (int icell = 1; icell <= p_n_cells; ++icell) {
    double Qcool0 = 0.0;
    for (int t = 1; t <= n_T; ++t) {
      double integ = 0.0;
      for (int lambda = 1; lambda <= n_lambda; ++lambda)
        integ = integ + kappa_abs_LTE[(size_t)(icell - 1) + (size_t)p_n_cells * (lambda - 1)] *
                            B[(size_t)(lambda - 1) + (size_t)n_lambda * (t - 1)];
      const double Qcool = integ * cst_E;
      if (t == 1) Qcool0 = Qcool;
      double extra_heating = Qcool0; /* .not.lextra_heating (:483-485) */
      if (dudt) {                     /* :486-494 */
        const double Temp = (double)tab_Temp[t - 1];
        const double h = (ufac_implicit > 0.0) ? (ufac_implicit * Temp - dudt[icell - 1]) / heating_norm[icell - 1]
                                               : dudt[icell - 1] / heating_norm[icell - 1];
        extra_heating = h > Qcool0 ? h : Qcool0;
      }
      const double q = Qcool - extra_heating;
      log_Qcool[(size_t)(t - 1) + (size_t)n_T * (icell - 1)] = (q > tiny_dp) ? log(q) : -1000.0;
    }
  }
  /* the re-emission CDF (:533-549) */
  for (int icell = 1; icell <= p_n_cells; ++icell)
    for (int t = 1; t <= n_T; ++t) {
      double *cdf = kdB_dT_CDF + (size_t)n_lambda * ((size_t)(t - 1) + (size_t)n_T * (icell - 1));
      integ3[0] = 0.0;
      for (int lambda = 1; lambda <= n_lambda; ++lambda)
        integ3[lambda] = integ3[lambda - 1] + kappa_abs_LTE[(size_t)(icell - 1) + (size_t)p_n_cells * (lambda - 1)] *
                                                  dB_dT[(size_t)(lambda - 1) + (size_t)n_lambda * (t - 1)];
      if (integ3[n_lambda] > tiny_dp)
        for (int lambda = 1; lambda <= n_lambda; ++lambda) cdf[lambda - 1] = integ3[lambda] / integ3[n_lambda];
      else
        for (int lambda = 1; lambda <= n_lambda; ++lambda) cdf[lambda - 1] = 0.0; /* the allocation value */
    }
  free(B); free(dB_dT); free(integ3);
  return 0;
}

/* ---------------------------------------------------------------------------------------------------------------
 * opacity (dust_prop.f90:791-1033) and calc_local_scattering_matrices (dust_prop.f90:1037-1243): the opacities and
 * the scattering tables of every cell class from the grains' cross sections and Mueller matrices and the local grain
 * densities -- the LTE / scattering-method-2 case, every wavelength with p_lambda = lambda.  The reference's types are
 * kept: default-real tables accumulate in default real (each sum is rounded to it, as the assignment does), the
 * opacities in double.  See mc_oracle.h. */
int oracle_opacity(int n_grains, int n_lambda, int p_n_cells, int nang, int aniso_method, int lsepar_pola,
                   int grain_RE_LTE_start, int grain_RE_LTE_end, const float *C_ext, const float *C_sca,
                   const float *C_abs, const float *tab_g, const float *tab_s11, const float *tab_s12,
                   const float *tab_s22, const float *tab_s33, const float *tab_s34, const float *tab_s44,
                   const float *S_grain, const double *nbre_grains, const double *dens, double *kappa,
                   double *kappa_abs_LTE, float *tab_albedo_pos, float *tab_g_pos, float *tab_s11_pos,
                   float *prob_s11_pos, float *s12_o_s11, float *s22_o_s11, float *s33_o_s11, float *s34_o_s11,
                   float *s44_o_s11) {
  const double AU_to_cm = 149597870700.0 * 100.0, mum_to_cm = 1.0e-4; /* constants.f90:61-73 */
  const double fact = AU_to_cm * (mum_to_cm * mum_to_cm);              /* :958 */
  const double dtheta = M_PI / (double)(float)nang;                    /* pi/real(nang_scatt), :1041 */
  const double two_pi = 2.0 * M_PI, four_pi = 4.0 * M_PI;
  const int na1 = nang + 1;
  if (n_grains < 1 || n_lambda < 1 || p_n_cells < 1 || nang < 2 || grain_RE_LTE_start < 1 || grain_RE_LTE_end > n_grains)
    return 1;
  if (aniso_method == 1 && (!tab_s11 || (lsepar_pola && (!tab_s12 || !tab_s22 || !tab_s33 || !tab_s34 || !tab_s44)))) return 1;
  for (int lambda = 1; lambda <= n_lambda; ++lambda) {
    const float *Ce = C_ext + (size_t)n_grains * (lambda - 1), *Cs = C_sca + (size_t)n_grains * (lambda - 1);
    const float *Ca = C_abs + (size_t)n_grains * (lambda - 1);
    const float *tg = tab_g ? tab_g + (size_t)n_grains * (lambda - 1) : NULL;
    for (int icell = 1; icell <= p_n_cells; ++icell) {
      const double *d = dens + (size_t)n_grains * (icell - 1);
      const size_t cl = (size_t)(icell - 1) + (size_t)p_n_cells * (lambda - 1); /* (p_n_cells, n_lambda) */
      /* ---- opacity(): the sums over the grains (:850-876) ---- */
      double kap = 0.0, k_sca_tot = 0.0;
      for (int k = 0; k < n_grains; ++k) {
        const double density = d[k] * nbre_grains[k];
        kap = kap + (double)Ce[k] * density;
        k_sca_tot = k_sca_tot + (double)Cs[k] * density;
      }
      float albedo = 0.0f; /* (the array is zero before the first call) */
      if (kap > (double)FLT_MIN) albedo = (float)(k_sca_tot / kap);
      float g_pos = 0.0f;
      if (aniso_method == 2) {
        for (int k = 0; k < n_grains; ++k) {
          const double density = d[k] * nbre_grains[k];
          g_pos = (float)((double)g_pos + ((double)Cs[k] * density) * (double)tg[k]);
        }
        if (k_sca_tot > (double)FLT_MIN) g_pos = (float)((double)g_pos / k_sca_tot);
        tab_g_pos[cl] = g_pos;
      } else if (tab_g_pos) tab_g_pos[cl] = 0.0f;
      double kabs = 0.0;
      for (int k = grain_RE_LTE_start - 1; k < grain_RE_LTE_end; ++k) kabs = kabs + ((double)Ca[k] * d[k]) * nbre_grains[k];
      kap = kap * fact;   /* :960-961 */
      kabs = kabs * fact;
      kappa[cl] = kap;
      kappa_abs_LTE[cl] = kabs;
      /* ---- calc_local_scattering_matrices() (scattering_method 2) ---- */
      const size_t row = (size_t)na1 * cl; /* (0:nang, p_n_cells, p_n_lambda) */
      float *s11 = tab_s11_pos + row, *prob = prob_s11_pos ? prob_s11_pos + row : NULL;
      float *m12 = lsepar_pola ? s12_o_s11 + row : NULL, *m22 = lsepar_pola ? s22_o_s11 + row : NULL;
      float *m33 = lsepar_pola ? s33_o_s11 + row : NULL, *m34 = lsepar_pola ? s34_o_s11 + row : NULL;
      float *m44 = lsepar_pola ? s44_o_s11 + row : NULL;
      if (aniso_method == 1) {
        for (int l = 0; l < na1; ++l) {
          s11[l] = 0.0f;
          if (lsepar_pola) { m12[l] = 0.0f; m22[l] = 0.0f; m33[l] = 0.0f; m34[l] = 0.0f; m44[l] = 0.0f; }
        }
        for (int k = 0; k < n_grains; ++k) { /* Mueller matrix averaging (:1098-1120) */
          const double density = d[k] * nbre_grains[k];
          const size_t g0 = (size_t)na1 * ((size_t)k + (size_t)n_grains * (lambda - 1));
          for (int l = 0; l < na1; ++l) {
            s11[l] = (float)((double)s11[l] + (double)(tab_s11[g0 + l] * S_grain[k]) * density);
            if (lsepar_pola) {
              m12[l] = (float)((double)m12[l] + (double)(tab_s12[g0 + l] * S_grain[k]) * density);
              m22[l] = (float)((double)m22[l] + (double)(tab_s22[g0 + l] * S_grain[k]) * density);
              m33[l] = (float)((double)m33[l] + (double)(tab_s33[g0 + l] * S_grain[k]) * density);
              m34[l] = (float)((double)m34[l] + (double)(tab_s34[g0 + l] * S_grain[k]) * density);
              m44[l] = (float)((double)m44[l] + (double)(tab_s44[g0 + l] * S_grain[k]) * density);
            }
          }
        }
      }
      k_sca_tot = kap * (double)albedo / fact; /* :1122 */
      if (k_sca_tot > (double)FLT_MIN) {
        if (aniso_method == 1) {
          prob[0] = 0.0f;
          prob[1] = 0.0f; /* (never assigned by the loop below: the allocation value) */
          for (int l = 2; l <= nang; ++l) {
            const double theta = (double)(float)l * dtheta;
            prob[l] = (float)((double)prob[l - 1] + ((double)s11[l] * sin(theta)) * dtheta);
          }
          const double last = (double)prob[nang];
          for (int l = 1; l <= nang; ++l) prob[l] = (float)(((double)prob[l] + k_sca_tot) - last); /* :1150 */
          for (int l = 0; l <= nang; ++l) prob[l] = (float)((double)prob[l] / k_sca_tot);
          for (int l = 0; l <= nang; ++l)
            if (s11[l] > FLT_MIN) {
              const float norm = 1.0f / s11[l];
              if (lsepar_pola) { m12[l] *= norm; m22[l] *= norm; m33[l] *= norm; m34[l] *= norm; m44[l] *= norm; }
            }
          for (int l = 0; l <= nang; ++l) s11[l] = (float)(((double)s11[l] * dtheta) / (k_sca_tot * two_pi)); /* :1172 */
        } else {
          for (int l = 0; l <= nang; ++l) { /* Henyey-Greenstein, for the ray tracer (:1190-1194) */
            const float g = g_pos, g2 = g * g;
            const float mu = (float)cos((double)((float)l / (float)nang) * M_PI);
            s11[l] = (float)((((1.0 / four_pi) * (double)(1.0f - g2)) * (double)powf((1.0f + g2) - (2.0f * g) * mu, -1.5f)) * dtheta);
          }
          if (prob) for (int l = 0; l <= nang; ++l) prob[l] = 0.0f; /* (not built for this method) */
          if (lsepar_pola)
            for (int l = 0; l <= nang; ++l) { m12[l] = 0.0f; m22[l] = 0.0f; m33[l] = 0.0f; m34[l] = 0.0f; m44[l] = 0.0f; }
        }
      } else { /* no scattering here (:1222-1236) */
        albedo = 0.0f;
        if (prob) { for (int l = 0; l <= nang; ++l) prob[l] = 1.0f; prob[0] = 0.0f; }
        for (int l = 0; l <= nang; ++l) s11[l] = 1.0f;
        if (lsepar_pola)
          for (int l = 0; l <= nang; ++l) { m12[l] = 0.0f; m22[l] = 0.0f; m33[l] = 0.0f; m34[l] = 0.0f; m44[l] = 0.0f; }
      }
      tab_albedo_pos[cl] = albedo;
    }
  }
  return 0;
}
