// opacity (dust_prop.f90:791-1033) and calc_local_scattering_matrices (dust_prop.f90:1037-1243) on the device: the
// opacities and scattering tables of every cell class from the grains' cross sections / Mueller matrices and the local
// grain densities (with lvariable_dust the tables gain a cell axis -- 181 x 7000 x 50 x 4 B = 253 MB
// per Mueller element -- and the sums over 100 grain sizes are what the reference calls "CPU intensive").
// LTE grains, scattering_method 2, every wavelength with p_lambda = lambda; the reference's types: default-real tables
// are rounded to default real after every term of a sum (the assignment does that), opacities are double.  No
// contraction to FMA in here: the sums then equal the CPU restatement's (oracle_opacity) bit for bit; only sin / cos /
// powf can differ by a unit in the last place of the default-real results.
// Inputs in the reference's layouts: C_*(n_grains, n_lambda), tab_s1x(0:nang, n_grains, n_lambda),
// dust_density_o_n_grains(n_grains, p_n_cells).  Outputs in the context's layouts ([class][lambda], [class][lambda][angle]).
// HBM traffic: the density rows and the grain tables are re-read by neighbouring threads from L2; what must move is the
// output, (1 + 1 + 5) tables x 4 B per (angle, class, wavelength).
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>

struct OpacityIn {
  int n_grains, n_lambda, n_classes, nang, aniso_method, lsepar_pola, re_lte_start, re_lte_end, prob_cols;
  const float *C_ext, *C_sca, *C_abs, *tab_g;
  const float *s11, *s12, *s22, *s33, *s34, *s44;
  const float* S_grain;
  const double* nbre_grains;
  const double* dens;
};
struct OpacityOut {
  double *kappa, *kabs;    // [class][lambda]
  float *albedo, *g;       // [class][lambda]
  float *s11, *prob;       // [class][lambda][0:nang]; prob: [class][prob_cols][0:nang]
  float *m12, *m22, *m33, *m34, *m44;
};

constexpr double OPA_AU_TO_CM = 149597870700.0 * 100.0;  // constants.f90:61-64
constexpr double OPA_FACT = OPA_AU_TO_CM * (1.0e-4 * 1.0e-4);  // AU_to_cm * mum_to_cm**2 (dust_prop.f90:958)

// one thread per (class, wavelength): the sums of opacity() (dust_prop.f90:850-876, :960-961)
static __global__ void k_opacity_sum(const OpacityIn I, const OpacityOut O) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= I.n_classes * I.n_lambda) return;
  const int c = i / I.n_lambda, l = i - c * I.n_lambda, ng = I.n_grains;
  const double* d = I.dens + (size_t)ng * c;
  const float *Ce = I.C_ext + (size_t)ng * l, *Cs = I.C_sca + (size_t)ng * l, *Ca = I.C_abs + (size_t)ng * l;
  double kap = 0.0, ksca = 0.0;
  for (int k = 0; k < ng; ++k) {
    const double density = d[k] * I.nbre_grains[k];
    kap = kap + (double)Ce[k] * density;
    ksca = ksca + (double)Cs[k] * density;
  }
  float albedo = 0.0f;
  if (kap > (double)FLT_MIN) albedo = (float)(ksca / kap);
  float g = 0.0f;
  if (I.aniso_method == 2) {
    const float* tg = I.tab_g + (size_t)ng * l;
    for (int k = 0; k < ng; ++k) {
      const double density = d[k] * I.nbre_grains[k];
      g = (float)((double)g + ((double)Cs[k] * density) * (double)tg[k]);
    }
    if (ksca > (double)FLT_MIN) g = (float)((double)g / ksca);
  }
  double kabs = 0.0;
  for (int k = I.re_lte_start - 1; k < I.re_lte_end; ++k) kabs = kabs + ((double)Ca[k] * d[k]) * I.nbre_grains[k];
  O.kappa[i] = kap * OPA_FACT;
  O.kabs[i] = kabs * OPA_FACT;
  O.albedo[i] = albedo;
  O.g[i] = g;
}

// one thread per (angle, class, wavelength): the Mueller matrices averaged over the grains (dust_prop.f90:1098-1120);
// the angle is the fastest thread index (coalesced reads of the grains' tables, coalesced writes), the density of
// (grain, class) is the same for the whole row
template <bool POLA>
__global__ void k_scatt_sum(const OpacityIn I, const OpacityOut O) {
#pragma clang fp contract(off)
  const int na1 = I.nang + 1;
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y, l = blockIdx.z, ng = I.n_grains;
  if (a >= na1) return;
  const double* d = I.dens + (size_t)ng * c;
  float s11 = 0.0f, m12 = 0.0f, m22 = 0.0f, m33 = 0.0f, m34 = 0.0f, m44 = 0.0f;
  for (int k = 0; k < ng; ++k) {
    const double density = d[k] * I.nbre_grains[k];
    const float S = I.S_grain[k];
    const size_t at = (size_t)na1 * ((size_t)k + (size_t)ng * l) + a;
    s11 = (float)((double)s11 + (double)(I.s11[at] * S) * density);
    if (POLA) {
      m12 = (float)((double)m12 + (double)(I.s12[at] * S) * density);
      m22 = (float)((double)m22 + (double)(I.s22[at] * S) * density);
      m33 = (float)((double)m33 + (double)(I.s33[at] * S) * density);
      m34 = (float)((double)m34 + (double)(I.s34[at] * S) * density);
      m44 = (float)((double)m44 + (double)(I.s44[at] * S) * density);
    }
  }
  const size_t o = ((size_t)c * I.n_lambda + l) * na1 + a;
  O.s11[o] = s11;
  if (POLA) { O.m12[o] = m12; O.m22[o] = m22; O.m33[o] = m33; O.m34[o] = m34; O.m44[o] = m44; }
}

// one thread per (class, wavelength): the cumulative scattering probability and the normalisations
// (dust_prop.f90:1122-1236) -- sequential along the angle like the reference (every partial sum is rounded to default real)
static __global__ void k_scatt_norm(const OpacityIn I, const OpacityOut O) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= I.n_classes * I.n_lambda) return;
  const int c = i / I.n_lambda, l = i - c * I.n_lambda, nang = I.nang, na1 = nang + 1;
  const double pi = 3.14159265358979323846, two_pi = 2.0 * pi, four_pi = 4.0 * pi;
  const double dtheta = pi / (double)(float)nang;
  float* s11 = O.s11 + (size_t)i * na1;
  float* prob = (l < I.prob_cols) ? O.prob + ((size_t)c * I.prob_cols + l) * na1 : nullptr;
  const bool pola = I.lsepar_pola != 0;
  float *m12 = O.m12 + (size_t)i * na1, *m22 = O.m22 + (size_t)i * na1, *m33 = O.m33 + (size_t)i * na1;
  float *m34 = O.m34 + (size_t)i * na1, *m44 = O.m44 + (size_t)i * na1;
  const float albedo = O.albedo[i];
  const double k_sca_tot = O.kappa[i] * (double)albedo / OPA_FACT;
  if (k_sca_tot > (double)FLT_MIN) {
    if (I.aniso_method == 1) {
      // (the sums run whether or not this column of prob_s11_pos is kept: p_lambda_fixed keeps the first only)
      float p_prev = 0.0f, p_last = 0.0f;
      for (int a = 2; a <= nang; ++a) {
        const double theta = (double)(float)a * dtheta;
        p_prev = (float)((double)p_prev + ((double)s11[a] * sin(theta)) * dtheta);
        if (prob) prob[a] = p_prev;
      }
      p_last = p_prev;
      if (prob) {
        prob[0] = 0.0f;
        prob[1] = 0.0f;
        for (int a = 1; a <= nang; ++a) prob[a] = (float)(((double)prob[a] + k_sca_tot) - (double)p_last);
        for (int a = 0; a <= nang; ++a) prob[a] = (float)((double)prob[a] / k_sca_tot);
      }
      for (int a = 0; a <= nang; ++a) {
        const float s = s11[a];
        if (pola && s > FLT_MIN) {
          const float norm = 1.0f / s;
          m12[a] *= norm; m22[a] *= norm; m33[a] *= norm; m34[a] *= norm; m44[a] *= norm;
        }
        s11[a] = (float)(((double)s * dtheta) / (k_sca_tot * two_pi));
      }
    } else {
      const float g = O.g[i], g2 = g * g;
      for (int a = 0; a <= nang; ++a) {
        const float mu = (float)cos((double)((float)a / (float)nang) * pi);
        s11[a] = (float)((((1.0 / four_pi) * (double)(1.0f - g2)) * (double)powf((1.0f + g2) - (2.0f * g) * mu, -1.5f)) * dtheta);
        if (prob) prob[a] = 0.0f;
        if (pola) { m12[a] = 0.0f; m22[a] = 0.0f; m33[a] = 0.0f; m34[a] = 0.0f; m44[a] = 0.0f; }
      }
    }
  } else {
    O.albedo[i] = 0.0f;
    for (int a = 0; a <= nang; ++a) {
      s11[a] = 1.0f;
      if (prob) prob[a] = a ? 1.0f : 0.0f;
      if (pola) { m12[a] = 0.0f; m22[a] = 0.0f; m33[a] = 0.0f; m34[a] = 0.0f; m44[a] = 0.0f; }
    }
  }
}

// ksca_CDF(0:n_grains, p_n_cells, n_lambda) (dust_prop.f90:24, built at :976-994 when scattering method 1 has the memory
// for it, mem.f90:245-258): per (class, wavelength) the running sum of C_sca(k, lambda) dust_density_o_n_grains(k, class)
// n_grains(k) over the grain sizes, normalised by its last entry -- or all ones where that is not positive ("at the
// surface ... only the smallest grains").  One thread per (class, wavelength), the sum in the reference's order and
// association, unfused.  Device layout [class][lambda][0:n_grains]: the dichotomy of select_grainsize_high_mem walks one row.
static __global__ void k_ksca_cdf(int n_classes, int n_lambda, int ng, const float* C_sca, const double* dens, const double* nk,
                                  double* out) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_classes * n_lambda) return;
  const int cls = i / n_lambda, l = i % n_lambda;
  double* row = out + (size_t)i * (ng + 1);
  double c = 0.0;
  row[0] = 0.0;
  for (int k = 1; k <= ng; ++k) {
    c = c + (double)C_sca[(size_t)(k - 1) + (size_t)ng * l] * dens[(size_t)(k - 1) + (size_t)ng * cls] * nk[k - 1];
    row[k] = c;
  }
  if (c > (double)1.17549435082228750797e-38f) {
    for (int k = 0; k <= ng; ++k) row[k] = row[k] / c;
  } else {
    for (int k = 0; k <= ng; ++k) row[k] = 1.0;
  }
}
