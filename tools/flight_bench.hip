// Micro-benchmark (tuning aid): the crossing loop alone (cross_cell_lean + LDS deposit +
// kappa_factor prefetch) at different occupancies, to price what a leaner-register flight
// kernel would buy.  Straight flights from random cell centres until exit, then restart.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstring>
#include "../mcfost_amd/csrc/mc_device.hip.h"
using namespace mcgpu;

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_fly(const DevModel M, int iters, unsigned long long* out) {
  extern __shared__ double lds_raw[];
  double* E_lds = lds_raw;
  const Lds T = lds_carve(lds_raw + M.n_cells, M);
  lds_stage(T, M);
  for (int i = threadIdx.x; i < M.n_cells; i += blockDim.x) E_lds[i] = 0.0;
  __syncthreads();
  uint32_t s = (blockIdx.x * BLOCK + threadIdx.x) * 2654435761u + 99u;
  double x = 0, y = 0, z = 0, u = 1, v = 0, w = 0, inv_a = 1, inv_w = 1, kf = 0, extr = 0;
  int ri = M.n_rad + 1, zj = 1, k = 1;
  unsigned long long n = 0;
  for (int it = 0; it < iters; ++it) {
    const bool out_ = (ri == M.n_rad + 1) || ((zj == M.nz + 1) && (fabs(z) > M.zmaxmax));
    if (out_ || extr <= 0.0) {  // restart: cheap pseudo-emission / pseudo-interaction
      s = s * 1664525u + 1013904223u;
      if (out_) {
        ri = 1 + (s >> 8) % M.n_rad; zj = 1 + (s >> 20) % 8; k = 1;
        const double r = sqrt(0.5 * (T.r_lim_2[ri - 1] + T.r_lim_2[ri]));
        x = r; y = 0.0; z = ((double)zj - 0.5) * T.ch[ri - 1];
      }
      const float a = (float)(s >> 8) * (1.0f / 16777216.0f);
      s = s * 1664525u + 1013904223u;
      const float b = (float)(s >> 8) * (1.0f / 16777216.0f);
      w = 2.0 * a - 1.0;
      const double uv = sqrt(1.0 - w * w);
      float sb, cb; __sincosf(6.2831853f * b, &sb, &cb);
      u = uv * cb; v = uv * sb;
      const double aa = u * u + v * v;
      inv_a = (aa > TINY_REAL) ? 1.0 / aa : HUGE_REAL;
      inv_w = (fabs(w) > TINY_REAL) ? 1.0 / w : copysign(HUGE_DP, w);
      extr = 0.05 + 3.0 * b;
      kf = M.kappa_factor[cell_index<false>(M.n_rad, M.nz, ri, zj, k)];
    }
    double x1, y1, z1, l; int ri1, zj1, k1;
    cross_cell_lean<false>(T, M, x, y, z, u, v, w, inv_a, inv_w, ri, zj, k, x1, y1, z1, ri1, zj1, k1, l);
    n++;
    const bool real_cell = is_real_cell<false>(M.n_rad, M.nz, ri, zj);
    const double tau = l * T.kappa[7] * kf;
    if (tau > extr) {
      const double lc = l * (extr / tau);
      if (real_cell) atomicAdd(&E_lds[cell_index<false>(M.n_rad, M.nz, ri, zj, k)], lc);
      x += lc * u; y += lc * v; z += lc * w; extr = 0.0;
    } else {
      extr -= tau;
      if (real_cell) atomicAdd(&E_lds[cell_index<false>(M.n_rad, M.nz, ri, zj, k)], l);
      x = x1; y = y1; z = z1; ri = ri1; zj = zj1; k = k1;
      kf = is_real_cell<false>(M.n_rad, M.nz, ri, zj) ? M.kappa_factor[cell_index<false>(M.n_rad, M.nz, ri, zj, k)] : 0.0;
    }
  }
  __syncthreads();
  unsigned long long t = n;
  for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, t + (threadIdx.x == 0 ? (unsigned long long)(E_lds[5] > 1e300) : 0ull));
}

template <typename T> T* up(const std::vector<T>& h) { T* d; hipMalloc(&d, h.size() * sizeof(T)); hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); return d; }

template <int BLOCK> void run(const DevModel& M, int iters) {
  unsigned long long* out; hipMalloc(&out, 8); hipMemset(out, 0, 8);
  const size_t lds = lds_bytes(M) + (size_t)M.n_cells * 8;
  hipFuncSetAttribute((const void*)k_fly<BLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k_fly<BLOCK><<<256, BLOCK, lds>>>(M, iters / 10, out); hipDeviceSynchronize(); hipMemset(out, 0, 8);
  hipEventRecord(a); k_fly<BLOCK><<<256, BLOCK, lds>>>(M, iters, out); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long n; hipMemcpy(&n, out, 8, hipMemcpyDeviceToHost);
  hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)k_fly<BLOCK>);
  printf("block %4d  vgpr %3d  %8.2f ms  %.3e crossings/s (%s)\n", BLOCK, fa.numRegs, ms, n / (ms * 1e-3), hipGetErrorString(hipGetLastError()));
}

int main() {
  const int n_rad = 100, nz = 70, n_cells = 7000, nl = 50, nT = 100, nang = 180;
  std::vector<double> r2(n_rad + 1), zmax(n_rad), ch(n_rad), tp(1, 0.0), kf(n_cells), dl(nl, 1e-3), dT(nT, 0.0), cdf(nl * nT, 0.0), cum(nl + 1, 0.0), ct(nang + 1, 0.0);
  std::vector<float> fl(nl, 0.5f), prob(nang + 1, 0.f);
  for (int i = 0; i <= n_rad; ++i) { double r = std::pow(300.0, i / 100.0); r2[i] = r * r; }
  for (int i = 0; i < n_rad; ++i) { double r = 0.5 * (std::sqrt(r2[i]) + std::sqrt(r2[i + 1])); zmax[i] = 7 * 10.0 * std::pow(r / 100.0, 1.125); ch[i] = zmax[i] / nz; }
  for (int c = 0; c < n_cells; ++c) { int j = c / n_rad; kf[c] = std::exp(-0.5 * (j * 7.0 / nz) * (j * 7.0 / nz)); }
  DevModel M; memset(&M, 0, sizeof(M));
  M.n_rad = n_rad; M.nz = nz; M.n_az = 1; M.n_cells = n_cells; M.r_lim_2 = up(r2); M.zmax = up(zmax); M.ch = up(ch); M.tan_phi_lim = up(tp);
  M.zmaxmax = zmax[n_rad - 1]; M.Rmax2 = r2[n_rad]; M.kappa_factor = up(kf); M.n_lambda = nl; M.kappa = up(dl); M.kappa_abs = up(dl); M.albedo = up(fl);
  M.nang = nang; M.p_lambda_fixed = 1; M.prob_s11 = up(prob); M.tab_g = up(fl); M.cos_tab = up(ct); M.n_T = nT; M.log_Qcool = up(dT); M.cdf = up(cdf); M.spec_cum = up(cum); M.frac_E_stars = up(dl);
  const int iters = 20000;
  run<256>(M, iters); run<512>(M, iters); run<768>(M, iters); run<1024>(M, iters);
  return 0;
}
