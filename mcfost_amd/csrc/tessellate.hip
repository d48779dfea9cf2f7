// Translation unit of the device tessellator (mc_tessellate.hip.h): the C-ABI entry that stands where the reference
// calls voro_C (Voronoi.f90:70-96 interface, :487-520 call; voro++_wrapper.cpp:43-277).
#include "../../include/mcgpu.h"

#include <hip/hip_runtime.h>

#include <cstring>

#include "mc_tessellate.hip.h"

using namespace mcgpu;

namespace {
template <typename T>
struct Dev {
  T* p = nullptr;
  ~Dev() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};
}  // namespace

#define TCHK(call) do { if ((call) != hipSuccess) return MCGPU_ERR_HIP; } while (0)

extern "C" int mcgpu_voronoi_tesselation(int device, int n, const double* xyz, const double* h, const double limits[6],
                                         double threshold, int n_vectors, const double* cutting_vectors,
                                         double cutting_distance_o_h, int n_run, const int* cells, int k, const int* knn,
                                         const int* knn_first, const double* extra_plane, int max_neighbours, int* n_neigh, int* neigh,
                                         double* volume, double* delta_edge, unsigned char* was_cut, double* kernel_ms,
                                         double* volume_uncut) {
  if (n < 1 || !xyz || !h || !limits || n_run < 0 || (k < 1 && !knn_first) || !knn || max_neighbours < 4 || !n_neigh || !neigh || !volume ||
      !delta_edge || !was_cut || n_vectors < 0 || n_vectors > 20 || (n_vectors > 0 && !cutting_vectors) || !(threshold > 0.0))
    return MCGPU_ERR_ARG;
  // The kernel keeps a vertex as the THREE planes that meet in it (general position).  The 12 cutting planes of the
  // dodecahedron -- the reference's solid, Voronoi.f90:243 -- meet three to a vertex; the 20 of the icosahedron's dual
  // meet five to a vertex, which the clipping does not represent: refused rather than cut wrongly.
  if (n_vectors > 12) return MCGPU_ERR_UNSUPPORTED;
  int n_have = 0;
  if (hipGetDeviceCount(&n_have) != hipSuccess || n_have <= 0) return MCGPU_ERR_NO_DEVICE;
  if (device < 0 || device >= n_have) return MCGPU_ERR_ARG;
  if (n_run == 0) { if (kernel_ms) *kernel_ms = 0.0; return MCGPU_OK; }
  TCHK(hipSetDevice(device));
  TessArgs A;
  std::memset(&A, 0, sizeof(A));
  A.n = n; A.threshold = threshold; A.n_vectors = n_vectors; A.cutting_distance_o_h = cutting_distance_o_h;
  A.k = k; A.n_run = n_run; A.max_neighbours = max_neighbours;
  for (int i = 0; i < 6; ++i) A.limits[i] = limits[i];
  for (int v = 0; v < n_vectors; ++v)
    for (int c = 0; c < 3; ++c) A.cut_vec[v][c] = cutting_vectors[3 * v + c];
  Dev<double> d_xyz, d_h, d_extra, d_vol, d_edge, d_vol0;
  Dev<int> d_knn, d_cells, d_nn, d_neigh, d_first;
  const size_t n_knn = knn_first ? (size_t)knn_first[n_run] : (size_t)n_run * k;
  Dev<unsigned char> d_cut;
  TCHK(d_xyz.alloc(3 * (size_t)n)); TCHK(d_h.alloc(n)); TCHK(d_knn.alloc(n_knn));
  TCHK(d_nn.alloc(n_run)); TCHK(d_neigh.alloc((size_t)n_run * max_neighbours)); TCHK(d_vol.alloc(n_run));
  TCHK(d_edge.alloc(n_run)); TCHK(d_cut.alloc(n_run));
  TCHK(hipMemcpy(d_xyz.p, xyz, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  TCHK(hipMemcpy(d_h.p, h, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  TCHK(hipMemcpy(d_knn.p, knn, n_knn * sizeof(int), hipMemcpyHostToDevice));
  if (knn_first) {
    TCHK(d_first.alloc((size_t)n_run + 1));
    TCHK(hipMemcpy(d_first.p, knn_first, ((size_t)n_run + 1) * sizeof(int), hipMemcpyHostToDevice));
  }
  if (cells) {
    TCHK(d_cells.alloc(n_run));
    TCHK(hipMemcpy(d_cells.p, cells, (size_t)n_run * sizeof(int), hipMemcpyHostToDevice));
  } else if (n_run > n) return MCGPU_ERR_ARG;
  if (extra_plane) {
    TCHK(d_extra.alloc(4 * (size_t)n_run));
    TCHK(hipMemcpy(d_extra.p, extra_plane, 4 * (size_t)n_run * sizeof(double), hipMemcpyHostToDevice));
  }
  TCHK(hipMemset(d_neigh.p, 0, (size_t)n_run * max_neighbours * sizeof(int)));
  TCHK(hipMemset(d_vol.p, 0, (size_t)n_run * sizeof(double)));
  TCHK(hipMemset(d_edge.p, 0, (size_t)n_run * sizeof(double)));
  TCHK(hipMemset(d_cut.p, 0, (size_t)n_run));
  A.knn_first = d_first.p;
  A.xyz = d_xyz.p; A.h = d_h.p; A.knn = d_knn.p; A.cells = d_cells.p; A.extra_plane = d_extra.p;
  if (volume_uncut) { TCHK(d_vol0.alloc(n_run)); TCHK(hipMemset(d_vol0.p, 0, (size_t)n_run * sizeof(double))); A.volume_uncut = d_vol0.p; }
  A.n_neigh = d_nn.p; A.neigh = d_neigh.p; A.volume = d_vol.p; A.delta_edge = d_edge.p; A.was_cut = d_cut.p;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess) return MCGPU_ERR_HIP;
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return MCGPU_ERR_HIP; }
  hipError_t re = hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_voronoi_cells, dim3((unsigned)((n_run + 63) / 64)), dim3(64), 0, 0, A);
  hipError_t le = hipGetLastError();
  if (re == hipSuccess) re = hipEventRecord(e1, 0);
  hipError_t se = hipDeviceSynchronize();
  float ms = 0.f;
  if (re == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (re != hipSuccess || le != hipSuccess || se != hipSuccess) return MCGPU_ERR_HIP;
  if (kernel_ms) *kernel_ms = ms;
  TCHK(hipMemcpy(n_neigh, d_nn.p, (size_t)n_run * sizeof(int), hipMemcpyDeviceToHost));
  TCHK(hipMemcpy(neigh, d_neigh.p, (size_t)n_run * max_neighbours * sizeof(int), hipMemcpyDeviceToHost));
  TCHK(hipMemcpy(volume, d_vol.p, (size_t)n_run * sizeof(double), hipMemcpyDeviceToHost));
  if (volume_uncut) TCHK(hipMemcpy(volume_uncut, d_vol0.p, (size_t)n_run * sizeof(double), hipMemcpyDeviceToHost));
  TCHK(hipMemcpy(delta_edge, d_edge.p, (size_t)n_run * sizeof(double), hipMemcpyDeviceToHost));
  TCHK(hipMemcpy(was_cut, d_cut.p, (size_t)n_run, hipMemcpyDeviceToHost));
  return MCGPU_OK;
}
