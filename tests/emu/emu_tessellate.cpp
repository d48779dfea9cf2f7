// emu_tessellate.cpp -- TEST INFRASTRUCTURE: compiles the device tessellator (mcfost_amd/csrc/mc_tessellate.hip.h) for
// the host, one emulated lane per cell, so that the clipping code can be checked against scipy.spatial on the CPU.  Not
// a CPU path of the product: nothing in mcfost_amd/ references it; built only by tests/test_tessellation.py.
#define MCGPU_LANE_EMULATION 1
#include <math.h>
#include <string.h>

#include <cmath>

#define __device__
#define __host__
#define __global__
#define __launch_bounds__(...)
struct emu_dim3 { unsigned x, y, z; };
static emu_dim3 threadIdx{0, 0, 0}, blockIdx{0, 0, 0}, blockDim{1, 1, 1};
using std::fabs; using std::sqrt; using std::atan2;

#include "../../mcfost_amd/csrc/mc_tessellate.hip.h"

extern "C" int emu_voronoi_tesselation(int n, const double* xyz, const double* h, const double* limits, double threshold,
                                       int n_vectors, const double* cutting_vectors, double cutting_distance_o_h, int n_run,
                                       const int* cells, int k, const int* knn, const int* knn_first, const double* extra_plane, int max_neighbours,
                                       int* n_neigh, int* neigh, double* volume, double* delta_edge, unsigned char* was_cut,
                                       double* volume_uncut) {
  mcgpu::TessArgs A;
  memset(&A, 0, sizeof(A));
  A.n = n; A.xyz = xyz; A.h = h; A.threshold = threshold; A.n_vectors = n_vectors; A.cutting_distance_o_h = cutting_distance_o_h;
  for (int i = 0; i < 6; ++i) A.limits[i] = limits[i];
  for (int v = 0; v < n_vectors; ++v) for (int c = 0; c < 3; ++c) A.cut_vec[v][c] = cutting_vectors[3 * v + c];
  A.k = k; A.knn = knn; A.knn_first = knn_first; A.cells = cells; A.n_run = n_run; A.extra_plane = extra_plane; A.max_neighbours = max_neighbours;
  A.volume_uncut = volume_uncut;
  A.n_neigh = n_neigh; A.neigh = neigh; A.volume = volume; A.delta_edge = delta_edge; A.was_cut = was_cut;
  for (int r = 0; r < n_run; ++r) { blockIdx.x = (unsigned)r; mcgpu::k_voronoi_cells(A); }
  return 0;
}
