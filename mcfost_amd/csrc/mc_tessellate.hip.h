// Voronoi tessellation on the device: what voro_C (voro++_wrapper.cpp:43-277, called from Voronoi_tesselation,
// Voronoi.f90:441-520) returns for the packet loop -- per cell its neighbours (sites, or -1 .. -6 for the walls of the
// box), its volume (after the cut of elongated cells by the Platonic solid, voro++_wrapper.cpp:209-227, and of a
// star's neighbours at the stellar surface, :229-262), the largest vertex distance (delta_edge) and was_cell_cut.
//
// voro++ builds every cell on its own -- the box clipped by the bisector planes of the sites around, nearest first,
// until no remaining site can reach the cell -- and that is embarrassingly parallel: ONE THREAD PER CELL here, the
// candidates taken from a k-nearest-neighbour list sorted by distance (the host's kd-tree; the reference links kdtree2
// for the same kind of search, Voronoi.f90:1625), the cell kept in the dual form of Ray, Sokolov, Lefebvre & Levy
// (2018, "Meshless Voronoi on the GPU"): a list of planes and a list of vertices, each vertex = the three planes that
// meet in it.  Clipping by a plane removes the vertices beyond it and adds one vertex for every edge that crosses it.
// A cell is COMPLETE once the next candidate is farther than twice the farthest vertex (no bisector can cut it any
// more); a cell whose k candidates do not get that far is reported incomplete and the host asks again with more.
// The state of a cell lives in the thread's private (scratch) memory: 6.6 KB, L2-resident; the work is ~1e5 FP64
// operations per cell, i.e. a millisecond-scale kernel per 1e5 cells -- the host's neighbour search dominates.
//
// Coordinates are relative to the cell's site, planes are (unit normal, distance): the site is the origin and lies
// inside every half-space.  Sites in general position (SPH particles); a vertex that lies ON a clipping plane to
// rounding counts as inside, so a plane that merely touches the cell adds no face.
#ifndef MCFOST_AMD_MC_TESSELLATE_HIP_H
#define MCFOST_AMD_MC_TESSELLATE_HIP_H

#ifndef MCGPU_LANE_EMULATION  // tests/emu compiles this header for one emulated lane on the CPU
#include <hip/hip_runtime.h>
#endif

namespace mcgpu {

constexpr int TESS_MAX_P = 96;    // planes that are (or were) faces of a cell, walls and cut planes included
constexpr int TESS_MAX_T = 160;   // vertices of a cell
constexpr int TESS_ID_WALL0 = -1;         // ids of the walls: -1 .. -6 (Voronoi.f90:560-600: a negative neighbour is a wall)
constexpr int TESS_ID_CUT = -1000;        // planes of the Platonic solid / the stellar surface: no neighbour

struct TessArgs {
  int n;                        // sites (stars last)
  const double* xyz;            // [n][3]
  const double* h;              // [n] smoothing lengths (huge for stars: never cut)
  double limits[6];             // xmin, xmax, ymin, ymax, zmin, zmax
  double threshold;             // a cell whose farthest vertex is beyond threshold * h is cut (Voronoi.f90:233: 3)
  int n_vectors;                // faces of the Platonic solid (<= 12: planes that meet three to a vertex, the dodecahedron)
  double cut_vec[20][3];        // their unit normals (init_Platonic_Solid, Voronoi.f90:108-181)
  double cutting_distance_o_h;  // distance of those faces in units of h
  int k;                        // candidates per cell (rows of knn), or 0 with knn_first
  const int* knn;               // [n_run][k] 0-based site ids by increasing distance, the site itself excluded; -1: none
  const int* knn_first;         // nullptr, or [n_run + 1] offsets into knn: row r is knn[knn_first[r] .. knn_first[r + 1]) and
                                // holds EVERY site that can cut the cell (its Delaunay neighbours): no security radius
  const int* cells;             // [n_run] the cells to build (nullptr: 0 .. n_run - 1)
  int n_run;
  const double* extra_plane;    // [n_run][4] unit normal + distance of one more cut (the stellar surface) or nullptr;
                                // a row with distance <= 0: none
  int max_neighbours;           // row length of neigh
  int* n_neigh;                 // [n_run] faces before the cuts; -1: incomplete with these k candidates; -2: overflow
  int* neigh;                   // [n_run][max_neighbours] in the order the faces appeared
  double* volume;               // [n_run] after the cuts
  double* volume_uncut;         // [n_run] or nullptr: before them (the volume a density estimate m / V of the particle wants)
  double* delta_edge;           // [n_run] farthest vertex before the cuts
  unsigned char* was_cut;       // [n_run]
};

struct TessCell {
  double pn[TESS_MAX_P][4];     // planes: unit normal, distance (n . x <= d)
  int pid[TESS_MAX_P];
  unsigned char tri[TESS_MAX_T][3];
  double vx[TESS_MAX_T][3];     // vertex positions (relative to the site)
  int np, nt;
  bool overflow;
};

// the point where three planes meet (Cramer)
__device__ inline void tess_vertex(const TessCell& C, int a, int b, int c, double* v) {
  const double* A = C.pn[a]; const double* B = C.pn[b]; const double* D = C.pn[c];
  const double bxd0 = B[1] * D[2] - B[2] * D[1], bxd1 = B[2] * D[0] - B[0] * D[2], bxd2 = B[0] * D[1] - B[1] * D[0];
  const double dxa0 = D[1] * A[2] - D[2] * A[1], dxa1 = D[2] * A[0] - D[0] * A[2], dxa2 = D[0] * A[1] - D[1] * A[0];
  const double axb0 = A[1] * B[2] - A[2] * B[1], axb1 = A[2] * B[0] - A[0] * B[2], axb2 = A[0] * B[1] - A[1] * B[0];
  const double det = A[0] * bxd0 + A[1] * bxd1 + A[2] * bxd2;
  const double inv = 1.0 / det;
  v[0] = (A[3] * bxd0 + B[3] * dxa0 + D[3] * axb0) * inv;
  v[1] = (A[3] * bxd1 + B[3] * dxa1 + D[3] * axb1) * inv;
  v[2] = (A[3] * bxd2 + B[3] * dxa2 + D[3] * axb2) * inv;
}

__device__ inline void tess_init_box(TessCell& C, const double* s, const double* lim) {
  // walls in the order of Voronoi.f90:1275-1280: -x, +x, -y, +y, -z, +z (ids -1 .. -6)
  for (int w = 0; w < 6; ++w) {
    const int ax = w >> 1;
    const double sg = (w & 1) ? 1.0 : -1.0;
    C.pn[w][0] = C.pn[w][1] = C.pn[w][2] = 0.0;
    C.pn[w][ax] = sg;
    C.pn[w][3] = (w & 1) ? (lim[w] - s[ax]) : (s[ax] - lim[w]);
    C.pid[w] = TESS_ID_WALL0 - w;
  }
  C.np = 6;
  C.nt = 0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int l = 0; l < 2; ++l) {
        const int t = C.nt++;
        C.tri[t][0] = (unsigned char)i; C.tri[t][1] = (unsigned char)(2 + j); C.tri[t][2] = (unsigned char)(4 + l);
        tess_vertex(C, i, 2 + j, 4 + l, C.vx[t]);
      }
  C.overflow = false;
}

__device__ inline bool tess_has(const unsigned char* t, int a, int b) {
  const bool ha = (t[0] == a) | (t[1] == a) | (t[2] == a);
  const bool hb = (t[0] == b) | (t[1] == b) | (t[2] == b);
  return ha & hb;
}

// clip the cell by n . x <= d; returns true when the plane became a face
__device__ inline bool tess_clip(TessCell& C, double nx, double ny, double nz, double d, int id) {
  unsigned int out[(TESS_MAX_T + 31) / 32];
  for (int w = 0; w < (TESS_MAX_T + 31) / 32; ++w) out[w] = 0u;
  int n_out = 0;
  double scale = fabs(d);
  for (int t = 0; t < C.nt; ++t) {
    const double r = fabs(C.vx[t][0]) + fabs(C.vx[t][1]) + fabs(C.vx[t][2]);
    scale = r > scale ? r : scale;
  }
  const double eps = 1.0e-12 * scale;
  for (int t = 0; t < C.nt; ++t) {
    const double sgn = nx * C.vx[t][0] + ny * C.vx[t][1] + nz * C.vx[t][2] - d;
    if (sgn > eps) { out[t >> 5] |= 1u << (t & 31); ++n_out; }
  }
  if (n_out == 0) return false;
  if (C.np >= TESS_MAX_P) { C.overflow = true; return false; }
  const int P = C.np++;
  C.pn[P][0] = nx; C.pn[P][1] = ny; C.pn[P][2] = nz; C.pn[P][3] = d;
  C.pid[P] = id;
  // every edge (pair of planes) with one vertex beyond the plane and one on this side gets a new vertex
  const int nt0 = C.nt;
  for (int t = 0; t < nt0; ++t) {
    if (!((out[t >> 5] >> (t & 31)) & 1u)) continue;
    for (int e = 0; e < 3; ++e) {
      const int a = C.tri[t][e], b = C.tri[t][(e + 1) % 3];
      // the other vertex of edge (a, b)
      int o = -1;
      for (int q = 0; q < nt0; ++q)
        if (q != t && tess_has(C.tri[q], a, b)) { o = q; break; }
      if (o < 0 || ((out[o >> 5] >> (o & 31)) & 1u)) continue;
      if (C.nt >= TESS_MAX_T) { C.overflow = true; continue; }
      const int v = C.nt++;
      C.tri[v][0] = (unsigned char)a; C.tri[v][1] = (unsigned char)b; C.tri[v][2] = (unsigned char)P;
      tess_vertex(C, a, b, P, C.vx[v]);
    }
  }
  // drop the vertices beyond the plane
  int w = 0;
  for (int t = 0; t < C.nt; ++t) {
    const bool gone = t < nt0 && ((out[t >> 5] >> (t & 31)) & 1u);
    if (gone) continue;
    if (w != t) {
      C.tri[w][0] = C.tri[t][0]; C.tri[w][1] = C.tri[t][1]; C.tri[w][2] = C.tri[t][2];
      C.vx[w][0] = C.vx[t][0]; C.vx[w][1] = C.vx[t][1]; C.vx[w][2] = C.vx[t][2];
    }
    ++w;
  }
  C.nt = w;
  return true;
}

__device__ inline double tess_rmax2(const TessCell& C) {
  double m = 0.0;
  for (int t = 0; t < C.nt; ++t) {
    const double r2 = C.vx[t][0] * C.vx[t][0] + C.vx[t][1] * C.vx[t][1] + C.vx[t][2] * C.vx[t][2];
    m = r2 > m ? r2 : m;
  }
  return m;
}

// volume = 1/3 sum over the faces of area x distance (the site is the origin, inside every half-space); a face's
// vertices are ordered by their angle around the face's centroid (a convex polygon)
__device__ inline double tess_volume(const TessCell& C) {
  double vol = 0.0;
  for (int f = 0; f < C.np; ++f) {
    int idx[32];
    double ang[32];
    int m = 0;
    double cx = 0.0, cy = 0.0, cz = 0.0;
    for (int t = 0; t < C.nt; ++t)
      if (C.tri[t][0] == f || C.tri[t][1] == f || C.tri[t][2] == f) {
        if (m < 32) idx[m] = t;
        ++m;
        cx += C.vx[t][0]; cy += C.vx[t][1]; cz += C.vx[t][2];
      }
    if (m < 3) continue;
    if (m > 32) m = 32;   // (a face with more than 32 corners does not occur; its area would be underestimated)
    cx /= m; cy /= m; cz /= m;
    const double* N = C.pn[f];
    // in-plane basis
    double e1x = C.vx[idx[0]][0] - cx, e1y = C.vx[idx[0]][1] - cy, e1z = C.vx[idx[0]][2] - cz;
    const double l1 = sqrt(e1x * e1x + e1y * e1y + e1z * e1z);
    if (!(l1 > 0.0)) continue;
    e1x /= l1; e1y /= l1; e1z /= l1;
    const double e2x = N[1] * e1z - N[2] * e1y, e2y = N[2] * e1x - N[0] * e1z, e2z = N[0] * e1y - N[1] * e1x;
    for (int i = 0; i < m; ++i) {
      const double dx = C.vx[idx[i]][0] - cx, dy = C.vx[idx[i]][1] - cy, dz = C.vx[idx[i]][2] - cz;
      ang[i] = atan2(dx * e2x + dy * e2y + dz * e2z, dx * e1x + dy * e1y + dz * e1z);
    }
    for (int i = 1; i < m; ++i) {   // insertion sort by angle
      const double a = ang[i];
      const int ii = idx[i];
      int j = i - 1;
      while (j >= 0 && ang[j] > a) { ang[j + 1] = ang[j]; idx[j + 1] = idx[j]; --j; }
      ang[j + 1] = a; idx[j + 1] = ii;
    }
    double ax = 0.0, ay = 0.0, az = 0.0;
    for (int i = 0; i < m; ++i) {
      const double* p = C.vx[idx[i]];
      const double* q = C.vx[idx[(i + 1) % m]];
      const double px = p[0] - cx, py = p[1] - cy, pz = p[2] - cz, qx = q[0] - cx, qy = q[1] - cy, qz = q[2] - cz;
      ax += py * qz - pz * qy; ay += pz * qx - px * qz; az += px * qy - py * qx;
    }
    const double area = 0.5 * fabs(ax * N[0] + ay * N[1] + az * N[2]);
    vol += area * N[3] * (1.0 / 3.0);
  }
  return vol;
}

__global__ void __launch_bounds__(64) k_voronoi_cells(const TessArgs A) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= A.n_run) return;
  const int ic = A.cells ? A.cells[r] : r;
  const double s[3] = {A.xyz[3 * (size_t)ic], A.xyz[3 * (size_t)ic + 1], A.xyz[3 * (size_t)ic + 2]};
  TessCell C;
  tess_init_box(C, s, A.limits);
  double rmax2 = tess_rmax2(C);
  const bool trusted = A.knn_first != nullptr;
  bool complete = trusted;
  const int* cand = trusted ? A.knn + A.knn_first[r] : A.knn + (size_t)r * A.k;
  const int n_cand = trusted ? A.knn_first[r + 1] - A.knn_first[r] : A.k;
  for (int q = 0; q < n_cand; ++q) {
    const int j = cand[q];
    if (j < 0) { complete = true; break; }   // (fewer sites than k: every one of them was a candidate)
    const double rx = A.xyz[3 * (size_t)j] - s[0], ry = A.xyz[3 * (size_t)j + 1] - s[1], rz = A.xyz[3 * (size_t)j + 2] - s[2];
    const double d2 = rx * rx + ry * ry + rz * rz;
    if (d2 > 4.0 * rmax2) { if (trusted) continue; complete = true; break; }   // security radius: no farther site can cut the cell
    const double dist = sqrt(d2), inv = 1.0 / dist;
    if (tess_clip(C, rx * inv, ry * inv, rz * inv, 0.5 * dist, j)) rmax2 = tess_rmax2(C);
  }
  if (C.overflow) { A.n_neigh[r] = -2; return; }
  if (!complete) { A.n_neigh[r] = -1; return; }
  // the faces, in the order they appeared (a plane whose vertices were all cut away again is no face)
  int nf = 0;
  for (int f = 0; f < C.np; ++f) {
    bool used = false;
    for (int t = 0; t < C.nt && !used; ++t) used = (C.tri[t][0] == f) | (C.tri[t][1] == f) | (C.tri[t][2] == f);
    if (!used) continue;
    if (nf < A.max_neighbours) A.neigh[(size_t)r * A.max_neighbours + nf] = C.pid[f];
    ++nf;
  }
  A.n_neigh[r] = nf <= A.max_neighbours ? nf : -2;
  const double delta_edge = sqrt(rmax2);
  A.delta_edge[r] = delta_edge;
  // elongated cells are intersected with the Platonic solid (voro++_wrapper.cpp:209-227); the neighbour list above is
  // the uncut cell's, like the reference's (its list is stored before the cut, :195-207)
  const double hc = A.h[ic];
  const bool any_cut = (delta_edge > A.threshold * hc) || (A.extra_plane && A.extra_plane[4 * (size_t)r + 3] > 0.0);
  double v0 = 0.0;
  if (A.volume_uncut && any_cut) { v0 = tess_volume(C); A.volume_uncut[r] = v0; }
  bool cut = false;
  if (delta_edge > A.threshold * hc) {
    const double cd = A.cutting_distance_o_h * hc;
    for (int v = 0; v < A.n_vectors; ++v) tess_clip(C, A.cut_vec[v][0], A.cut_vec[v][1], A.cut_vec[v][2], cd, TESS_ID_CUT);
    cut = true;
  }
  A.was_cut[r] = cut ? 1 : 0;
  if (A.extra_plane) {   // a star's neighbour closer than 2 R*: cut at the stellar surface (:241-262)
    const double* e = A.extra_plane + 4 * (size_t)r;
    if (e[3] > 0.0) tess_clip(C, e[0], e[1], e[2], e[3], TESS_ID_CUT);
  }
  const double v1 = C.overflow ? -1.0 : tess_volume(C);
  A.volume[r] = v1;
  if (A.volume_uncut && !any_cut) A.volume_uncut[r] = v1;
}

}  // namespace mcgpu
#endif
