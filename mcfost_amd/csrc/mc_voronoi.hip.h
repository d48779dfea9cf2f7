// Voronoi-grid backend of the thermal packet kernel (Voronoi.f90).
//
// Same state machine, random streams and shared emission/interaction code as the cylindrical
// kernel (mc_device.hip.h); only the grid operators differ:
//   cross_Voronoi_cell    Voronoi.f90:839-992   (distance_to_wall :1289, distance_to_star :1321)
//   test_exit_grid        :1446  (icell < 0)
//   move_to_grid_Voronoi  :1379  (+ find_Voronoi_cell over the wall's neighbour list, :1485)
//   index_cell_voronoi    :1548  (brute force; only the rounding-error recovery path uses it)
//   pos_em_cell_voronoi   :1510  (the cell centre)
//
// HBM layout (MI355X: memory is plentiful, scattered 12-byte gathers are not):
//   * VoroCell[n_cells], 32 B, one aligned load per crossing: the site in default real, the CSR
//     offset/count, the flags and the cell's opacity factor.
//   * the neighbour list is stored INLINED: one float4 per (cell, neighbour) = the neighbour
//     site's coordinates + its id in the 4th word, in the reference's list order.  A crossing
//     streams ~15 consecutive float4 of its own cell instead of chasing 15 random sites (the
//     idea of the reference's disabled `Voronoi_neighbour_xyz`, Voronoi.f90:18-20,886-887,
//     paid for with 16 B x nnz of HBM: 250 MB per million cells).
//   * the plane tests are done in default real exactly like the reference (:859), with the
//     products and sums left unfused so that the CPU oracle reproduces them bit for bit.
//   * absorbed energy: the grid (8 B x n_cells) does not fit a CU's LDS, but the deposits are
//     extremely concentrated (every packet starts in the handful of cells around the star), and
//     same-address global_atomic_add_f64 serialise in L2: measured 8x slower than no deposits at
//     all.  Each workgroup therefore keeps a hashed DEPOSIT CACHE in LDS (tag + FP64 sum per
//     slot, slots claimed first-come and then fixed for the launch): hits are ds_add_f64, misses
//     go straight to HBM, and the waves fold rotating slices of the cache into HBM without a
//     workgroup barrier, exactly like the 2D kernel's private grid.
#pragma once
#include "mc_device.hip.h"

namespace mcgpu {

struct VoroCell {
  float x, y, z;  // Voronoi_xyz(:,icell)
  int first;      // 0-based offset of the cell's entries in VoroGrid::nb
  int count;      // last_neighbour - first_neighbour + 1
  int flags;      // bit 0 was_cut, bit 1 is_star_neighbour
  double kf;      // kappa_factor(icell)
};
static_assert(sizeof(VoroCell) == 32, "VoroCell is one 32-byte record");

struct VoroNb {
  float x, y, z;  // Voronoi_xyz(:,id) (unused for walls)
  int id;         // neighbour cell (> 0) or -iwall
};
static_assert(sizeof(VoroNb) == 16, "VoroNb is one 16-byte record");

struct VoroGrid {
  int n_cells;
  const VoroCell* cell;
  const VoroNb* nb;
  const double* h;        // Voronoi(:)%h, read for cut cells only
  const double* xyz_dp;   // Voronoi(:)%xyz (3 per cell)
  const int* wall_first;  // [7]
  const int* wall_cells;  // wall(iwall)%neighbour_list, concatenated
  const unsigned char* nb_cls;  // per entry of nb: the neighbour cell's list-length class (vp_class_of, mc_voronoi_pool.hip.h); 0 for walls
  float walls[24];        // 6 x (x1,x2,x3,x4) (Voronoi.f90:1275-1280)
  double cut_o_h;         // PS%cutting_distance_o_h
};

// default-real dot product, evaluated left to right without contraction
__device__ inline float dot3f(float a0, float a1, float a2, float b0, float b1, float b2) {
  return nf_add(nf_add(nf_mul(a0, b0), nf_mul(a1, b1)), nf_mul(a2, b2));
}

// distance_to_wall (Voronoi.f90:1289-1317)
__device__ inline double voro_distance_to_wall(const VoroGrid& G, double x, double y, double z, double u,
                                               double v, double w, int iwall) {
  const float* W = G.walls + 4 * (iwall - 1);
  const double n0 = W[0], n1 = W[1], n2 = W[2];
  const double p0 = (double)W[3] * fabs(n0), p1 = (double)W[3] * fabs(n1), p2 = (double)W[3] * fabs(n2);
  const float den = (float)nd_add(nd_add(nd_mul(n0, u), nd_mul(n1, v)), nd_mul(n2, w));
  if (fabsf(den) > FLT_TINY) {
    const double num = nd_add(nd_add(nd_mul(n0, p0 - x), nd_mul(n1, p1 - y)), nd_mul(n2, p2 - z));
    return num / (double)den;
  }
  return (double)FLT_HUGE;
}

// distance_to_star (Voronoi.f90:1321-1375)
__device__ inline double voro_distance_to_star(const DevModel& M, double x, double y, double z, double u,
                                               double v, double w, int& i_star) {
  double d = HUGE_DP;
  i_star = 0;
  for (int i = 1; i <= M.n_stars; ++i) {
    const double* s4 = &M.star_xyzr[4 * (i - 1)];
    const double dx = x - s4[0], dy = y - s4[1], dz = z - s4[2];
    const double b = nd_add(nd_add(nd_mul(dx, u), nd_mul(dy, v)), nd_mul(dz, w));
    const double c = nd_add(nd_add(nd_add(nd_mul(dx, dx), nd_mul(dy, dy)), nd_mul(dz, dz)),
                               -nd_mul(s4[3], s4[3]));
    const double delta = nd_add(nd_mul(b, b), -c);
    if (delta >= 0.0) {
      const double rac = sqrt(delta), s1 = -b - rac;
      if (s1 < 0) {
        const double s2 = -b + rac;
        if (s2 > 0) { d = 0.0; i_star = i; }
      } else if (s1 < d) {
        d = s1; i_star = i;
      }
    }
  }
  return d;
}

// is_in_volume (Voronoi.f90:1462-1478)
__device__ inline bool voro_is_in_volume(const VoroGrid& G, double x, double y, double z) {
  return (x > (double)G.walls[3]) && (x < (double)G.walls[7]) && (y > (double)G.walls[11]) &&
         (y < (double)G.walls[15]) && (z > (double)G.walls[19]) && (z < (double)G.walls[23]);
}

__device__ inline float voro_dist2f(const double* c, double x, double y, double z) {
  const double dx = c[0] - x, dy = c[1] - y, dz = c[2] - z;
  return (float)nd_add(nd_add(nd_mul(dx, dx), nd_mul(dy, dy)), nd_mul(dz, dz));
}

// index_cell_voronoi (Voronoi.f90:1548-1570): brute force with default-real distances
__device__ inline int voro_index_cell(const VoroGrid& G, double x, double y, double z) {
  float dmin = FLT_HUGE;
  int ic = 0;
  for (int i = 1; i <= G.n_cells; ++i) {
    const float d2 = voro_dist2f(G.xyz_dp + 3 * (size_t)(i - 1), x, y, z);
    if (d2 < dmin) { ic = i; dmin = d2; }
  }
  return ic;
}

#ifdef MCGPU_VORO_DIAG  // diagnostic builds (tests/devtools/voro_diag.py): where a wave's instructions go
// VD(w, l): this statement is reached by some lanes of the wave: count the wave once (w) and the lanes (l)
struct VoroDiag { unsigned int c[10]; };
#define VD(D, w, l) do { const unsigned long long m__ = __ballot(1); \
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m__) - 1) (D).c[w]++; (D).c[l]++; } while (0)
#define VDARG , VoroDiag* VDp = nullptr
#define VDPASS , nullptr, &VDg
#else
#define VDARG
#define VDPASS
#endif

// cross_Voronoi_cell (Voronoi.f90:839-992).  C = the cell's record (loaded by the caller, who also
// needs its opacity factor).
// BATCH > 0 (the pool schedule): the records are requested BATCH at a time, all of a batch back to back before the first
// is looked at -- a list of up to BATCH neighbours costs ONE memory latency instead of one per group of four (measured:
// the scan was 46 % of the pool's wave time, waiting; tests/devtools/voro_pool_diag.py) -- at the price of 4 BATCH registers.
template <int BATCH = 0>
__device__ inline void voro_cross_cell(const VoroGrid& G, const DevModel& M, const VoroCell& C, double x,
                                       double y, double z, double u, double v, double w, int icell,
                                       int previous_cell, double& x1, double& y1, double& z1,
                                       int& next_cell, double& s_out, double& s_contrib,
                                       double& s_void_before, int* cls_out = nullptr VDARG) {
  const float r0 = (float)x, r1 = (float)y, r2 = (float)z;
  const float k0 = (float)u, k1 = (float)v, k2 = (float)w;
  // The reference keeps the smallest quotient s_tmp = num / den over the neighbours (:859-905), one FP64 division per
  // neighbour.  num and den are default reals, so the cross products num * den_best and num_best * den are exact in
  // FP64 (24 + 24 bits) and "num / den < num_best / den_best" can be decided without dividing: two different such
  // fractions differ by more than 2^-48 relatively, which the rounded quotients resolve as well, and equal fractions
  // are "not smaller" either way.  The running minimum is (s_num, s_den); ONE division at the end (or when a wall of
  // the box, whose distance is a genuine double, has to be compared).
  double s_num = 1.00000001504746621988e+30, s_den = 1.0;  // real 1e30
  next_cell = 0;
  const VoroNb* nb = G.nb + C.first;
  const int cnt = C.count;
  const int last = cnt > 0 ? cnt - 1 : 0;  // (where the fetches past the end of the list are pointed)
  // The walls of the box among the neighbours (ids -1 .. -6; their distance is a genuine double with two more
  // divisions) are only noted in the scan -- (position, wall) in 10 bits each, at most six -- and compared after it:
  // a wave runs this loop as long as its longest list, and the wall branch would be taken on almost every trip by
  // some lane.  The reference's sequential scan keeps the FIRST of equal minima, so the late comparison carries the
  // list positions: the result is the one of the scan in list order.
  unsigned long long walls = 0ull;
  int best_pos = 128;
  // The records are fetched four at a time -- one 64-byte line's worth, requested back to back so that the line is
  // fetched once -- and one group ahead: the scan is a chain of dependent gathers, and a wave that waits for each of
  // them in turn spends more time waiting than computing (wait_frac 0.59 with one record per trip).
#ifndef MCGPU_VORO_GROUP
#define MCGPU_VORO_GROUP 4
#endif
  constexpr int VG = MCGPU_VORO_GROUP;
#ifndef MCGPU_VORO_AHEAD
#define MCGPU_VORO_AHEAD 1
#endif
  // one neighbour of the list (position i): Voronoi.f90:879-917
  auto look_at = [&](const VoroNb& N, int i) {
    if (i >= cnt || N.id == previous_cell) return;
    if (N.id > 0) {
      const float n0 = nf_sub(N.x, C.x), n1 = nf_sub(N.y, C.y), n2 = nf_sub(N.z, C.z);
      const double den = (double)dot3f(n0, n1, n2, k0, k1, k2);
      if (den <= 0.0) return;
      const float p0 = nf_mul(0.5f, nf_add(N.x, C.x)), p1 = nf_mul(0.5f, nf_add(N.y, C.y)),
                  p2 = nf_mul(0.5f, nf_add(N.z, C.z));
      const double num = (double)dot3f(n0, n1, n2, nf_sub(p0, r0), nf_sub(p1, r1), nf_sub(p2, r2));
      // num < 0: the reference sets s_tmp = huge(1.0) > 1e30, never the minimum
      if (!(num < 0.0) && nd_mul(num, s_den) < nd_mul(s_num, den)) { s_num = num; s_den = den; next_cell = N.id; best_pos = i; }
    } else {
      walls = (walls << 10) | (unsigned long long)(((i < 127 ? i : 127) << 3) | (-N.id));
    }
  };
  if (BATCH > 0) {
    constexpr int VB = BATCH > 0 ? BATCH : 4;
    for (int i0 = 0; i0 < cnt; i0 += VB) {
      VoroNb Nb[VB];
#pragma unroll
      for (int j = 0; j < VB; ++j) Nb[j] = nb[(i0 + j < cnt) ? i0 + j : last];
#pragma unroll
      for (int j = 0; j < VB; ++j) {
        if ((j & 3) == 0 && j > 0 && __ballot(i0 + j < cnt) == 0ull) break;   // (no lane of the wave has a neighbour left in this batch)
        look_at(Nb[j], i0 + j);
      }
    }
  } else {
  VoroNb Nn[VG];
  if (MCGPU_VORO_AHEAD) {
#pragma unroll
    for (int j = 0; j < VG; ++j) Nn[j] = nb[j < cnt ? j : last];
  }
  for (int i0 = 0; i0 < cnt; i0 += VG) {
#if MCGPU_VORO_DIAG == 2
    if (VDp) VD(*VDp, 1, 2);
#endif
    VoroNb Nc[VG];
    if (MCGPU_VORO_AHEAD) {
#pragma unroll
      for (int j = 0; j < VG; ++j) Nc[j] = Nn[j];
      if (i0 + VG < cnt) {
#pragma unroll
        for (int j = 0; j < VG; ++j) Nn[j] = nb[(i0 + VG + j < cnt) ? i0 + VG + j : last];
      }
    } else {
#pragma unroll
      for (int j = 0; j < VG; ++j) Nc[j] = nb[(i0 + j < cnt) ? i0 + j : last];
    }
#pragma unroll
    for (int j = 0; j < VG; ++j) look_at(Nc[j], i0 + j);
  }
  }
  double s = s_num / s_den;
  while (__builtin_expect(walls != 0ull, 0)) {
#if MCGPU_VORO_DIAG == 2
    if (VDp) VD(*VDp, 3, 4);
#endif
    const int wid = (int)(walls & 7ull), pos = (int)((walls >> 3) & 127ull);
    walls >>= 10;
    double s_tmp = voro_distance_to_wall(G, x, y, z, u, v, w, wid);
    if (s_tmp < 0.0) s_tmp = (double)FLT_HUGE;
    if (s_tmp < s || (s_tmp == s && pos < best_pos)) { s = s_tmp; next_cell = -wid; best_pos = pos; }
  }
  s = nd_mul(s, 1.0 + (double)1e-5f);
  x1 = nd_add(x, nd_mul(u, s));
  y1 = nd_add(y, nd_mul(v, s));
  z1 = nd_add(z, nd_mul(w, s));
  if (next_cell == 0) {  // rounding error somewhere (:926-937)
#if MCGPU_VORO_DIAG == 2
    if (VDp) VDp->c[9]++;
#endif
    best_pos = -1;
    x1 = x; y1 = y; z1 = z; s = 0.0;
    if (voro_is_in_volume(G, x, y, z)) {
      next_cell = voro_index_cell(G, x, y, z);
      if (icell == next_cell) next_cell = -1;
    } else {
      next_cell = -1;
    }
  }
  // (the pool schedule, mc_voronoi_pool.hip.h: the list-length class of the cell entered, a byte per neighbour; read here,
  // ahead of the cut-cell and star code, so that its latency is theirs; -1: the next cell did not come out of the scan)
  int cls_next = -1;
  if (cls_out && next_cell > 0 && best_pos >= 0 && best_pos < cnt) cls_next = (int)G.nb_cls[C.first + best_pos];
  if (C.flags & 1) {  // cut cell: only the sphere of radius h*cutting_distance_o_h holds matter (:939-975)
#if MCGPU_VORO_DIAG == 2
    if (VDp) VD(*VDp, 5, 6);
#endif
    const double d0 = (double)nf_sub(r0, C.x), d1 = (double)nf_sub(r1, C.y), d2 = (double)nf_sub(r2, C.z);
    const double b = nd_add(nd_add(nd_mul(d0, (double)k0), nd_mul(d1, (double)k1)), nd_mul(d2, (double)k2));
    const double hc = nd_mul(G.h[icell - 1], G.cut_o_h);
    const double c = nd_add(nd_add(nd_add(nd_mul(d0, d0), nd_mul(d1, d1)), nd_mul(d2, d2)),
                               -nd_mul(hc, hc));
    const double delta = nd_add(nd_mul(b, b), -c);
    if (delta < 0.0) {
      s_void_before = s; s_contrib = 0.0;
    } else {
      const double rac = sqrt(delta), s1 = -b - rac, s2 = -b + rac;
      if (s1 < 0) {
        if (s2 < 0) { s_void_before = s; s_contrib = 0.0; }
        else { s_void_before = 0.0; s_contrib = fmin(s2, s); }
      } else if (s1 < s) {
        s_void_before = s1; s_contrib = fmin(s2, s) - s1;
      } else {
        s_void_before = s; s_contrib = 0.0;
      }
    }
  } else {
    s_void_before = 0.0; s_contrib = s;
  }
  if (C.flags & 2) {  // star neighbour (:977-988)
#if MCGPU_VORO_DIAG == 2
    if (VDp) VD(*VDp, 7, 8);
#endif
    int i_star;
    const double d_to_star = voro_distance_to_star(M, x, y, z, u, v, w, i_star);
    if (i_star > 0 && d_to_star < s) {
      s_contrib = d_to_star;
      next_cell = M.star_cell[4 * (i_star - 1)];
      cls_next = -1;
    }
  }
  s_out = s;
  if (cls_out) *cls_out = cls_next;
}

// move_to_grid_Voronoi (Voronoi.f90:1379-1442) + find_Voronoi_cell (:1625; the kd-tree's answer by direct search)
__device__ inline bool voro_move_to_grid(const VoroGrid& G, double& x, double& y, double& z, double u,
                                         double v, double w, int& icell) {
  double s_walls[6];
  int order[6];
  for (int iw = 1; iw <= 6; ++iw) {
    const double l = voro_distance_to_wall(G, x, y, z, u, v, w, iw);
    s_walls[iw - 1] = (l >= 0) ? nd_mul(l, 1.0 + 1.e-6) : (double)FLT_HUGE;
    order[iw - 1] = iw;
  }
  for (int a = 1; a < 6; ++a)
    for (int b = a; b > 0 && s_walls[order[b] - 1] < s_walls[order[b - 1] - 1]; --b) {
      const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t;
    }
  int iwall = 0;
  double xt = 0, yt = 0, zt = 0;
  bool found = false;
  for (int i = 0; i < 6 && !found; ++i) {
    iwall = order[i];
    const double l = s_walls[iwall - 1];
    xt = nd_add(x, nd_mul(l, u)); yt = nd_add(y, nd_mul(l, v)); zt = nd_add(z, nd_mul(l, w));
    found = voro_is_in_volume(G, xt, yt, zt);
  }
  if (!found) { icell = 0; return false; }
  x = xt; y = yt; z = zt;
  // find_Voronoi_cell (:1625-1645): kdtree2_n_nearest, i.e. the nearest site of the wall's list in kdkind = dp
  double dmin = HUGE_DP;
  int imin = 0;
  for (int q = G.wall_first[iwall - 1]; q < G.wall_first[iwall]; ++q) {
    const int ic = G.wall_cells[q];
    const double* c = G.xyz_dp + 3 * (size_t)(ic - 1);
    const double dx = c[0] - xt, dy = c[1] - yt, dz = c[2] - zt;
    const double d2 = nd_add(nd_add(nd_mul(dx, dx), nd_mul(dy, dy)), nd_mul(dz, dz));
    if (d2 < dmin) { imin = ic; dmin = d2; }
  }
  icell = imin;
  return true;
}

// the Voronoi grid's operators for emit_packet (mc_device.hip.h); the packet's cell is `icell`
struct VoroEmitOps {
  const VoroGrid& G;
  const DevModel& M;
  int& icell;
  __device__ inline void star_cell(int i_star, double, double, double) { icell = M.star_cell[4 * (i_star - 1)]; }  // stars.f90:155-156
  __device__ inline bool enter_grid(double& x, double& y, double& z, double u, double v, double w) {
    return voro_move_to_grid(G, x, y, z, u, v, w, icell);
  }
  __device__ inline void disk_cell(int ic, float, float, float, double& x, double& y, double& z) {
    icell = ic;
    const double* c = G.xyz_dp + 3 * (size_t)(ic - 1);  // pos_em_cell_voronoi (Voronoi.f90:1510-1542): the cell centre
    x = c[0]; y = c[1]; z = c[2];
  }
};

// distance_to_closest_wall_Voronoi (Voronoi.f90:996-1061) in its working form: the perpendicular distance from the
// point to the closest face, n . (p - r) / |n| with n the vector to the neighbour's site and p the midpoint -- the
// reference divides by n . n, which is that distance in units of the neighbour separation.  As there: 0 (no walk) in a
// cut cell and in a cell that touches the box.
__device__ inline double voro_distance_to_closest_wall(const VoroGrid& G, const VoroCell& C, double x, double y, double z) {
  if (C.flags & 1) return 0.0;
  const VoroNb* nb = G.nb + C.first;
  double s = 1.0e30;
  for (int i = 0; i < C.count; ++i) {
    const VoroNb N = nb[i];
    if (N.id <= 0) return 0.0;
    const double n0 = (double)N.x - (double)C.x, n1 = (double)N.y - (double)C.y, n2 = (double)N.z - (double)C.z;
    const double p0 = 0.5 * ((double)N.x + (double)C.x), p1 = 0.5 * ((double)N.y + (double)C.y), p2 = 0.5 * ((double)N.z + (double)C.z);
    double d = (n0 * (p0 - x) + n1 * (p1 - y) + n2 * (p2 - z)) / sqrt(n0 * n0 + n1 * n1 + n2 * n2);
    if (d < 0.0) d = 0.0;  // (a point that rounding has put beyond a face: no walk)
    s = fmin(s, d);
  }
  return s;
}

constexpr int VORO_CACHE_BLOCK = 1024;  // most threads of a cached-deposit workgroup (one per CU)

// ---------------------------------------------------------------------------
// The thermal packet kernel on a Voronoi grid.  CACHE: deposits go through the workgroup's LDS
// deposit cache; otherwise straight to HBM (global_atomic_add_f64).
// ---------------------------------------------------------------------------
// MRW: the modified random walk (mc_device.hip.h) with the Voronoi cell's closest face.
// VAR: lvariable_dust -- the tables of the cell's class (DevModel::cell_class), as thermal_body has it on cylindrical grids
template <bool POLA, bool CACHE, bool MRW = false, bool VAR = false>
__device__ __forceinline__ void thermal_body_voro(const DevModel& M, const RunArgs& A, const VoroGrid& G,
                                                  double* lds_base, int cache_log_ns) {
  const Lds T = lds_carve(lds_base, M);
  lds_stage(T, M);
  DepCache DC;
  DC.log_ns = cache_log_ns;
  DC.val = lds_base + (lds_bytes(M) + sizeof(double) - 1) / sizeof(double);
  DC.tag = reinterpret_cast<int*>(DC.val + ((size_t)1 << cache_log_ns));
  if (CACHE)
    for (int i = threadIdx.x; i < (1 << cache_log_ns); i += blockDim.x) { DC.val[i] = 0.0; DC.tag[i] = 0; }
  __syncthreads();
  const int lane = threadIdx.x & 63;

  int st = S_EMIT;
  double x = 0, y = 0, z = 0, u = 0, v = 0, w = 1, extr = 0;
  int icell = 0, prev_cell = 0, lambda = 1;
  int star_icell = 0;  // cell of the star this flight would hit (0: none)
  bool flag_star = false, flag_scatt = false, flag_ism = false;
  double S[4] = {1.0, 0.0, 0.0, 0.0};
  Rng rng;
  rng.init(0, 0);
  unsigned int c_cross = 0, c_flight = 0, c_scatt = 0, c_abs = 0, c_esc = 0, c_kill = 0, c_pack = 0;
  unsigned int pk_cross = 0;
  unsigned long long pk_next = 0, pk_end = 0;
  float tau_rand = 0.0f;
  int n_inter = 0;            // MRW: interactions in a row whose flights never left the cell (0..7)
  bool first_cross = true;    // MRW: the flight is still in the cell it started in
  unsigned int c_walks = 0, c_steps = 0;
#ifdef MCGPU_VORO_DIAG
  VoroDiag VDg;
  for (int q = 0; q < 10; ++q) VDg.c[q] = 0u;
#endif

  for (int ep = 0;; ++ep) {
#if MCGPU_VORO_DIAG == 1
    if (st != S_DONE) VD(VDg, 1, 2);   // outer rounds, lanes that own a packet
#endif
    if (st == S_EXITED) {
#if MCGPU_VORO_DIAG == 1
      VD(VDg, 9, 9);
#endif
      if (!flag_ism) {
        capteur<POLA>(M, A.sed, lambda, u, v, w, S, flag_star, flag_scatt);
        c_esc++;
      }
      st = S_EMIT;
    }
    {  // EMIT: packet ids from this wave's reserved batch (see thermal_body)
      const bool need = (st == S_EMIT);
      const unsigned long long mask = __ballot(need);
      if (mask) {
        if (pk_next >= pk_end) {
          const int leader = __ffsll((long long)mask) - 1;
          unsigned long long base = 0;
          if (lane == leader) base = atomicAdd(A.next_packet, (unsigned long long)PK_BATCH);
          base = __shfl(base, leader);
          pk_next = base < A.n_packets ? base : A.n_packets;
          pk_end = (base + PK_BATCH < A.n_packets) ? base + PK_BATCH : A.n_packets;
          if (pk_end < pk_next) pk_end = pk_next;
        }
        const unsigned long long avail = pk_end - pk_next;
        const unsigned long long rank = (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
        const unsigned long long cnt = (unsigned long long)__popcll(mask);
        const unsigned long long my = pk_next + rank;
        const bool served = need && (rank < avail);
        if (need && !served && pk_next >= A.n_packets) st = S_DONE;
        pk_next += (cnt < avail) ? cnt : avail;
        if (served) {
#if MCGPU_VORO_DIAG == 1
          VD(VDg, 7, 8);
#endif
          // mc_photon_loop body (dust_transfer.f90:529-541)
          rng.init(A.seed, A.first_packet + my);
          c_pack++;
          pk_cross = 0;
          float f[12];
          rng.emission_event(f);
          tau_rand = f[8];
          lambda = select_wl_em(T, M, f[0]);
          lds_count_sent(T, lambda);
          bool lintersect;
          flag_scatt = false;
          S[0] = 1.0; S[1] = 0.0; S[2] = 0.0; S[3] = 0.0;
          VoroEmitOps ops{G, M, icell};
          const int rc = emit_packet(M, f, lambda, T.fstar[lambda - 1], M.frac_E_disk[lambda - 1],
                                     M.prob_E_cell ? M.prob_E_cell + (size_t)(M.n_cells + 1) * (lambda - 1) : nullptr,
                                     ops, x, y, z, u, v, w, flag_star, flag_ism, lintersect);
          if (rc) {
            *A.err = rc;
            st = S_DONE;
          }
          if (st != S_DONE) st = lintersect ? S_NEWFLIGHT : S_EXITED;
        }
      }
    }

    if (st == S_INTERACT) {  // dust_transfer.f90:1260-1402
#if MCGPU_VORO_DIAG == 1
      VD(VDg, 3, 4);
#endif
      float g[8];
      rng.interaction_event(g, M.m1 != 0);
      tau_rand = g[5];
      double u1, v1, w1;
      const int ic = icell - 1;
      const int cls = VAR ? M.cell_class[ic] : -1;
      const Lds Tc = VAR ? class_tables(T, M, cls) : T;   // (lvariable_dust: this cell's tables)
      interact<POLA>(Tc, M, g, lambda, u, v, w, u1, v1, w1, S, flag_star, flag_scatt, c_scatt, c_abs, [&]() {
        if (A.frozen) return A.E_prior[ic];
        // what every workgroup has put into HBM so far + this workgroup's pending part standing
        // in for the others' (the reference's partial * nb_proc, thermal_emission.f90:670)
        double E = __hip_atomic_load(&A.E_abs[ic], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (CACHE) E += DC.pending(ic + 1) * (double)gridDim.x;
        return E * A.qscale;
      }, M.volume + ic, false, nullptr, -1, (VAR && M.v_scatt) ? cls : -1, (VAR && M.m1) ? cls : -1);
      if (!flag_scatt) flag_ism = false;
      u = u1; v = v1; w = w1;
      if (MRW) {  // dust_transfer.f90:1222-1239: a packet its cell has just re-emitted for the (n_inter + 1)-th time in a row
        if (__builtin_expect(!flag_scatt && !flag_star && n_inter > M.mrw_n_inter, 0)) {
          const VoroCell C = G.cell[ic];
          mrw_walk_with(Tc, M, rng.k0, rng.k1, rng.p_lo, rng.p_hi, rng.event, ic, C.kf, S[0], x, y, z, u, v, w, lambda,
                        [&](double px, double py, double pz) { return voro_distance_to_closest_wall(G, C, px, py, pz); },
                        [&]() {
                          if (A.frozen) return A.E_prior[ic];
                          double E = __hip_atomic_load(&A.E_abs[ic], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                          if (CACHE) E += DC.pending(ic + 1) * (double)gridDim.x;
                          return E * A.qscale;
                        },
                        [&](double e) { if (!(CACHE && DC.add(ic + 1, e))) atomic_add_f64(&A.E_abs[ic], e); }, c_walks, c_steps);
        }
      }
      st = S_NEWFLIGHT;
    }

    if (st == S_NEWFLIGHT) {
#if MCGPU_VORO_DIAG == 3
      VD(VDg, 7, 8);
#endif
      const float rand = tau_rand;  // dust_transfer.f90:1208-1215
      extr = tau_of_draw(rand);
      const int i_star = intersect_stars(M, x, y, z, u, v, w);  // optical_depth.f90:68
      star_icell = (i_star > 0) ? M.star_cell[4 * (i_star - 1)] : 0;
      c_flight++;
      prev_cell = 0;
      first_cross = true;
      st = S_FLIGHT;
    }

    if (__ballot(st != S_DONE) == 0ull) break;

    // FLIGHT: cell crossings (physical_length, optical_depth.f90:77-178)
#pragma unroll 1
    for (int it = 0; it < A.inner_iters; ++it) {
      if (A.min_active > 0 && it > 0) {
        const int flying = __popcll(__ballot(st == S_FLIGHT)), alive = __popcll(__ballot(st != S_DONE));
        if (flying * 64 < A.min_active * alive) break;
      }
      if (st == S_FLIGHT) {
#if MCGPU_VORO_DIAG == 1
        VD(VDg, 5, 6);
#endif
        if (icell < 0) {  // test_exit_grid_Voronoi (:1446)
          st = S_EXITED;
        } else if (star_icell > 0 && icell == star_icell) {  // optical_depth.f90:91-97
          c_kill++;
          st = S_EMIT;
        } else {
          const VoroCell C = G.cell[icell - 1];
          double opacity, kabs_c;
          if (VAR) {
            const size_t row = (size_t)M.cell_class[icell - 1] * M.n_lambda + (lambda - 1);
            opacity = M.v_kappa[row] * C.kf;
            kabs_c = M.v_kabs[row];
          } else {
            opacity = T.kappa[lambda - 1] * C.kf;
            kabs_c = T.kabs[lambda - 1];
          }
          double x1, y1, z1, l, l_contrib, l_void;
          int next;
          voro_cross_cell(G, M, C, x, y, z, u, v, w, icell, prev_cell, x1, y1, z1, next, l, l_contrib, l_void VDPASS);
          c_cross++;
          const double tau = l_contrib * opacity;
          if (tau > extr) {
#if MCGPU_VORO_DIAG == 3
            VD(VDg, 1, 2);
#endif
            const double lc = l_contrib * (extr / tau);
            const double ls = l_void + lc;
            const double dE = kabs_c * lc * S[0];
            if (dE != 0.0 && !MCGPU_DIAG(A.flags, 1)) {
              if (!(CACHE && DC.add(icell, dE))) atomic_add_f64(&A.E_abs[icell - 1], dE);
            }
            radiation_field_extras(M, A, icell - 1, lambda, lc * S[0]);
            x = nd_add(x, nd_mul(ls, u));
            y = nd_add(y, nd_mul(ls, v));
            z = nd_add(z, nd_mul(ls, w));
            st = S_INTERACT;
            if (MRW) n_inter = first_cross ? (n_inter < 7 ? n_inter + 1 : 7) : 0;  // dust_transfer.f90:1244-1249
          } else {
#if MCGPU_VORO_DIAG == 3
            VD(VDg, 3, 4);
#endif
            first_cross = false;
            extr = extr - tau;
            const double dE = kabs_c * l_contrib * S[0];
            if (dE != 0.0 && !MCGPU_DIAG(A.flags, 1)) {
              if (!(CACHE && DC.add(icell, dE))) atomic_add_f64(&A.E_abs[icell - 1], dE);
            }
            radiation_field_extras(M, A, icell - 1, lambda, l_contrib * S[0]);
            x = x1; y = y1; z = z1;
            prev_cell = icell;
            icell = next;
          }
          if (++pk_cross > 200000000u) {  // a packet that never leaves: flag it, drop it
            *A.err = 13;
            st = S_EMIT;
          }
        }
      }
    }
    // barrier-free partial fold of the deposit cache (see thermal_body): a slot's owner never
    // changes, so swapping its sum to zero and adding it to the owner's HBM cell is race-free
    if (CACHE && ((ep + 1) % A.flush_every) == 0) {
      const int n_waves = (blockDim.x + 63) >> 6, wave = threadIdx.x >> 6;
      const int slice = (wave + (ep + 1) / A.flush_every) % n_waves;
      const int ns = 1 << cache_log_ns, per = (ns + n_waves - 1) / n_waves;
      const int i0 = slice * per, i1 = (i0 + per < ns) ? i0 + per : ns;
      for (int i = i0 + lane; i < i1; i += 64) {
        const int t = DC.tag[i];
        if (t == 0) continue;
        const unsigned long long bits = atomicExch(reinterpret_cast<unsigned long long*>(&DC.val[i]), 0ull);
        const double e = __longlong_as_double((long long)bits);
        if (e != 0.0) atomic_add_f64(&A.E_abs[t - 1], e);
      }
    }
  }

  __syncthreads();  // every wave of the workgroup is done emitting and depositing
  lds_flush_sent(T, M, A.n_sent);
  if (CACHE) {  // final fold
    for (int i = threadIdx.x; i < (1 << cache_log_ns); i += blockDim.x) {
      const double e = DC.val[i];
      if (e != 0.0) atomic_add_f64(&A.E_abs[DC.tag[i] - 1], e);
    }
  }

  unsigned int cs[8] = {c_pack, c_cross, c_flight, c_scatt, c_abs, c_esc, c_kill, 0u};
#ifdef MCGPU_VORO_DIAG  // counters 1..9 carry the statistics of this diagnostic build instead
  for (int q = 1; q < 8; ++q) cs[q] = VDg.c[q];
  c_walks = VDg.c[8]; c_steps = VDg.c[9];
#endif
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    unsigned long long vsum = cs[q];
    for (int off = 32; off > 0; off >>= 1) vsum += __shfl_down(vsum, off);
    if (lane == 0 && vsum) atomicAdd(&A.counters[q], vsum);
  }
#ifdef MCGPU_VORO_DIAG
  if (true) {
#else
  if (MRW) {
#endif
    unsigned long long v8 = c_walks, v9 = c_steps;
    for (int off = 32; off > 0; off >>= 1) { v8 += __shfl_down(v8, off); v9 += __shfl_down(v9, off); }
    if (lane == 0 && v8) atomicAdd(&A.counters[8], v8);
    if (lane == 0 && v9) atomicAdd(&A.counters[9], v9);
  }
}

template <bool POLA>
__global__ void __launch_bounds__(256) k_thermal_voro(const DevModel M, const RunArgs A, const VoroGrid G) {
  extern __shared__ double lds_raw[];
  thermal_body_voro<POLA, false>(M, A, G, lds_raw, 0);
}

// ... with the modified random walk (HBM deposits)
template <bool POLA>
__global__ void __launch_bounds__(256) k_thermal_voro_mrw(const DevModel M, const RunArgs A, const VoroGrid G) {
  extern __shared__ double lds_raw[];
  thermal_body_voro<POLA, false, true>(M, A, G, lds_raw, 0);
}

// lvariable_dust (the deposit cache, 768 threads; MRW: with the random walk)
template <bool POLA, bool MRW>
__global__ void __launch_bounds__(768) k_thermal_voro_var(const DevModel M, const RunArgs A, const VoroGrid G, int cache_log_ns) {
  extern __shared__ double lds_raw[];
  thermal_body_voro<POLA, true, MRW, true>(M, A, G, lds_raw, cache_log_ns);
}

// BLOCK = 1024: 4 waves/SIMD at <= 128 VGPRs (spills); 768 (default): 3 waves/SIMD at <= 168, no scratch; 512: 2 waves/SIMD
template <bool POLA, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_thermal_voro_cache(const DevModel M, const RunArgs A,
                                                              const VoroGrid G, int cache_log_ns) {
  extern __shared__ double lds_raw[];
  thermal_body_voro<POLA, true>(M, A, G, lds_raw, cache_log_ns);
}

// probe: one cross_Voronoi_cell per thread (tests)
static __global__ void k_probe_cross_voro(const DevModel M, const VoroGrid G, int n, const double* x0, const double* y0,
                                   const double* z0, const double* u, const double* v, const double* w,
                                   const int* cell, const int* prev, double* x1, double* y1, double* z1,
                                   int* next, double* l, double* l_contrib, double* l_void) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const VoroCell C = G.cell[cell[i] - 1];
  voro_cross_cell(G, M, C, x0[i], y0[i], z0[i], u[i], v[i], w[i], cell[i], prev[i], x1[i], y1[i], z1[i], next[i],
                  l[i], l_contrib[i], l_void[i]);
}

}  // namespace mcgpu
