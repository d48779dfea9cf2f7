// mcgpu.hip -- host side of libmcfost_hip.so: the C-ABI declared in
// include/mcgpu.h.  Owns the HBM copies of the model tables, launches the
// persistent packet kernel and hands the accumulators back.  No CPU fallback:
// without a HIP device every call fails.
#include "../../include/mcgpu.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "mc_device.hip.h"
#include "mc_voronoi.hip.h"
#include "mc_voronoi_pool.hip.h"
#include "mc_mono.hip.h"
#include "mc_mono_voronoi.hip.h"
#include "mc_raytrace.hip.h"
#include "mc_raytrace_voronoi.hip.h"
#include "mc_roles.hip.h"
#include "mc_binned.hip.h"
#include "mc_tail.hip.h"
#include "mc_opacity.hip.h"
#include "mc_rt2.hip.h"
#include "mc_kernels.h"
#include <algorithm>
#include <chrono>
#include "host_tail.h"
#include "mc_xilog.hip.h"

using namespace mcgpu;

// Tuning knobs.  The shipped library takes its configuration from the context alone (mcgpu_set_option);
// a -DMCGPU_TUNING build additionally reads MCGPU_<NAME> from the environment (tools/*.py sweeps).
static int tune(const char* name, int dflt, int lo, int hi) {
#ifdef MCGPU_TUNING
  if (const char* e = getenv(name)) { const int v = atoi(e); if (v >= lo && v <= hi) return v; }
#else
  (void)name; (void)lo; (void)hi;
#endif
  return dflt;
}

// rotation (scattering.f90:553-590): host copy for the direction tables of mcgpu_rt2_source
static void host_rotation(double xinit, double yinit, double zinit, double u1, double v1, double w1, double& xfin, double& yfin,
                          double& zfin) {
  double cost, sint, sing;
  if (w1 > 0.999999999) {
    cost = 1.0; sint = 0.0; sing = 0.0;
  } else if (std::fabs(u1) < (double)FLT_MIN) {
    cost = 0.0; sint = 1.0;
    sing = std::sqrt(1.0 - w1 * w1);
  } else {
    const double theta = std::atan2(v1, u1);
    cost = std::cos(theta);
    sint = std::sin(theta);
    sing = std::sqrt(1.0 - w1 * w1);
  }
  const double prod = cost * xinit + sint * yinit;
  xfin = sing * prod + w1 * zinit;
  yfin = cost * yinit - sint * xinit;
  zfin = sing * zinit - w1 * prod;
}

constexpr int WORK_SLOT = 12;  // where the kernels' work counter lives in d_counters
constexpr int CNT_SLOTS = 24;  // d_counters: the counters, the longest packet's events (10), the work counter, the longest packet's own counts (16..20, TAIL_LONGEST)
static_assert(MCGPU_N_COUNTERS <= WORK_SLOT, "counter buffer layout");
static_assert(MCGPU_N_COUNTERS == TAIL_N_COUNTERS, "mc_tail.hip.h counts the same events");

struct mcgpu_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_tail = nullptr;     // recorded in front of k_tail: the launch's tail is [ev_tail, ev1]
  bool tail_launched = false;
  hipDeviceProp_t prop;
  std::string err;
  DevModel M;
  bool have_grid = false, have_stars = false, have_opacity = false, have_scatt = false,
       have_thermal = false, have_sed = false;
  bool reemission_pending = false;  // set_thermal / set_variable_dust left the LTE tables to mcgpu_init_reemission
  bool pending_single = false, pending_classes = false;  // ... which of the two sets
  int lsepar_pola = 0;
  int mrw_classes = 0;              // the number of classes the random walk's tables were set for
  double2* d_vkk = nullptr;         // the variable-dust role kernel's per-cell opacity pairs (built at the first launch)
  bool vkk_valid = false;
  std::vector<void*> opacity_allocs;  // the per-class tables mcgpu_opacity built (freed by the next call)
  float T_min = 1.0f;
  // mcgpu_set_option
  int opt_deposit = 0;      // 0 = automatic, 1 = HBM atomics, 2 = LDS-private grid / deposit cache, 3 = binned deposits
  int opt_log_mb = 0;       // binned deposits: size of the log in MiB (0 = automatic)
  int opt_tail = -1;        // role kernels hand their last packets to the tail kernel (mc_tail.hip.h) once a workgroup has this
                            // many left; 0: never; -1 (default): automatic -- 48 where packets get trapped (see tail_threshold())
  std::vector<double> h_r_lim;     // host copy of r_lim (cylindrical grids): the optical-thickness estimate below
  double tau_midplane = -1.0;      // radial optical depth of the midplane at the most opaque wavelength (-1: unknown)
  double last_inter_pp = -1.0;     // interactions per packet of the context's last completed thermal launch (-1: none yet)
  unsigned int* d_tail_next = nullptr;  // the tail kernel's work counter
  // the tail's last packets on the host (mc_tail.hip.h "The last packets on the host", host_tail.cpp)
  int opt_tail_where = 0;        // 0 = automatic (the host), 1 = k_tail finishes every packet, 2 = the host finishes the last ones
  int opt_host_threads = 0;      // host threads of a tail (0: the machine's, shared among its GPUs; at most 32)
  int opt_tail_host_max = 0;     // packets k_tail leaves to the host (0: 8 per host thread)
  unsigned int* d_tail_ctl = nullptr;   // [0] packets k_tail has finished, [1] records it has written to d_tail_out
  void* d_tail_out = nullptr;           // [tail_out_cap] records for the host
  unsigned int tail_out_cap = 0;
  char* h_arena = nullptr;              // pinned: the launch's table copies, accumulators, counters and records
  size_t h_arena_bytes = 0;
  hipStream_t side_stream = nullptr;    // copies the tables while k_tail runs
  hipEvent_t ev_side_in = nullptr, ev_side_out = nullptr;
  bool tail_on_host = false;            // the last thermal launch handed its last packets to the host
  // what the last host tail did (written by its callback; read after a synchronisation)
  double host_tail_ms = 0.0;
  unsigned int host_tail_packets = 0;
  int host_tail_threads = 0;
  unsigned long long host_tail_events = 0;
  // binned deposits (mc_binned.hip.h): the log and its plan
  BinLog bin{};
  unsigned int *d_bin_off = nullptr, *d_bin_cap = nullptr;
  double* d_bin_want = nullptr;  // [n_buckets] scratch of k_plan_bins
  unsigned long long bin_total_blocks = 0;
  bool bin_log_capped = false;      // the log was cut to a share of the free memory (asking again would not get more)
  int bin_max_parts = 0;
  double bin_dep_per_packet = 0.0;  // deposits per packet of the last run (0: not yet known)
  int bin_chunks = 0;               // chunks of the last launch
  double accum_packets = 0.0;       // packets whose deposits the accumulators hold (across accumulate-launches)
  // packets a chunk leaves unfinished (mc_roles.hip.h, "Chunks without tails"): two record buffers used in turns
  void* d_carry[2] = {nullptr, nullptr};
  unsigned int* d_carry_n = nullptr;  // [2]
  size_t carry_cap = 0;               // records per buffer
  int opt_schedule = 0;     // 0 = automatic (waves with roles where the queues fit), 1 = single-role kernel
  int opt_speculation = 1;  // SED mode: commit most of every stream before the scout pass
  int opt_cache_log_slots = 13;  // Voronoi deposit cache: 2^13 slots = 96 KB of LDS
  int opt_crossing = 0;          // 1: the flight-parametric 2D crossing in the role kernel's flying waves (statistical parity only)
  int opt_pool_log_rec = 12;     // Voronoi pool schedule: 2^12 packet records per workgroup (mc_voronoi_pool.hip.h)
  int opt_radiation_field = 0;   // bit 0: xN_abs, bit 1: xJ_abs (thermal step; radiation_field.f90:54-55)
  unsigned long long* d_xN = nullptr;  // [n_cells] (64-bit: a hot cell passes 2^32 segments within one 1e9-packet run)
  double* d_xJ = nullptr;        // (n_cells, n_lambda)
  std::vector<void*> allocs;   // every table buffer (freed in destroy)
  // per-setter buffers that may be replaced
  int *d_cmi = nullptr, *d_cmj = nullptr, *d_cmk = nullptr;
  float* d_tab_Temp = nullptr;
  // accumulators: [E_abs | sed | n_sent | counters as doubles (mcgpu_counters_to_accum)]
  double* d_accum = nullptr;
  size_t n_accum = 0;
  unsigned long long* d_counters = nullptr;  // [CNT_SLOTS]: MCGPU_N_COUNTERS counters, pad, the work counter at WORK_SLOT
  int* d_err = nullptr;
  double* d_E_prior = nullptr;
  bool launched = false;
  // SED mode (mc_mono.hip.h)
  bool have_rt1 = false;
  int RT_n_incl = 0, RT_n_az = 0, n_az_rt = 0, n_theta_rt = 0, N_type_flux = 0, lsepar_contrib = 0, n_lambda_pos = 0;
  const double *d_rt_u = nullptr, *d_rt_v = nullptr, *d_rt_w = nullptr;
  const float* d_tab_s11 = nullptr;
  double* d_xI = nullptr;
  bool have_rt2 = false;            // ray tracing method 2 (mcgpu_set_rt2): I_spec, I_spec_star
  int n_theta_I = 0, n_phi_I = 0, rt2_N_type_flux = 0, rt2_contrib = 0;
  double *d_I_spec = nullptr, *d_I_spec_star = nullptr;
  // the source function of the last mcgpu_rt2_source, resident for mcgpu_rt2_dust_map / mcgpu_rt2_image
  float *d_eps2 = nullptr, *d_eps2_star = nullptr;
  double* d_rt2_zgrid = nullptr;
  int rt2_src_ibin = 0, rt2_src_lambda = 0, rt2_src_nang = 0, rt2_src_nang_star = 0;
  size_t n_xI = 0;
  int xI_bytes = 8;  // accumulator type of xI_scatt on the device: 8 = FP64 (default), 4 = default real (mcgpu_set_xI_precision)
  double* d_prob_E = nullptr;               // prob_E_cell(0:n_cells) of the current wavelength
  int prob_E_lambda = 0;                    // the wavelength mcgpu_repartition_energie left in d_prob_E (0: none)
  double prob_E_fstar = 0.0, prob_E_fdisk = 0.0;
  unsigned long long* d_mono_u64 = nullptr; // [5 * n_chunks + 1]: need | sent | item_base(+1) | start | hit_count
  int* d_mono_i32 = nullptr;                // [2 * n_chunks]: active | done
  int mono_chunks = 0;
  unsigned char* d_hits = nullptr;
  size_t hits_cap = 0;
  // the SED commit pass's deposit log (mc_mono.hip.h "The deposits as a log", mc_xilog.hip.h)
  int opt_xi_log = 1;               // 1 (default): default-real xI_scatt of one dust class on a cylindrical grid is summed from a log
                                    // where a wavelength's flights are long enough for that to pay; 0: atomics; 2: the log always
  unsigned int* d_xlog_keys[2] = {nullptr, nullptr};       // [0]: the launch's log; [1]: the sorted copy
  unsigned long long* d_xlog_vals[2] = {nullptr, nullptr};
  float* d_xlog_rows = nullptr;
  unsigned long long* d_xlog_ctl = nullptr;     // [0] records, [1] flights the launch's waves reserved
  size_t xlog_cap = 0, xlog_rows_cap = 0, xlog_row_floats = 0;
  void* d_xlog_temp = nullptr;
  size_t xlog_temp_bytes = 0;
  int xlog_chunks = 0;              // launches of the last wavelength's commit passes
  unsigned long long xlog_records = 0, xlog_flights = 0;   // ... and what they logged
  // Voronoi grid (mc_voronoi.hip.h)
  bool voro = false;
  VoroGrid V;
  std::vector<VoroCell> h_cells;  // host copy: kappa_factor is patched in by mcgpu_set_opacity
  VoroCell* d_cells = nullptr;
  void* d_pool = nullptr;          // the pool schedule's packet records (mc_voronoi_pool.hip.h), [blocks][1 << log_rec] x 128 B
  size_t pool_bytes = 0;
  VpBlob* d_pool_blob = nullptr;   // ... and the copy of a launch's arguments its emission phase reads
  VpBlob h_pool_blob;
};

static void bin_release(mcgpu_ctx* ctx);
static void grid_release(mcgpu_ctx* ctx);
static int tail_threshold(const mcgpu_ctx* ctx);

#define HIPCHK(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);               \
      return MCGPU_ERR_HIP;                                                       \
    }                                                                             \
  } while (0)

// n values from the host into a new device array of max(n, n_alloc) values (the rest zeroed)
template <typename Tp>
static int upload(mcgpu_ctx* ctx, const Tp* host, size_t n, const Tp** dev_out, size_t n_alloc = 0) {
  Tp* d = nullptr;
  const size_t na = n_alloc > n ? n_alloc : n;
  HIPCHK(hipMalloc((void**)&d, (na ? na : 1) * sizeof(Tp)));
  ctx->allocs.push_back(d);
  if (na > n) HIPCHK(hipMemset(d, 0, na * sizeof(Tp)));
  if (n) HIPCHK(hipMemcpy(d, host, n * sizeof(Tp), hipMemcpyHostToDevice));
  *dev_out = d;
  return MCGPU_OK;
}

// xI_scatt on the device: values (of xI_bytes each) and bytes.  FP64: one line of XI_LINE values per (sub-bin, observer);
// default real: the packed layout of mc_xi32.hip.h
static inline Xi32Lay xi_layout_of(const mcgpu_ctx* ctx) { return xi32_layout(ctx->RT_n_incl * ctx->RT_n_az, ctx->lsepar_pola != 0, ctx->lsepar_contrib != 0); }
static inline int xi_bin_floats_of(const mcgpu_ctx* ctx) { return xi_layout_of(ctx).binf; }
static inline size_t xi_dev_values(const mcgpu_ctx* ctx) {
  const size_t bins = (size_t)ctx->n_az_rt * ctx->n_theta_rt * (size_t)ctx->M.n_cells;
  return ctx->xI_bytes == 4 ? bins * (size_t)xi_bin_floats_of(ctx) : bins * XI_LINE * (size_t)(ctx->RT_n_incl * ctx->RT_n_az);
}
static inline size_t xi_dev_bytes(const mcgpu_ctx* ctx) { return xi_dev_values(ctx) * (size_t)ctx->xI_bytes; }

// a device array that lives as long as the scope
template <typename Tp>
struct DevBuf {
  Tp* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(Tp)); }
  hipError_t put(const Tp* h, size_t n) { return hipMemcpy(p, h, n * sizeof(Tp), hipMemcpyHostToDevice); }
  hipError_t get(Tp* h, size_t n) { return hipMemcpy(h, p, n * sizeof(Tp), hipMemcpyDeviceToHost); }
};

static int fail(mcgpu_ctx* ctx, int code, const char* msg) {
  if (ctx) ctx->err = msg;
  return code;
}

extern "C" int mcgpu_create(int device, mcgpu_ctx** out) {
  if (!out) return MCGPU_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MCGPU_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return MCGPU_ERR_ARG;
  mcgpu_ctx* ctx = new mcgpu_ctx();
  ctx->device = device;
  std::memset(&ctx->M, 0, sizeof(DevModel));
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&ctx->prop, device) != hipSuccess ||
      hipStreamCreate(&ctx->own_stream) != hipSuccess || hipEventCreate(&ctx->ev0) != hipSuccess ||
      hipEventCreate(&ctx->ev1) != hipSuccess || hipEventCreate(&ctx->ev_tail) != hipSuccess) {
    delete ctx;
    return MCGPU_ERR_HIP;
  }
  ctx->stream = ctx->own_stream;
  ctx->M.midplane_snap = 0;  // the reference's literal arithmetic; mcgpu_set_midplane_snap(ctx, 1) is the option
  if (hipMalloc((void**)&ctx->d_counters, CNT_SLOTS * sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc((void**)&ctx->d_err, sizeof(int)) != hipSuccess) {
    delete ctx;
    return MCGPU_ERR_HIP;
  }
  hipMemset(ctx->d_counters, 0, CNT_SLOTS * sizeof(unsigned long long));
  hipMemset(ctx->d_err, 0, sizeof(int));
  *out = ctx;
  return MCGPU_OK;
}

extern "C" int mcgpu_destroy(mcgpu_ctx* ctx) {
  if (!ctx) return MCGPU_OK;
  hipSetDevice(ctx->device);
  hipDeviceSynchronize();
  for (void* p : ctx->allocs) hipFree(p);
  if (ctx->d_accum) hipFree(ctx->d_accum);
  if (ctx->d_counters) hipFree(ctx->d_counters);
  if (ctx->d_xN) hipFree(ctx->d_xN);
  if (ctx->d_xJ) hipFree(ctx->d_xJ);
  if (ctx->d_err) hipFree(ctx->d_err);
  if (ctx->d_E_prior) hipFree(ctx->d_E_prior);
  if (ctx->d_xI) hipFree(ctx->d_xI);
  if (ctx->d_I_spec) hipFree(ctx->d_I_spec);
  if (ctx->d_vkk) hipFree(ctx->d_vkk);
  if (ctx->d_I_spec_star) hipFree(ctx->d_I_spec_star);
  if (ctx->d_prob_E) hipFree(ctx->d_prob_E);
  if (ctx->d_mono_u64) hipFree(ctx->d_mono_u64);
  if (ctx->d_mono_i32) hipFree(ctx->d_mono_i32);
  if (ctx->d_hits) hipFree(ctx->d_hits);
  if (ctx->d_pool) hipFree(ctx->d_pool);
  if (ctx->d_pool_blob) hipFree(ctx->d_pool_blob);
  for (int i = 0; i < 2; ++i) { if (ctx->d_xlog_keys[i]) hipFree(ctx->d_xlog_keys[i]); if (ctx->d_xlog_vals[i]) hipFree(ctx->d_xlog_vals[i]); }
  if (ctx->d_xlog_rows) hipFree(ctx->d_xlog_rows);
  if (ctx->d_xlog_ctl) hipFree(ctx->d_xlog_ctl);
  if (ctx->d_xlog_temp) hipFree(ctx->d_xlog_temp);
  if (ctx->d_tail_ctl) hipFree(ctx->d_tail_ctl);
  if (ctx->d_tail_out) hipFree(ctx->d_tail_out);
  if (ctx->h_arena) hipHostFree(ctx->h_arena);
  if (ctx->side_stream) hipStreamDestroy(ctx->side_stream);
  if (ctx->ev_side_in) hipEventDestroy(ctx->ev_side_in);
  if (ctx->ev_side_out) hipEventDestroy(ctx->ev_side_out);
  bin_release(ctx);
  if (ctx->ev0) hipEventDestroy(ctx->ev0);
  if (ctx->ev1) hipEventDestroy(ctx->ev1);
  if (ctx->ev_tail) hipEventDestroy(ctx->ev_tail);
  if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return MCGPU_OK;
}

extern "C" const char* mcgpu_last_error(const mcgpu_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int mcgpu_set_stream(mcgpu_ctx* ctx, void* s) {
  if (!ctx) return MCGPU_ERR_ARG;
  ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
  return MCGPU_OK;
}

extern "C" int mcgpu_set_grid_cyl(mcgpu_ctx* ctx, int n_rad, int nz, int n_az, int l3D,
                                  const double* r_lim_2, const double* zmax, const double* z_lim,
                                  const double* tan_phi_lim, double zmaxmax, double Rmax2,
                                  const double* volume, const int* cell_map, const int* cell_map_i,
                                  const int* cell_map_j, const int* cell_map_k, const int* lexit_cell) {
  if (!ctx || n_rad < 1 || nz < 1 || n_az < 1 || !r_lim_2 || !zmax || !z_lim || !tan_phi_lim || !volume ||
      !cell_map || !cell_map_i || !cell_map_j || !cell_map_k || !lexit_cell)
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_grid_cyl: bad argument");
  if (!l3D && n_az != 1) return fail(ctx, MCGPU_ERR_ARG, "2D grid needs n_az = 1");
  if (ctx->voro) return fail(ctx, MCGPU_ERR_STATE, "the context already holds a Voronoi grid");
  HIPCHK(hipSetDevice(ctx->device));
  const int n_cells = l3D ? 2 * n_rad * nz * n_az : n_rad * nz;
  const int jlo = l3D ? -nz - 1 : 0;
  ctx->h_r_lim.resize((size_t)n_rad + 1);
  for (int i = 0; i <= n_rad; ++i) ctx->h_r_lim[i] = std::sqrt(r_lim_2[i]);
  ctx->tau_midplane = -1.0; ctx->last_inter_pp = -1.0;
  const int jn = nz + 1 - jlo + 1;
  const int ntot2 = l3D ? (n_rad + 2) * (2 * nz + 2) * n_az : (n_rad + 2) * (nz + 2) * n_az;
  // verify the closed-form mapping against the host's arrays
  for (int k = 1; k <= n_az; ++k)
    for (int j = jlo; j <= nz + 1; ++j) {
      if (l3D && j == 0) continue;
      for (int i = 0; i <= n_rad + 1; ++i) {
        const int ic = cell_map[i + (n_rad + 2) * ((j - jlo) + jn * (k - 1))];
        if (ic != icell_of(n_rad, nz, n_az, l3D, i, j, k))
          return fail(ctx, MCGPU_ERR_UNSUPPORTED, "cell_map differs from build_cylindrical_cell_mapping order");
        if (ic < 1 || ic > ntot2 || cell_map_i[ic - 1] != i || cell_map_j[ic - 1] != j || cell_map_k[ic - 1] != k)
          return fail(ctx, MCGPU_ERR_UNSUPPORTED, "cell_map_i/j/k inconsistent with cell_map");
        const int aj = j < 0 ? -j : j;
        int le = 0;
        if (ic > n_cells) {
          if (i == n_rad + 1) le = 1;
          else if (aj == nz + 1) le = 2;
        }
        if (lexit_cell[ic - 1] != le) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "lexit_cell differs from the reference rule");
      }
    }
  // verify the uniform vertical grid the kernel evaluates in closed form
  std::vector<double> ch(n_rad);
  for (int i = 1; i <= n_rad; ++i) {
    ch[i - 1] = (nz >= 2) ? z_lim[(i - 1) + n_rad * 1] : zmax[i - 1];
    for (int j = 1; j <= nz; ++j)
      if (z_lim[(i - 1) + n_rad * (j - 1)] != ((double)j - 1.0) * ch[i - 1])
        return fail(ctx, MCGPU_ERR_UNSUPPORTED, "z_lim is not the uniform grid (j-1)*cell_height");
    if (z_lim[(i - 1) + n_rad * nz] != zmax[i - 1]) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "z_lim(i,nz+1) != zmax(i)");
    if (z_lim[(i - 1) + n_rad * (nz + 1)] != (double)1.0e30f)
      return fail(ctx, MCGPU_ERR_UNSUPPORTED, "z_lim(i,nz+2) != 1.0e30");
  }
  DevModel& M = ctx->M;
  M.n_rad = n_rad; M.nz = nz; M.n_az = n_az; M.l3D = l3D ? 1 : 0; M.n_cells = n_cells;
  M.zmaxmax = zmaxmax; M.Rmax2 = Rmax2;
  int rc;
  if ((rc = upload(ctx, r_lim_2, (size_t)n_rad + 1, &M.r_lim_2))) return rc;
  if ((rc = upload(ctx, zmax, (size_t)n_rad, &M.zmax))) return rc;
  if ((rc = upload(ctx, ch.data(), (size_t)n_rad, &M.ch))) return rc;
  if ((rc = upload(ctx, tan_phi_lim, (size_t)n_az, &M.tan_phi_lim))) return rc;
  if ((rc = upload(ctx, volume, (size_t)n_cells, &M.volume))) return rc;
  const int *a, *b, *c;
  if ((rc = upload(ctx, cell_map_i, (size_t)ntot2, &a))) return rc;
  if ((rc = upload(ctx, cell_map_j, (size_t)ntot2, &b))) return rc;
  if ((rc = upload(ctx, cell_map_k, (size_t)ntot2, &c))) return rc;
  ctx->d_cmi = (int*)a; ctx->d_cmj = (int*)b; ctx->d_cmk = (int*)c;
  bin_release(ctx);
  grid_release(ctx);   // (d_prob_E, xN_abs, xJ_abs were sized for the old grid)
  ctx->have_grid = true;
  return MCGPU_OK;
}

// The spherical grid (grid_type = 2): the arrays define_cylindrical_grid fills in its spherical branch
// (cylindrical_grid.f90:496-580) and the cell mapping both structured grids share.
extern "C" int mcgpu_set_grid_sph(mcgpu_ctx* ctx, int n_rad, int nz, int n_az, int l3D, const double* r_lim_2,
                                  const double* r_lim_3, const double* tan_theta_lim, const double* theta_lim,
                                  const double* tan_phi_lim, double Rmax2, const double* volume, const int* cell_map,
                                  const int* cell_map_i, const int* cell_map_j, const int* cell_map_k,
                                  const int* lexit_cell) {
  if (!ctx || n_rad < 1 || nz < 1 || n_az < 1 || !r_lim_2 || !r_lim_3 || !tan_theta_lim || !theta_lim || !tan_phi_lim ||
      !volume || !cell_map || !cell_map_i || !cell_map_j || !cell_map_k || !lexit_cell)
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_grid_sph: bad argument");
  if (!l3D && n_az != 1) return fail(ctx, MCGPU_ERR_ARG, "2D grid needs n_az = 1");
  if (ctx->have_grid) return fail(ctx, MCGPU_ERR_STATE, "the grid of a context is set once");
  HIPCHK(hipSetDevice(ctx->device));
  const int n_cells = l3D ? 2 * n_rad * nz * n_az : n_rad * nz;
  const int jlo = l3D ? -nz - 1 : 0;
  const int jn = nz + 1 - jlo + 1;
  const int ntot2 = l3D ? (n_rad + 2) * (2 * nz + 2) * n_az : (n_rad + 2) * (nz + 2) * n_az;
  for (int k = 1; k <= n_az; ++k)   // the closed-form mapping of build_cylindrical_cell_mapping, as for the cylindrical grid
    for (int j = jlo; j <= nz + 1; ++j) {
      if (l3D && j == 0) continue;
      for (int i = 0; i <= n_rad + 1; ++i) {
        const int ic = cell_map[i + (n_rad + 2) * ((j - jlo) + jn * (k - 1))];
        if (ic != icell_of(n_rad, nz, n_az, l3D, i, j, k))
          return fail(ctx, MCGPU_ERR_UNSUPPORTED, "cell_map differs from build_cylindrical_cell_mapping order");
        if (ic < 1 || ic > ntot2 || cell_map_i[ic - 1] != i || cell_map_j[ic - 1] != j || cell_map_k[ic - 1] != k)
          return fail(ctx, MCGPU_ERR_UNSUPPORTED, "cell_map_i/j/k inconsistent with cell_map");
        int le = 0;
        const int aj = j < 0 ? -j : j;
        if (ic > n_cells) { if (i == n_rad + 1) le = 1; else if (aj == nz + 1) le = 2; }
        if (lexit_cell[ic - 1] != le) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "lexit_cell differs from the reference rule");
      }
    }
  for (int j = 1; j <= nz; ++j)
    if (!(tan_theta_lim[j] > tan_theta_lim[j - 1])) return fail(ctx, MCGPU_ERR_ARG, "tan_theta_lim must increase");
  DevModel& M = ctx->M;
  M.n_rad = n_rad; M.nz = nz; M.n_az = n_az; M.l3D = l3D ? 1 : 0; M.n_cells = n_cells;
  M.zmaxmax = 0.0; M.Rmax2 = Rmax2; M.grid_sph = 1;
  std::vector<double> ones(n_rad, 1.0);   // (the cylindrical vectors the shared LDS carve stages; unused by this grid)
  int rc;
  if ((rc = upload(ctx, r_lim_2, (size_t)n_rad + 1, &M.r_lim_2))) return rc;
  if ((rc = upload(ctx, ones.data(), (size_t)n_rad, &M.zmax))) return rc;
  if ((rc = upload(ctx, ones.data(), (size_t)n_rad, &M.ch))) return rc;
  if ((rc = upload(ctx, tan_phi_lim, (size_t)n_az, &M.tan_phi_lim))) return rc;
  if ((rc = upload(ctx, volume, (size_t)n_cells, &M.volume))) return rc;
  if ((rc = upload(ctx, r_lim_3, (size_t)n_rad + 1, &M.r_lim_3))) return rc;
  if ((rc = upload(ctx, tan_theta_lim, (size_t)nz + 1, &M.tan_theta_lim))) return rc;
  if ((rc = upload(ctx, theta_lim, (size_t)nz + 1, &M.theta_lim))) return rc;
  const int *a, *b, *c;
  if ((rc = upload(ctx, cell_map_i, (size_t)ntot2, &a))) return rc;
  if ((rc = upload(ctx, cell_map_j, (size_t)ntot2, &b))) return rc;
  if ((rc = upload(ctx, cell_map_k, (size_t)ntot2, &c))) return rc;
  ctx->d_cmi = (int*)a; ctx->d_cmj = (int*)b; ctx->d_cmk = (int*)c;
  bin_release(ctx);
  grid_release(ctx);   // (d_prob_E, xN_abs, xJ_abs were sized for the old grid)
  ctx->have_grid = true;
  return MCGPU_OK;
}

// The arrays Voronoi_tesselation hands to the packet loop (Voronoi.f90:23-67, 385-640).
extern "C" int mcgpu_set_grid_voronoi(mcgpu_ctx* ctx, int n_cells, const float* voronoi_xyz,
                                      const double* xyz_dp, const double* h, const int* first_neighbour,
                                      const int* last_neighbour, const int* neighbours_list,
                                      long long n_neighbours, const unsigned char* was_cut,
                                      const unsigned char* is_star_neighbour, const float* walls,
                                      double cutting_distance_o_h, const int* wall_first,
                                      const int* wall_cells, const double* volume) {
  if (!ctx || n_cells < 1 || !voronoi_xyz || !xyz_dp || !h || !first_neighbour || !last_neighbour ||
      !neighbours_list || n_neighbours < 1 || !walls || !wall_first || !wall_cells || !volume)
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_grid_voronoi: bad argument");
  if (ctx->have_grid) return fail(ctx, MCGPU_ERR_STATE, "the grid of a context is set once");
  for (int iw = 0; iw < 6; ++iw) {  // plane walls in the order of Voronoi.f90:1275-1280
    const float* W = walls + 4 * iw;
    const float want = (iw & 1) ? 1.0f : -1.0f;
    for (int a = 0; a < 3; ++a)
      if (W[a] != ((a == iw / 2) ? want : 0.0f))
        return fail(ctx, MCGPU_ERR_UNSUPPORTED, "walls must be the 6 axis-aligned planes (-x,+x,-y,+y,-z,+z)");
  }
  HIPCHK(hipSetDevice(ctx->device));
  std::vector<VoroNb> nb((size_t)n_neighbours);
  ctx->h_cells.assign((size_t)n_cells, VoroCell{});
  for (int i = 0; i < n_cells; ++i) {
    const long long f = first_neighbour[i], l = last_neighbour[i];
    if (f < 1 || l < f - 1 || l > n_neighbours)
      return fail(ctx, MCGPU_ERR_ARG, "first_neighbour/last_neighbour out of range");
    VoroCell& C = ctx->h_cells[i];
    C.x = voronoi_xyz[3 * (size_t)i]; C.y = voronoi_xyz[3 * (size_t)i + 1]; C.z = voronoi_xyz[3 * (size_t)i + 2];
    C.first = (int)(f - 1);
    C.count = (int)(l - f + 1);
    C.flags = ((was_cut && was_cut[i]) ? 1 : 0) | ((is_star_neighbour && is_star_neighbour[i]) ? 2 : 0);
    C.kf = 0.0;
    for (long long q = f - 1; q < l; ++q) {
      const int id = neighbours_list[q];
      if (id == 0 || id > n_cells || id < -6) return fail(ctx, MCGPU_ERR_ARG, "neighbours_list entry out of range");
      VoroNb& N = nb[(size_t)q];
      N.id = id;
      if (id > 0) {
        N.x = voronoi_xyz[3 * (size_t)(id - 1)]; N.y = voronoi_xyz[3 * (size_t)(id - 1) + 1];
        N.z = voronoi_xyz[3 * (size_t)(id - 1) + 2];
      } else {
        N.x = N.y = N.z = 0.0f;
      }
    }
  }
  // per (cell, neighbour): the class of the neighbour cell's list length -- where the pool schedule queues a packet that
  // enters it (mc_voronoi_pool.hip.h)
  std::vector<unsigned char> nb_cls((size_t)n_neighbours, (unsigned char)0);
  for (size_t q = 0; q < (size_t)n_neighbours; ++q)
    if (nb[q].id > 0) nb_cls[q] = (unsigned char)vp_class_of(ctx->h_cells[(size_t)nb[q].id - 1].count);
  if (wall_first[0] != 0) return fail(ctx, MCGPU_ERR_ARG, "wall_first[0] must be 0");
  for (int iw = 0; iw < 6; ++iw)
    if (wall_first[iw + 1] < wall_first[iw]) return fail(ctx, MCGPU_ERR_ARG, "wall_first must not decrease");
  for (int q = 0; q < wall_first[6]; ++q)
    if (wall_cells[q] < 1 || wall_cells[q] > n_cells) return fail(ctx, MCGPU_ERR_ARG, "wall_cells entry out of range");
  DevModel& M = ctx->M;
  M.n_rad = 0; M.nz = 0; M.n_az = 0; M.l3D = 1; M.n_cells = n_cells;
  VoroGrid& V = ctx->V;
  V.n_cells = n_cells;
  V.cut_o_h = cutting_distance_o_h;
  std::memcpy(V.walls, walls, 24 * sizeof(float));
  int rc;
  const double dummy = 0.0;
  if ((rc = upload(ctx, &dummy, 1, &M.r_lim_2))) return rc;  // the shared LDS carve stages r_lim_2(0:n_rad)
  if ((rc = upload(ctx, nb.data(), nb.size(), &V.nb))) return rc;
  if ((rc = upload(ctx, nb_cls.data(), nb_cls.size(), &V.nb_cls))) return rc;
  if ((rc = upload(ctx, h, (size_t)n_cells, &V.h))) return rc;
  if ((rc = upload(ctx, xyz_dp, 3 * (size_t)n_cells, &V.xyz_dp))) return rc;
  if ((rc = upload(ctx, wall_first, 7, &V.wall_first))) return rc;
  if ((rc = upload(ctx, wall_cells, (size_t)wall_first[6], &V.wall_cells))) return rc;
  if ((rc = upload(ctx, volume, (size_t)n_cells, &M.volume))) return rc;
  const VoroCell* dc;
  if ((rc = upload(ctx, ctx->h_cells.data(), ctx->h_cells.size(), &dc))) return rc;
  ctx->d_cells = (VoroCell*)dc;
  V.cell = dc;
  ctx->voro = true;
  bin_release(ctx);
  grid_release(ctx);   // (d_prob_E, xN_abs, xJ_abs were sized for the old grid)
  ctx->have_grid = true;
  return MCGPU_OK;
}

// the binned-deposit log is allocated by the first launch that uses it and kept until the context goes (or the grid /
// the option that sizes it changes)
// what was sized for the grid that is being replaced
static void grid_release(mcgpu_ctx* ctx) {
  if (ctx->d_prob_E) hipFree(ctx->d_prob_E);
  ctx->d_prob_E = nullptr;
  ctx->prob_E_lambda = 0;
  if (ctx->d_xN) hipFree(ctx->d_xN);
  ctx->d_xN = nullptr;
  if (ctx->d_xJ) hipFree(ctx->d_xJ);
  ctx->d_xJ = nullptr;
}

static void bin_release(mcgpu_ctx* ctx) {
  if (ctx->bin.keys) hipFree(ctx->bin.keys);
  if (ctx->bin.vals) hipFree(ctx->bin.vals);
  if (ctx->bin.count) hipFree(ctx->bin.count);
  if (ctx->bin.stats) hipFree(ctx->bin.stats);
  if (ctx->d_bin_off) hipFree(ctx->d_bin_off);
  if (ctx->d_bin_cap) hipFree(ctx->d_bin_cap);
  if (ctx->d_bin_want) hipFree(ctx->d_bin_want);
  ctx->d_bin_want = nullptr;
  for (int i = 0; i < 2; ++i) { if (ctx->d_carry[i]) hipFree(ctx->d_carry[i]); ctx->d_carry[i] = nullptr; }
  if (ctx->d_carry_n) hipFree(ctx->d_carry_n);
  if (ctx->d_tail_next) hipFree(ctx->d_tail_next);
  ctx->d_carry_n = nullptr;
  ctx->d_tail_next = nullptr;
  ctx->carry_cap = 0;
  ctx->bin = BinLog{};
  ctx->d_bin_off = ctx->d_bin_cap = nullptr;
  ctx->bin_total_blocks = 0;
  ctx->bin_max_parts = 0;
}

extern "C" int mcgpu_set_option(mcgpu_ctx* ctx, const char* name, int value) {
  if (!ctx || !name) return MCGPU_ERR_ARG;
  if (!strcmp(name, "deposit")) { if (value < 0 || value > 3) return fail(ctx, MCGPU_ERR_ARG, "deposit: 0, 1, 2 or 3"); ctx->opt_deposit = value; }
  else if (!strcmp(name, "tail")) { if (value < -1 || value > (1 << 20)) return fail(ctx, MCGPU_ERR_ARG, "tail: -1 (automatic), 0 (off), or the packets left per workgroup at the hand-over"); ctx->opt_tail = value; }
  else if (!strcmp(name, "xi_log")) { if (value < 0 || value > 2) return fail(ctx, MCGPU_ERR_ARG, "xi_log: 0, 1 or 2"); ctx->opt_xi_log = value; }
  else if (!strcmp(name, "tail_where")) { if (value < 0 || value > 2) return fail(ctx, MCGPU_ERR_ARG, "tail_where: 0 (automatic), 1 (device), 2 (host)"); ctx->opt_tail_where = value; }
  else if (!strcmp(name, "host_threads")) { if (value < 0 || value > 256) return fail(ctx, MCGPU_ERR_ARG, "host_threads: 0 (automatic) .. 256"); ctx->opt_host_threads = value; }
  else if (!strcmp(name, "tail_host_packets")) { if (value < 0 || value > 65536) return fail(ctx, MCGPU_ERR_ARG, "tail_host_packets: 0 (automatic) .. 65536"); ctx->opt_tail_host_max = value; }
  else if (!strcmp(name, "deposit_log_mb")) {
    if (value < 0) return fail(ctx, MCGPU_ERR_ARG, "deposit_log_mb: >= 0");
    if (value != ctx->opt_log_mb) bin_release(ctx);
    ctx->opt_log_mb = value;
  }
  else if (!strcmp(name, "schedule")) { if (value < 0 || value > 3) return fail(ctx, MCGPU_ERR_ARG, "schedule: 0, 1, 2 or 3"); ctx->opt_schedule = value; }
  else if (!strcmp(name, "speculation")) ctx->opt_speculation = value ? 1 : 0;
  else if (!strcmp(name, "crossing")) { if (value < 0 || value > 1) return fail(ctx, MCGPU_ERR_ARG, "crossing: 0 or 1"); ctx->opt_crossing = value; }
  else if (!strcmp(name, "voronoi_pool_log_records")) { if (value < 6 || value > VP_MAX_LOG_REC) return fail(ctx, MCGPU_ERR_ARG, "voronoi_pool_log_records: 6..12"); ctx->opt_pool_log_rec = value; }
  else if (!strcmp(name, "voronoi_cache_log_slots")) { if (value < 6 || value > 13) return fail(ctx, MCGPU_ERR_ARG, "voronoi_cache_log_slots: 6..13"); ctx->opt_cache_log_slots = value; }
  else if (!strcmp(name, "radiation_field")) { if (value < 0 || value > 3) return fail(ctx, MCGPU_ERR_ARG, "radiation_field: bit 0 xN_abs, bit 1 xJ_abs"); ctx->opt_radiation_field = value; }
  else return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_option: unknown option");
  return MCGPU_OK;
}

extern "C" int mcgpu_get_info(mcgpu_ctx* ctx, const char* name, double* value) {
  if (!ctx || !name || !value) return MCGPU_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  if (!strcmp(name, "bin_buckets")) *value = ctx->bin.n_buckets;
  else if (!strcmp(name, "bin_log_blocks")) *value = (double)ctx->bin_total_blocks;
  else if (!strcmp(name, "bin_log_bytes")) *value = (double)ctx->bin_total_blocks * (double)BIN_H * (double)(sizeof(double) + sizeof(unsigned int));
  else if (!strcmp(name, "bin_chunks")) *value = ctx->bin_chunks;
  else if (!strcmp(name, "bin_deposits_per_packet")) *value = ctx->bin_dep_per_packet;
  else if (!strcmp(name, "tail_threshold")) *value = tail_threshold(ctx);
  else if (!strcmp(name, "tail_ms")) {   // k_tail's share of the last thermal launch (0: that launch had no tail kernel)
    *value = 0.0;
    if (ctx->launched && ctx->tail_launched) {
      HIPCHK(hipStreamSynchronize(ctx->stream));
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, ctx->ev_tail, ctx->ev1));
      *value = ms;
    }
  } else if (!strcmp(name, "longest_packet_events")) {   // crossings + interactions of the longest packet binned by the
    unsigned long long v = 0ull;                            // context's launches since the last non-accumulating one
    if (ctx->d_counters) {
      HIPCHK(hipStreamSynchronize(ctx->stream));
      HIPCHK(hipMemcpy(&v, ctx->d_counters + 10, sizeof(v), hipMemcpyDeviceToHost));
    }
    *value = (double)v;
  }
  else if (!strncmp(name, "longest_packet_", 15) && strcmp(name, "longest_packet_events")) {
    // the longest packet's own counts, as far as the tail kernel ran it (mc_tail.hip.h: the packet with the most events
    // wins every slot, its events in the upper half of the word): crossings, scatterings, absorptions, walks, steps
    static const char* what[5] = {"crossings", "scatterings", "absorptions", "walks", "steps"};
    int k = -1;
    for (int q = 0; q < 5; ++q) if (!strcmp(name + 15, what[q])) k = q;
    if (k < 0) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_get_info: unknown name");
    unsigned long long v = 0ull;
    if (ctx->d_counters) {
      HIPCHK(hipStreamSynchronize(ctx->stream));
      HIPCHK(hipMemcpy(&v, ctx->d_counters + TAIL_LONGEST + k, sizeof(v), hipMemcpyDeviceToHost));
    }
    *value = (double)(v & 0xFFFFFFFFull);
  }
  else if (!strcmp(name, "tail_where") || !strncmp(name, "tail_host_", 10)) {
    // where the last thermal launch finished its last packets (0: it had no tail, 1: k_tail, 2: host threads) and what the
    // host did: its wall time, packets, threads, events (crossings + interactions)
    *value = 0.0;
    if (ctx->launched && ctx->tail_launched) {
      HIPCHK(hipStreamSynchronize(ctx->stream));
      if (!strcmp(name, "tail_where")) *value = ctx->tail_on_host ? 2.0 : 1.0;
      else if (!ctx->tail_on_host) *value = 0.0;
      else if (!strcmp(name, "tail_host_ms")) *value = ctx->host_tail_ms;
      else if (!strcmp(name, "tail_host_packets")) *value = (double)ctx->host_tail_packets;
      else if (!strcmp(name, "tail_host_threads")) *value = (double)ctx->host_tail_threads;
      else if (!strcmp(name, "tail_host_events")) *value = (double)ctx->host_tail_events;
      else return fail(ctx, MCGPU_ERR_ARG, "mcgpu_get_info: unknown name");
    }
  }
  else if (!strcmp(name, "xi_log_chunks")) *value = ctx->xlog_chunks;     // the last mcgpu_run_mono: launches of its commit passes,
  else if (!strcmp(name, "xi_log_records")) *value = (double)ctx->xlog_records;   // records (crossings with a deposit) and
  else if (!strcmp(name, "xi_log_flights")) *value = (double)ctx->xlog_flights;   // flights they logged (0: atomics)
  else if (!strcmp(name, "tau_midplane")) *value = ctx->tau_midplane;
  // the packed default-real layout of xI_scatt this context would use (mc_xi32.hip.h; needs mcgpu_set_rt1): default reals
  // per sub-bin, 64-byte lines one crossing's deposits touch, 1 = the split arrangement
  else if (!strcmp(name, "xi_bin_floats")) *value = xi_layout_of(ctx).binf;
  else if (!strcmp(name, "xi_lines_per_crossing")) *value = xi32_lines_touched(xi_layout_of(ctx), ctx->RT_n_incl * ctx->RT_n_az);
  else if (!strcmp(name, "xi_split")) *value = xi_layout_of(ctx).split;
  else if (!strcmp(name, "bin_overflow_blocks") || !strcmp(name, "bin_drained_records")) {
    unsigned long long st[2] = {0ull, 0ull};
    if (ctx->bin.stats) {
      HIPCHK(hipStreamSynchronize(ctx->stream));
      HIPCHK(hipMemcpy(st, ctx->bin.stats, sizeof(st), hipMemcpyDeviceToHost));
    }
    *value = (double)st[!strcmp(name, "bin_overflow_blocks") ? 0 : 1];
  } else return fail(ctx, MCGPU_ERR_ARG, "mcgpu_get_info: unknown name");
  return MCGPU_OK;
}

extern "C" int mcgpu_set_midplane_snap(mcgpu_ctx* ctx, int on) {
  if (!ctx) return MCGPU_ERR_ARG;
  ctx->M.midplane_snap = on ? 1 : 0;
  return MCGPU_OK;
}

extern "C" int mcgpu_set_stars(mcgpu_ctx* ctx, int n_stars, const double* x, const double* y,
                               const double* z, const double* r, const int* icell, const int* out_model) {
  if (!ctx || n_stars < 1 || !x || !y || !z || !r || !icell || !out_model)
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_stars: bad argument");
  if (!ctx->have_grid) return fail(ctx, MCGPU_ERR_STATE, "set the grid before the stars");
  HIPCHK(hipSetDevice(ctx->device));
  const DevModel& G = ctx->M;
  std::vector<double> xyzr(4 * n_stars);
  std::vector<int> cell(4 * n_stars);
  // invert the closed-form mapping on the host by search over (i,j,k)
  const int jlo = G.l3D ? -G.nz - 1 : 0;
  for (int s = 0; s < n_stars; ++s) {
    xyzr[4 * s + 0] = x[s]; xyzr[4 * s + 1] = y[s]; xyzr[4 * s + 2] = z[s]; xyzr[4 * s + 3] = r[s];
    if (ctx->voro) {  // a star inside the box is a site of its own (Voronoi.f90:357-376)
      if (icell[s] < 0 || icell[s] > G.n_cells || (icell[s] == 0 && !out_model[s]))
        return fail(ctx, MCGPU_ERR_ARG, "star icell is not a cell of the Voronoi grid");
      cell[4 * s + 0] = icell[s]; cell[4 * s + 1] = 0; cell[4 * s + 2] = 0;
      cell[4 * s + 3] = out_model[s] ? 1 : 0;
      continue;
    }
    bool found = false;
    for (int k = 1; k <= G.n_az && !found; ++k)
      for (int j = jlo; j <= G.nz + 1 && !found; ++j) {
        if (G.l3D && j == 0) continue;
        for (int i = 0; i <= G.n_rad + 1; ++i)
          if (icell_of(G.n_rad, G.nz, G.n_az, G.l3D, i, j, k) == icell[s]) {
            cell[4 * s + 0] = i; cell[4 * s + 1] = j; cell[4 * s + 2] = k;
            found = true;
            break;
          }
      }
    if (!found) return fail(ctx, MCGPU_ERR_ARG, "star icell is not a cell of the grid");
    cell[4 * s + 3] = out_model[s] ? 1 : 0;
  }
  ctx->M.n_stars = n_stars;
  int rc;
  if ((rc = upload(ctx, xyzr.data(), xyzr.size(), &ctx->M.star_xyzr))) return rc;
  if ((rc = upload(ctx, cell.data(), cell.size(), &ctx->M.star_cell))) return rc;
  ctx->have_stars = true;
  return MCGPU_OK;
}

extern "C" int mcgpu_set_opacity(mcgpu_ctx* ctx, int n_lambda, const double* kappa, const double* kappa_abs_LTE,
                                 const float* tab_albedo_pos, const double* kappa_factor,
                                 const unsigned char* l_dark_zone) {
  if (!ctx || n_lambda < 1 || !kappa || !kappa_abs_LTE || !tab_albedo_pos || !kappa_factor)
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_opacity: bad argument");
  if (!ctx->have_grid) return fail(ctx, MCGPU_ERR_STATE, "set the grid before the opacities");
  if (ctx->M.n_lambda && ctx->M.n_lambda != n_lambda) return fail(ctx, MCGPU_ERR_ARG, "n_lambda mismatch");
  HIPCHK(hipSetDevice(ctx->device));
  ctx->vkk_valid = false;  // (kappa_factor enters the variable-dust role kernel's per-cell pairs)
  DevModel& M = ctx->M;
  M.n_lambda = n_lambda;
  int rc;
  if ((rc = upload(ctx, kappa, (size_t)n_lambda, &M.kappa))) return rc;
  if ((rc = upload(ctx, kappa_abs_LTE, (size_t)n_lambda, &M.kappa_abs))) return rc;
  if ((rc = upload(ctx, tab_albedo_pos, (size_t)n_lambda, &M.albedo))) return rc;
  ctx->tau_midplane = -1.0; ctx->last_inter_pp = -1.0;
  if (!ctx->voro && !M.grid_sph && (int)ctx->h_r_lim.size() == M.n_rad + 1) {
    // radial optical depth through the first layer above the midplane (cells (i, j = 1, k = 1)) at the most opaque wavelength
    double kmax = 0.0, tau = 0.0;
    for (int l = 0; l < n_lambda; ++l) kmax = kappa[l] > kmax ? kappa[l] : kmax;
    for (int i = 0; i < M.n_rad; ++i) {
      const size_t ic = M.l3D ? (size_t)i + (size_t)M.n_rad * (size_t)M.nz : (size_t)i;   // (3D: j = +1 is row nz of k = 1)
      tau += kmax * kappa_factor[ic] * (ctx->h_r_lim[i + 1] - ctx->h_r_lim[i]);
    }
    ctx->tau_midplane = tau;
  }
  {  // one extra entry, 0: the factor of "no cell" (the 2D crossing reads it for the virtual cells, mc_roles.hip.h)
    std::vector<double> kfp((size_t)M.n_cells + 1, 0.0);
    std::memcpy(kfp.data(), kappa_factor, (size_t)M.n_cells * sizeof(double));
    if ((rc = upload(ctx, kfp.data(), kfp.size(), &M.kappa_factor))) return rc;
  }
  M.dark = nullptr;
  if (l_dark_zone) {
    bool any = false;
    for (int i = 0; i < M.n_cells; ++i) any |= (l_dark_zone[i] != 0);
    if (any && ctx->voro)  // the reference never builds a dark zone there (dust_transfer.f90:290-293)
      return fail(ctx, MCGPU_ERR_UNSUPPORTED, "no dark zone on a Voronoi grid");
    if (any && (rc = upload(ctx, l_dark_zone, (size_t)M.n_cells, &M.dark))) return rc;
  }
  if (ctx->voro) {  // the cell records carry the opacity factor
    for (int i = 0; i < M.n_cells; ++i) ctx->h_cells[i].kf = kappa_factor[i];
    HIPCHK(hipMemcpy(ctx->d_cells, ctx->h_cells.data(), ctx->h_cells.size() * sizeof(VoroCell), hipMemcpyHostToDevice));
  }
  ctx->have_opacity = true;
  return MCGPU_OK;
}

extern "C" int mcgpu_set_scattering(mcgpu_ctx* ctx, int nang_scatt, int aniso_method, int lisotropic,
                                    int lsepar_pola, int p_lambda_fixed, const float* prob_s11_pos,
                                    const float* s12, const float* s22, const float* s33, const float* s34,
                                    const float* s44, const float* tab_g_pos) {
  if (!ctx || nang_scatt < 2 || (aniso_method != 1 && aniso_method != 2) || !prob_s11_pos || !tab_g_pos)
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_scattering: bad argument");
  if (!ctx->M.n_lambda) return fail(ctx, MCGPU_ERR_STATE, "set the opacities before the scattering tables");
  if (lsepar_pola && aniso_method == 1 && (!s12 || !s22 || !s33 || !s34 || !s44))
    return fail(ctx, MCGPU_ERR_ARG, "lsepar_pola needs the five Mueller ratio tables");
  HIPCHK(hipSetDevice(ctx->device));
  DevModel& M = ctx->M;
  M.nang = nang_scatt; M.aniso_method = aniso_method; M.lisotropic = lisotropic ? 1 : 0;
  M.p_lambda_fixed = p_lambda_fixed ? 1 : 0;
  ctx->lsepar_pola = (lsepar_pola && aniso_method == 1) ? 1 : 0;
  const size_t nt = (size_t)(nang_scatt + 1) * M.n_lambda;
  int rc;
  if ((rc = upload(ctx, prob_s11_pos, nt, &M.prob_s11))) return rc;
  if ((rc = upload(ctx, tab_g_pos, (size_t)M.n_lambda, &M.tab_g))) return rc;
  {  // bin-edge cosines of angle_diff_theta_pos (scattering.f90:1470-1471), same expression
    std::vector<double> ct(nang_scatt + 1);
    for (int k = 0; k <= nang_scatt; ++k) ct[k] = std::cos(((double)k) * PI / (double)nang_scatt);
    if ((rc = upload(ctx, ct.data(), ct.size(), &M.cos_tab))) return rc;
  }
  if (ctx->lsepar_pola) {
    if ((rc = upload(ctx, s12, nt, &M.s12))) return rc;
    if ((rc = upload(ctx, s22, nt, &M.s22))) return rc;
    if ((rc = upload(ctx, s33, nt, &M.s33))) return rc;
    if ((rc = upload(ctx, s34, nt, &M.s34))) return rc;
    if ((rc = upload(ctx, s44, nt, &M.s44))) return rc;
  }
  ctx->have_scatt = true;
  return MCGPU_OK;
}

extern "C" int mcgpu_set_thermal(mcgpu_ctx* ctx, int n_T, const float* tab_Temp, const double* log_Qcool,
                                 const double* kdB_dT_CDF, const double* spectre_emission_cumul,
                                 const double* frac_E_stars, const double* frac_E_disk,
                                 const double* CDF_E_star, const double* prob_E_cell, double L_packet_th,
                                 float T_min) {
  if (!ctx || n_T < 2 || !tab_Temp || !spectre_emission_cumul || !frac_E_stars || !frac_E_disk || !CDF_E_star ||
      ((log_Qcool == nullptr) != (kdB_dT_CDF == nullptr)))
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_thermal: bad argument");
  if (!ctx->have_opacity || !ctx->have_stars) return fail(ctx, MCGPU_ERR_STATE, "set opacities and stars first");
  for (int t = 2; log_Qcool && t < n_T; ++t)
    if (log_Qcool[t] < log_Qcool[t - 1]) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "log_Qcool must increase with T");
  HIPCHK(hipSetDevice(ctx->device));
  DevModel& M = ctx->M;
  M.n_T = n_T; M.L_packet_th = L_packet_th;
  ctx->T_min = T_min;
  int rc;
  const float* tt;
  if ((rc = upload(ctx, tab_Temp, (size_t)n_T, &tt))) return rc;
  ctx->d_tab_Temp = (float*)tt;
  // both NULL: the two re-emission tables are left to mcgpu_init_reemission (a launch before it is refused)
  ctx->pending_single = (log_Qcool == nullptr);
  ctx->reemission_pending = ctx->pending_single || ctx->pending_classes;
  if ((rc = upload(ctx, log_Qcool, log_Qcool ? (size_t)n_T : 0, &M.log_Qcool, (size_t)n_T))) return rc;
  if ((rc = upload(ctx, kdB_dT_CDF, kdB_dT_CDF ? (size_t)n_T * M.n_lambda : 0, &M.cdf, (size_t)n_T * M.n_lambda))) return rc;
  if ((rc = upload(ctx, spectre_emission_cumul, (size_t)M.n_lambda + 1, &M.spec_cum))) return rc;
  if ((rc = upload(ctx, frac_E_stars, (size_t)M.n_lambda, &M.frac_E_stars))) return rc;
  if ((rc = upload(ctx, frac_E_disk, (size_t)M.n_lambda, &M.frac_E_disk))) return rc;
  if ((rc = upload(ctx, CDF_E_star, (size_t)M.n_lambda * (M.n_stars + 1), &M.CDF_E_star))) return rc;
  M.prob_E_cell = nullptr;
  if (prob_E_cell && (rc = upload(ctx, prob_E_cell, (size_t)(M.n_cells + 1) * M.n_lambda, &M.prob_E_cell))) return rc;
  if (!prob_E_cell)
    for (int l = 0; l < M.n_lambda; ++l)
      if (frac_E_stars[l] < 1.0) return fail(ctx, MCGPU_ERR_ARG, "frac_E_stars < 1 needs prob_E_cell");
  ctx->have_thermal = true;
  return MCGPU_OK;
}

// lvariable_dust (mem.f90:213-244: the tables gain the cell axis p_n_cells): see include/mcgpu.h.  The arrays come in
// the reference's own layouts and are re-laid class-major for the device.
extern "C" int mcgpu_set_variable_dust(mcgpu_ctx* ctx, int p_n_cells, const int* p_icell, const double* kappa,
                                       const double* kappa_abs_LTE, const float* tab_albedo_pos,
                                       const double* log_Qcool, const double* kdB_dT_CDF, const float* prob_s11_pos,
                                       const float* tab_s12_o_s11_pos, const float* tab_s22_o_s11_pos,
                                       const float* tab_s33_o_s11_pos, const float* tab_s34_o_s11_pos,
                                       const float* tab_s44_o_s11_pos, const float* tab_g_pos) {
  if (!ctx) return MCGPU_ERR_ARG;
  DevModel& M = ctx->M;
  if (p_n_cells == 0) {  // back to one class (what the class tables were waiting for no longer matters)
    M.n_classes = 0;
    M.m1 = 0;
    ctx->pending_classes = false;
    ctx->reemission_pending = ctx->pending_single;
    return MCGPU_OK;
  }
  if (p_n_cells < 1 || !p_icell || !kappa || !kappa_abs_LTE || !tab_albedo_pos ||
      ((log_Qcool == nullptr) != (kdB_dT_CDF == nullptr)))
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_variable_dust: bad argument");
  if (!ctx->have_grid || !ctx->have_opacity || !ctx->have_thermal)
    return fail(ctx, MCGPU_ERR_STATE, "set the grid, the opacities and the thermal tables first");
  for (int i = 0; i < M.n_cells; ++i)
    if (p_icell[i] < 1 || p_icell[i] > p_n_cells) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_variable_dust: p_icell out of range");
  const int nc = p_n_cells, nl = M.n_lambda, nT = M.n_T;
  for (int c = 0; log_Qcool && c < nc; ++c)
    for (int t = 2; t < nT; ++t)
      if (log_Qcool[(size_t)c * nT + t] < log_Qcool[(size_t)c * nT + t - 1])
        return fail(ctx, MCGPU_ERR_UNSUPPORTED, "log_Qcool must increase with T");
  HIPCHK(hipSetDevice(ctx->device));
  std::vector<int> cls(M.n_cells);
  for (int i = 0; i < M.n_cells; ++i) cls[i] = p_icell[i] - 1;
  // kappa(p_n_cells, n_lambda), kappa_abs_LTE, tab_albedo_pos: class fastest in the reference -> [class][lambda]
  std::vector<double> k((size_t)nc * nl), ka((size_t)nc * nl);
  std::vector<float> al((size_t)nc * nl);
  for (int c = 0; c < nc; ++c)
    for (int l = 0; l < nl; ++l) {
      k[(size_t)c * nl + l] = kappa[(size_t)c + (size_t)nc * l];
      ka[(size_t)c * nl + l] = kappa_abs_LTE[(size_t)c + (size_t)nc * l];
      al[(size_t)c * nl + l] = tab_albedo_pos[(size_t)c + (size_t)nc * l];
    }
  int rc;
  if ((rc = upload(ctx, cls.data(), (size_t)M.n_cells, &M.cell_class))) return rc;
  if ((rc = upload(ctx, k.data(), k.size(), &M.v_kappa))) return rc;
  if ((rc = upload(ctx, ka.data(), ka.size(), &M.v_kabs))) return rc;
  if ((rc = upload(ctx, al.data(), al.size(), &M.v_albedo))) return rc;
  // log_Qcool_minus_extra_heating(n_T, p_n_cells) and kdB_dT_CDF(n_lambda, n_T, p_n_cells): class slowest already
  // (both NULL: left to mcgpu_init_reemission -- 280 MB at 7000 classes that never cross the bus)
  ctx->pending_classes = (log_Qcool == nullptr);
  ctx->reemission_pending = ctx->pending_single || ctx->pending_classes;
  if ((rc = upload(ctx, log_Qcool, log_Qcool ? (size_t)nc * nT : 0, &M.v_lq, (size_t)nc * nT))) return rc;
  if ((rc = upload(ctx, kdB_dT_CDF, kdB_dT_CDF ? (size_t)nc * nT * nl : 0, &M.v_cdf, (size_t)nc * nT * nl))) return rc;
  // scattering tables per class (all or none): (0:nang, p_n_cells, n_lambda) in the reference -> [class][lambda][angle]
  M.v_scatt = 0;
  M.v_s11 = nullptr;
  M.m1 = 0;  // (scattering method 1 belongs to a set of classes: mcgpu_set_scattering_method1 again)
  const bool any_sc = prob_s11_pos || tab_s12_o_s11_pos || tab_s22_o_s11_pos || tab_s33_o_s11_pos || tab_s34_o_s11_pos ||
                      tab_s44_o_s11_pos || tab_g_pos;
  if (any_sc) {
    if (!ctx->have_scatt) return fail(ctx, MCGPU_ERR_STATE, "set the scattering tables first");
    if (!(prob_s11_pos && tab_s12_o_s11_pos && tab_s22_o_s11_pos && tab_s33_o_s11_pos && tab_s34_o_s11_pos &&
          tab_s44_o_s11_pos && tab_g_pos))
      return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_variable_dust: pass all seven scattering tables or none");
    const int na1 = M.nang + 1, ncol = M.p_lambda_fixed ? 1 : nl;
    auto relay = [&](const float* src, int cols, const float** dst) {
      std::vector<float> t((size_t)nc * cols * na1);
      for (int c = 0; c < nc; ++c)
        for (int l = 0; l < cols; ++l)
          std::memcpy(&t[((size_t)c * cols + l) * na1], &src[((size_t)l * nc + c) * na1], na1 * sizeof(float));
      return upload(ctx, t.data(), t.size(), dst);
    };
    if ((rc = relay(prob_s11_pos, ncol, &M.v_prob))) return rc;   // (p_lambda_fixed: only the column of p_lambda = 1 is read)
    if ((rc = relay(tab_s12_o_s11_pos, nl, &M.v_s12))) return rc;
    if ((rc = relay(tab_s22_o_s11_pos, nl, &M.v_s22))) return rc;
    if ((rc = relay(tab_s33_o_s11_pos, nl, &M.v_s33))) return rc;
    if ((rc = relay(tab_s34_o_s11_pos, nl, &M.v_s34))) return rc;
    if ((rc = relay(tab_s44_o_s11_pos, nl, &M.v_s44))) return rc;
    std::vector<float> gg((size_t)nc * nl);
    for (int c = 0; c < nc; ++c)
      for (int l = 0; l < nl; ++l) gg[(size_t)c * nl + l] = tab_g_pos[(size_t)c + (size_t)nc * l];
    if ((rc = upload(ctx, gg.data(), gg.size(), &M.v_g))) return rc;
    M.v_scatt = 1;
  }
  M.n_classes = nc;
  ctx->vkk_valid = false;
  return MCGPU_OK;
}

// tab_s11_pos(0:nang, p_n_cells, n_lambda) of the classes: the phase function the rt1 deposits and the ray tracer read
extern "C" int mcgpu_set_variable_dust_s11(mcgpu_ctx* ctx, const float* tab_s11_pos) {
  if (!ctx || !tab_s11_pos) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_variable_dust_s11: bad argument");
  DevModel& M = ctx->M;
  if (!M.n_classes) return fail(ctx, MCGPU_ERR_STATE, "mcgpu_set_variable_dust first");
  HIPCHK(hipSetDevice(ctx->device));
  const int nc = M.n_classes, nl = M.n_lambda, na1 = M.nang + 1;
  std::vector<float> t((size_t)nc * nl * na1);
  for (int c = 0; c < nc; ++c)
    for (int l = 0; l < nl; ++l)
      std::memcpy(&t[((size_t)c * nl + l) * na1], &tab_s11_pos[((size_t)l * nc + c) * na1], na1 * sizeof(float));
  return upload(ctx, t.data(), t.size(), &M.v_s11);
}

// scattering method 1 (dust_transfer.f90:1288-1316): see include/mcgpu.h
extern "C" int mcgpu_set_scattering_method1(mcgpu_ctx* ctx, const mcgpu_grain_tables* G, const float* prob_s11,
                                            int p_n_cells, const double* dust_density_o_n_grains) {
  if (!ctx) return MCGPU_ERR_ARG;
  DevModel& M = ctx->M;
  if (!G) { M.m1 = 0; return MCGPU_OK; }  // off: back to method 2
  if (!M.n_classes) return fail(ctx, MCGPU_ERR_STATE, "scattering method 1 runs on a variable-dust context: mcgpu_set_variable_dust or mcgpu_opacity first");
  if (p_n_cells != M.n_classes || !dust_density_o_n_grains) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_scattering_method1: the densities of the context's classes");
  const int ng = G->n_grains, nl = M.n_lambda, na1 = M.nang + 1;
  const bool mueller = M.aniso_method == 1, pola = ctx->lsepar_pola != 0;
  if (ng < 1 || !G->C_sca || !G->n_grains_k || (mueller && !prob_s11) || (!mueller && !G->tab_g) ||
      (mueller && pola && (!G->tab_s11 || !G->tab_s12 || !G->tab_s22 || !G->tab_s33 || !G->tab_s34 || !G->tab_s44)))
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_scattering_method1: a grain table is missing");
  HIPCHK(hipSetDevice(ctx->device));
  int rc;
  const size_t n_gl = (size_t)ng * nl, n_agl = n_gl * na1;
  if ((rc = upload(ctx, G->C_sca, n_gl, &M.m1_Csca)) || (rc = upload(ctx, G->n_grains_k, (size_t)ng, &M.m1_nk)) ||
      (rc = upload(ctx, dust_density_o_n_grains, (size_t)ng * p_n_cells, &M.m1_dens)))
    return rc;
  if (mueller) {
    // prob_s11(n_lambda, n_grains, 0:nang) -> [lambda][grain][angle]: the bisection of one (wavelength, grain) walks one row
    std::vector<float> t(n_agl);
    for (int l = 0; l < nl; ++l)
      for (int k = 0; k < ng; ++k)
        for (int a = 0; a < na1; ++a) t[((size_t)l * ng + k) * na1 + a] = prob_s11[(size_t)l + (size_t)nl * ((size_t)k + (size_t)ng * a)];
    if ((rc = upload(ctx, t.data(), t.size(), &M.m1_prob))) return rc;
    if (pola && ((rc = upload(ctx, G->tab_s11, n_agl, &M.m1_s11)) || (rc = upload(ctx, G->tab_s12, n_agl, &M.m1_s12)) ||
                 (rc = upload(ctx, G->tab_s22, n_agl, &M.m1_s22)) || (rc = upload(ctx, G->tab_s33, n_agl, &M.m1_s33)) ||
                 (rc = upload(ctx, G->tab_s34, n_agl, &M.m1_s34)) || (rc = upload(ctx, G->tab_s44, n_agl, &M.m1_s44))))
      return rc;
  } else if ((rc = upload(ctx, G->tab_g, n_gl, &M.m1_g))) return rc;
  M.m1_ng = ng;
  M.m1 = 1;
  M.m1_ksca = nullptr;   // (low_mem_scattering until mcgpu_build_ksca_CDF)
  return MCGPU_OK;
}

// ksca_CDF on the device (dust_prop.f90:976-994): the grain selection of scattering method 1 then is the dichotomy of
// select_grainsize_high_mem instead of the walk of the low-memory mode -- the reference takes this branch when
// n_grains x p_n_cells x n_lambda x 4 bytes fit max_mem (mem.f90:245-258); 288 GB of HBM make that the rule.
// ksca_CDF_out (or NULL): the table in the reference's layout (0:n_grains, p_n_cells, n_lambda).  build = 0: back to
// the low-memory walk.
extern "C" int mcgpu_build_ksca_CDF(mcgpu_ctx* ctx, int build, double* ksca_CDF_out) {
  if (!ctx) return MCGPU_ERR_ARG;
  DevModel& M = ctx->M;
  if (!build) { M.m1_ksca = nullptr; return MCGPU_OK; }
  if (!M.m1 || !M.n_classes) return fail(ctx, MCGPU_ERR_STATE, "mcgpu_build_ksca_CDF: mcgpu_set_scattering_method1 first");
  HIPCHK(hipSetDevice(ctx->device));
  const int ng = M.m1_ng, nc = M.n_classes, nl = M.n_lambda;
  const size_t n = (size_t)nc * nl * (ng + 1);
  double* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, n * sizeof(double)));
  ctx->allocs.push_back(d);
  hipLaunchKernelGGL(k_ksca_cdf, dim3((unsigned)((nc * nl + 127) / 128)), dim3(128), 0, ctx->stream, nc, nl, ng, M.m1_Csca, M.m1_dens,
                     M.m1_nk, d);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ksca_CDF_out) {   // [class][lambda][k] -> (k, class, lambda) with k fastest
    std::vector<double> h(n);
    HIPCHK(hipMemcpy(h.data(), d, n * sizeof(double), hipMemcpyDeviceToHost));
    for (int c = 0; c < nc; ++c)
      for (int l = 0; l < nl; ++l)
        for (int k = 0; k <= ng; ++k)
          ksca_CDF_out[(size_t)k + (size_t)(ng + 1) * ((size_t)c + (size_t)nc * l)] = h[((size_t)c * nl + l) * (ng + 1) + k];
  }
  M.m1_ksca = d;
  return MCGPU_OK;
}

// opacity + calc_local_scattering_matrices on the device (dust_prop.f90:791-1243): see include/mcgpu.h
extern "C" int mcgpu_opacity(mcgpu_ctx* ctx, const mcgpu_grain_tables* G, int p_n_cells, const int* p_icell,
                             const double* dust_density_o_n_grains, const mcgpu_opacity_tables* out) {
  if (!ctx || !G || !p_icell || !dust_density_o_n_grains || p_n_cells < 1) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_opacity: bad argument");
  if (!ctx->have_grid || !ctx->have_opacity || !ctx->have_thermal || !ctx->have_scatt)
    return fail(ctx, MCGPU_ERR_STATE, "set the grid, the opacities, the scattering and the thermal tables first");
  DevModel& M = ctx->M;
  const int ng = G->n_grains, nl = M.n_lambda, nc = p_n_cells, nT = M.n_T, na1 = M.nang + 1;
  const bool pola = ctx->lsepar_pola != 0, mueller = M.aniso_method == 1;
  if (ng < 1 || G->grain_RE_LTE_start < 1 || G->grain_RE_LTE_end > ng || !G->C_ext || !G->C_sca || !G->C_abs || !G->S_grain ||
      !G->n_grains_k || (M.aniso_method == 2 && !G->tab_g) || (mueller && !G->tab_s11) ||
      (mueller && pola && (!G->tab_s12 || !G->tab_s22 || !G->tab_s33 || !G->tab_s34 || !G->tab_s44)))
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_opacity: a grain table is missing");
  for (int i = 0; i < M.n_cells; ++i)
    if (p_icell[i] < 1 || p_icell[i] > nc) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_opacity: p_icell out of range");
  HIPCHK(hipSetDevice(ctx->device));
  int rc;
  // the grains' tables: here for the time of the call
  DevBuf<float> dCe, dCs, dCa, dg, d11, d12, d22, d33, d34, d44, dS;
  DevBuf<double> dn, dd;
  auto put = [&](auto& buf, const auto* host, size_t n) -> int {
    if (!host) return MCGPU_OK;
    using Tp = std::remove_cv_t<std::remove_pointer_t<decltype(host)>>;
    HIPCHK(hipMalloc((void**)&buf.p, n * sizeof(Tp)));
    HIPCHK(hipMemcpyAsync(buf.p, host, n * sizeof(Tp), hipMemcpyHostToDevice, ctx->stream));
    return MCGPU_OK;
  };
  const size_t n_gl = (size_t)ng * nl, n_agl = (size_t)na1 * ng * nl;
  if ((rc = put(dCe, G->C_ext, n_gl)) || (rc = put(dCs, G->C_sca, n_gl)) || (rc = put(dCa, G->C_abs, n_gl)) ||
      (rc = put(dg, M.aniso_method == 2 ? G->tab_g : nullptr, n_gl)) || (rc = put(dS, G->S_grain, (size_t)ng)) ||
      (rc = put(dn, G->n_grains_k, (size_t)ng)) || (rc = put(dd, dust_density_o_n_grains, (size_t)ng * nc)))
    return rc;
  if (mueller) {
    if ((rc = put(d11, G->tab_s11, n_agl))) return rc;
    if (pola && ((rc = put(d12, G->tab_s12, n_agl)) || (rc = put(d22, G->tab_s22, n_agl)) || (rc = put(d33, G->tab_s33, n_agl)) ||
                 (rc = put(d34, G->tab_s34, n_agl)) || (rc = put(d44, G->tab_s44, n_agl))))
      return rc;
  }
  // the context's per-class tables (mcgpu_set_variable_dust's): built in place; those of the last call go
  for (void* q : ctx->opacity_allocs) {
    for (auto it = ctx->allocs.begin(); it != ctx->allocs.end(); ++it)
      if (*it == q) { ctx->allocs.erase(it); break; }
    hipFree(q);
  }
  ctx->opacity_allocs.clear();
  M.n_classes = 0;
  M.m1 = 0;
  const size_t allocs_before = ctx->allocs.size();
  std::vector<int> cls(M.n_cells);
  for (int i = 0; i < M.n_cells; ++i) cls[i] = p_icell[i] - 1;
  const int pcols = M.p_lambda_fixed ? 1 : nl;
  const size_t n_cl = (size_t)nc * nl, n_cla = n_cl * na1;
  if ((rc = upload(ctx, cls.data(), (size_t)M.n_cells, &M.cell_class))) return rc;
  if ((rc = upload<double>(ctx, nullptr, 0, &M.v_kappa, n_cl)) || (rc = upload<double>(ctx, nullptr, 0, &M.v_kabs, n_cl)) ||
      (rc = upload<float>(ctx, nullptr, 0, &M.v_albedo, n_cl)) || (rc = upload<float>(ctx, nullptr, 0, &M.v_g, n_cl)) ||
      (rc = upload<double>(ctx, nullptr, 0, &M.v_lq, (size_t)nc * nT)) || (rc = upload<double>(ctx, nullptr, 0, &M.v_cdf, (size_t)nc * nT * nl)) ||
      (rc = upload<float>(ctx, nullptr, 0, &M.v_s11, n_cla)) || (rc = upload<float>(ctx, nullptr, 0, &M.v_prob, (size_t)nc * pcols * na1)) ||
      (rc = upload<float>(ctx, nullptr, 0, &M.v_s12, n_cla)) || (rc = upload<float>(ctx, nullptr, 0, &M.v_s22, n_cla)) ||
      (rc = upload<float>(ctx, nullptr, 0, &M.v_s33, n_cla)) || (rc = upload<float>(ctx, nullptr, 0, &M.v_s34, n_cla)) ||
      (rc = upload<float>(ctx, nullptr, 0, &M.v_s44, n_cla)))
    return rc;
  ctx->opacity_allocs.assign(ctx->allocs.begin() + allocs_before, ctx->allocs.end());
  OpacityIn I{ng, nl, nc, M.nang, M.aniso_method, pola ? 1 : 0, G->grain_RE_LTE_start, G->grain_RE_LTE_end, pcols,
              dCe.p, dCs.p, dCa.p, dg.p, d11.p, d12.p, d22.p, d33.p, d34.p, d44.p, dS.p, dn.p, dd.p};
  OpacityOut O{const_cast<double*>(M.v_kappa), const_cast<double*>(M.v_kabs), const_cast<float*>(M.v_albedo), const_cast<float*>(M.v_g),
               const_cast<float*>(M.v_s11), const_cast<float*>(M.v_prob), const_cast<float*>(M.v_s12), const_cast<float*>(M.v_s22),
               const_cast<float*>(M.v_s33), const_cast<float*>(M.v_s34), const_cast<float*>(M.v_s44)};
  const int threads = 64, n_rows = nc * nl;
  hipLaunchKernelGGL(k_opacity_sum, dim3((n_rows + threads - 1) / threads), dim3(threads), 0, ctx->stream, I, O);
  if (mueller) {
    const dim3 grid((na1 + 63) / 64, nc, nl);
    if (pola) hipLaunchKernelGGL(k_scatt_sum<true>, grid, dim3(64), 0, ctx->stream, I, O);
    else hipLaunchKernelGGL(k_scatt_sum<false>, grid, dim3(64), 0, ctx->stream, I, O);
  }
  hipLaunchKernelGGL(k_scatt_norm, dim3((n_rows + threads - 1) / threads), dim3(threads), 0, ctx->stream, I, O);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  M.v_scatt = 1;
  M.n_classes = nc;
  ctx->vkk_valid = false;
  ctx->pending_classes = true;  // log_Qcool and kdB_dT_CDF of the classes: mcgpu_init_reemission, from the new kappa_abs_LTE
  ctx->reemission_pending = true;
  if (out) {  // copies in the reference's layouts: (p_n_cells, n_lambda) and (0:nang, p_n_cells, n_lambda)
    auto fetch2 = [&](const auto* dev, auto* host) -> int {
      if (!host) return MCGPU_OK;
      using Tp = std::remove_cv_t<std::remove_pointer_t<decltype(dev)>>;
      std::vector<Tp> t(n_cl);
      HIPCHK(hipMemcpy(t.data(), dev, n_cl * sizeof(Tp), hipMemcpyDeviceToHost));
      for (int c = 0; c < nc; ++c)
        for (int l = 0; l < nl; ++l) host[(size_t)c + (size_t)nc * l] = t[(size_t)c * nl + l];
      return MCGPU_OK;
    };
    auto fetch3 = [&](const float* dev, float* host, int cols) -> int {
      if (!host) return MCGPU_OK;
      std::vector<float> t((size_t)nc * cols * na1);
      HIPCHK(hipMemcpy(t.data(), dev, t.size() * sizeof(float), hipMemcpyDeviceToHost));
      for (int c = 0; c < nc; ++c)
        for (int l = 0; l < cols; ++l)
          std::memcpy(&host[((size_t)l * nc + c) * na1], &t[((size_t)c * cols + l) * na1], na1 * sizeof(float));
      return MCGPU_OK;
    };
    if ((rc = fetch2(M.v_kappa, out->kappa)) || (rc = fetch2(M.v_kabs, out->kappa_abs_LTE)) || (rc = fetch2(M.v_albedo, out->tab_albedo_pos)) ||
        (rc = fetch2(M.v_g, out->tab_g_pos)) || (rc = fetch3(M.v_s11, out->tab_s11_pos, nl)) || (rc = fetch3(M.v_prob, out->prob_s11_pos, pcols)))
      return rc;
    if (pola && ((rc = fetch3(M.v_s12, out->tab_s12_o_s11_pos, nl)) || (rc = fetch3(M.v_s22, out->tab_s22_o_s11_pos, nl)) ||
                 (rc = fetch3(M.v_s33, out->tab_s33_o_s11_pos, nl)) || (rc = fetch3(M.v_s34, out->tab_s34_o_s11_pos, nl)) ||
                 (rc = fetch3(M.v_s44, out->tab_s44_o_s11_pos, nl))))
      return rc;
  }
  return MCGPU_OK;
}

// init_reemission on the device (thermal_emission.f90:404-550): see include/mcgpu.h
extern "C" int mcgpu_init_reemission(mcgpu_ctx* ctx, const double* tab_lambda, const double* tab_delta_lambda,
                                     double* log_Qcool, double* kdB_dT_CDF) {
  return mcgpu_init_reemission_ex(ctx, tab_lambda, tab_delta_lambda, nullptr, nullptr, 0.0, log_Qcool, kdB_dT_CDF);
}

// lextra_heating (thermal_emission.f90:486-494, 622-631): dudt[nc], heating_norm[nc] = AU_to_m^2 volume kappa_factor per
// class (nc = p_n_cells, or 1); ufac_implicit > 0: ldudt_implicit.  Both NULL: mcgpu_init_reemission.
extern "C" int mcgpu_init_reemission_ex(mcgpu_ctx* ctx, const double* tab_lambda, const double* tab_delta_lambda,
                                        const double* dudt, const double* heating_norm, double ufac_implicit,
                                        double* log_Qcool, double* kdB_dT_CDF) {
  if (!ctx || !tab_lambda || !tab_delta_lambda || ((dudt == nullptr) != (heating_norm == nullptr)))
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_init_reemission: bad argument");
  if (!ctx->have_opacity || !ctx->have_thermal) return fail(ctx, MCGPU_ERR_STATE, "set the opacities and the thermal tables first");
  HIPCHK(hipSetDevice(ctx->device));
  DevModel& M = ctx->M;
  const int nl = M.n_lambda, nT = M.n_T;
  for (int l = 0; l < nl; ++l)
    if (!(tab_lambda[l] > 0.0) || !(tab_delta_lambda[l] > 0.0))
      return fail(ctx, MCGPU_ERR_ARG, "mcgpu_init_reemission: wavelengths and bin widths must be positive");
  double *d_lam = nullptr, *d_dlam = nullptr;
  HIPCHK(hipMalloc((void**)&d_lam, 2 * (size_t)nl * sizeof(double)));
  d_dlam = d_lam + nl;
  hipError_t e = hipMemcpy(d_lam, tab_lambda, (size_t)nl * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_dlam, tab_delta_lambda, (size_t)nl * sizeof(double), hipMemcpyHostToDevice);
  // the extra heating's per-class terms (they belong to the tables the loop reads: the classes' when there are classes)
  const int n_heat = M.n_classes ? M.n_classes : 1;
  double *d_dudt = nullptr, *d_hnorm = nullptr;
  if (dudt && e == hipSuccess) {
    for (int c = 0; c < n_heat; ++c)
      if (!(heating_norm[c] > 0.0)) { hipFree(d_lam); return fail(ctx, MCGPU_ERR_ARG, "mcgpu_init_reemission: heating_norm must be positive"); }
    e = hipMalloc((void**)&d_dudt, 2 * (size_t)n_heat * sizeof(double));
    if (e == hipSuccess) { d_hnorm = d_dudt + n_heat; e = hipMemcpy(d_dudt, dudt, (size_t)n_heat * sizeof(double), hipMemcpyHostToDevice); }
    if (e == hipSuccess) e = hipMemcpy(d_hnorm, heating_norm, (size_t)n_heat * sizeof(double), hipMemcpyHostToDevice);
  }
  auto build = [&](int nc, const double* kabs, const double* lq, const double* cdf, bool heat) {
    const int n = nc * nT, threads = 64;  // one (class, T) row per thread: short rows, many of them
    hipLaunchKernelGGL(k_init_reemission, dim3((n + threads - 1) / threads), dim3(threads), 0, ctx->stream, nc, nT, nl,
                       ctx->d_tab_Temp, d_lam, d_dlam, kabs, const_cast<double*>(lq), const_cast<double*>(cdf),
                       heat ? d_dudt : nullptr, heat ? d_hnorm : nullptr, ufac_implicit);
  };
  // the tables the setters left to this call; called with none pending, it rebuilds all of them (an explicit request),
  // otherwise tables the host supplied are left alone
  const bool all = !ctx->pending_single && !ctx->pending_classes;
  const bool do_single = all || ctx->pending_single, do_classes = M.n_classes && (all || ctx->pending_classes);
  if (e == hipSuccess) {
    if (do_single) build(1, M.kappa_abs, M.log_Qcool, M.cdf, !M.n_classes);
    if (do_classes) build(M.n_classes, M.v_kabs, M.v_lq, M.v_cdf, true);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  // the same gate as the setters: Temp_LTE's search needs log_Qcool to increase with T
  const int nc = M.n_classes ? M.n_classes : 1;
  const double* d_lq = M.n_classes ? M.v_lq : M.log_Qcool;
  const double* d_cdf = M.n_classes ? M.v_cdf : M.cdf;
  std::vector<double> lq((size_t)nc * nT);
  if (e == hipSuccess) e = hipMemcpy(lq.data(), d_lq, lq.size() * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && kdB_dT_CDF)
    e = hipMemcpy(kdB_dT_CDF, d_cdf, (size_t)nc * nT * nl * sizeof(double), hipMemcpyDeviceToHost);
  hipFree(d_lam);
  if (d_dudt) hipFree(d_dudt);
  if (e != hipSuccess) return fail(ctx, MCGPU_ERR_HIP, hipGetErrorString(e));
  if (log_Qcool) std::memcpy(log_Qcool, lq.data(), lq.size() * sizeof(double));
  // (a table that fails the gate stays in the context -- it was built in place --, so the context is marked as waiting for
  // its re-emission tables again: ready() refuses every launch until a build, or tables from the host, pass)
  auto refuse = [&]() {
    ctx->pending_single = ctx->pending_single || do_single;
    ctx->pending_classes = ctx->pending_classes || do_classes;
    ctx->reemission_pending = true;
    return fail(ctx, MCGPU_ERR_UNSUPPORTED, "log_Qcool must increase with T (the tables just built do not: no launch until they are replaced)");
  };
  for (int c = 0; c < nc; ++c)
    for (int t = 2; t < nT; ++t)
      if (lq[(size_t)c * nT + t] < lq[(size_t)c * nT + t - 1]) return refuse();
  if (M.n_classes && do_single) {  // (the single-class table too: spherical / MRW / SED paths read it)
    std::vector<double> lq1((size_t)nT);
    HIPCHK(hipMemcpy(lq1.data(), M.log_Qcool, lq1.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int t = 2; t < nT; ++t)
      if (lq1[t] < lq1[t - 1]) return refuse();
  }
  ctx->pending_single = ctx->pending_classes = false;
  ctx->reemission_pending = false;
  return MCGPU_OK;
}

// Modified random walk (MRW.f90; dust_transfer.f90:1222-1239): see include/mcgpu.h
extern "C" int mcgpu_set_mrw(mcgpu_ctx* ctx, int n_zeta, const double* zeta, const double* chi, const double* kappa_dep,
                             const double* ext, double gamma, int n_interactions, const double* r_lim) {
  if (!ctx) return MCGPU_ERR_ARG;
  DevModel& M = ctx->M;
  if (n_zeta == 0) { M.mrw = 0; return MCGPU_OK; }  // off
  if (n_zeta < 2 || !zeta || !chi || !kappa_dep || !ext || (!r_lim && !ctx->voro) || !(gamma > 0.0) || n_interactions < 0 || n_interactions > 6)
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_mrw: bad argument (n_interactions is 0..6)");
  if (!ctx->have_grid || !ctx->have_thermal) return fail(ctx, MCGPU_ERR_STATE, "set the grid and the thermal tables first");
  for (int i = 1; i < n_zeta; ++i)
    if (!(zeta[i] >= zeta[i - 1])) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_mrw: zeta must not decrease");
  HIPCHK(hipSetDevice(ctx->device));
  int rc;
  if ((rc = upload(ctx, zeta, (size_t)n_zeta, &M.mrw_zeta))) return rc;
  {  // the guide of mrw_sample_y (mc_device.hip.h): per bucket edge b / MRW_GUIDE the last entry not above it
    std::vector<int> guide((size_t)MRW_GUIDE + 1);
    int i = 0;
    for (int b = 0; b <= MRW_GUIDE; ++b) {
      const double edge = (double)b / (double)MRW_GUIDE;
      while (i + 1 < n_zeta && zeta[i + 1] <= edge) ++i;
      guide[b] = i;   // (zeta[0] = 0 <= every edge)
    }
    if ((rc = upload(ctx, guide.data(), guide.size(), &M.mrw_guide))) return rc;
  }
  // (lvariable_dust, set before this call: one row of n_T values per class)
  const size_t n_tab = (size_t)M.n_T * (M.n_classes ? M.n_classes : 1);
  ctx->mrw_classes = M.n_classes;
  if ((rc = upload(ctx, chi, n_tab, &M.mrw_chi))) return rc;
  if ((rc = upload(ctx, kappa_dep, n_tab, &M.mrw_kdep))) return rc;
  if ((rc = upload(ctx, ext, n_tab, &M.mrw_ext))) return rc;
  if (!ctx->voro && (rc = upload(ctx, r_lim, (size_t)M.n_rad + 1, &M.r_lim))) return rc;
  if (M.l3D && !ctx->voro) {
    // sin_phi_lim, cos_phi_lim of the azimuthal walls (cylindrical_grid.f90:586-599, default-real phi; both grid types) for
    // distance_to_closest_wall_cyl's 3D branch (:1198-1218).  Where the reference stores the sentinel pair
    // (cos, sin) = (0, 1e300) for a wall at phi = pi/2 (mod pi) -- which makes that wall infinitely far for the walk -- the
    // true pair (0, 1) is used: |x sin - y cos| = |x| is the distance to that wall.
    std::vector<double> sp((size_t)M.n_az), cp((size_t)M.n_az);
    const float pi_sp = (float)3.14159265358979323846;
    const float delta_phi = 2.0f * pi_sp / (float)M.n_az;
    for (int k = 1; k <= M.n_az; ++k) {
      const float phi = delta_phi * (float)k;
      float md = fmodf(phi - 0.5f * pi_sp, pi_sp);
      if (md < 0.0f) md += pi_sp;
      if (fabsf(md) < 1.0e-6f) { cp[k - 1] = 0.0; sp[k - 1] = 1.0; }
      else { cp[k - 1] = (double)cosf(phi); sp[k - 1] = (double)sinf(phi); }
    }
    if ((rc = upload(ctx, sp.data(), sp.size(), &M.sin_phi)) || (rc = upload(ctx, cp.data(), cp.size(), &M.cos_phi))) return rc;
  }
  M.mrw_n_zeta = n_zeta; M.mrw_gamma = (float)gamma; M.mrw_n_inter = n_interactions; M.mrw = 1;
  M.mrw_exit_cdf = nullptr;   // (mcgpu_set_mrw_exit_spectrum, after this call)
  return MCGPU_OK;
}

// The spectrum a walk's last step leaves its sphere with: exit_cdf[class][n_T][n_lambda], cumulative over the wavelengths
// (non-decreasing rows ending in 1), or NULL for the cell's emission spectrum kdB_dT_CDF.
extern "C" int mcgpu_set_mrw_exit_spectrum(mcgpu_ctx* ctx, const double* exit_cdf) {
  if (!ctx) return MCGPU_ERR_ARG;
  DevModel& M = ctx->M;
  if (!M.mrw) return fail(ctx, MCGPU_ERR_STATE, "mcgpu_set_mrw_exit_spectrum: call mcgpu_set_mrw first");
  if (!exit_cdf) { M.mrw_exit_cdf = nullptr; return MCGPU_OK; }
  const size_t rows = (size_t)M.n_T * (M.n_classes ? M.n_classes : 1);
  for (size_t r = 0; r < rows; ++r) {
    const double* c = exit_cdf + r * M.n_lambda;
    for (int l = 1; l < M.n_lambda; ++l)
      if (!(c[l] >= c[l - 1])) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_mrw_exit_spectrum: a row decreases");
    if (!(c[0] >= 0.0) || !(c[M.n_lambda - 1] <= 1.0 + 1e-12)) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_mrw_exit_spectrum: a row leaves [0, 1]");
    // (a row that never reaches 1 -- all zero, or not normalised -- would send every draw above its end to the last wavelength)
    if (!(c[M.n_lambda - 1] >= 1.0 - 1e-12)) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_mrw_exit_spectrum: a row does not end in 1");
  }
  HIPCHK(hipSetDevice(ctx->device));
  return upload(ctx, exit_cdf, rows * (size_t)M.n_lambda, &M.mrw_exit_cdf);
}

extern "C" int mcgpu_set_ism(mcgpu_ctx* ctx, double R_ISM, const double* centre_ISM) {
  if (!ctx || !(R_ISM >= 0.0) || (R_ISM > 0.0 && !centre_ISM)) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_ism: bad argument");
  ctx->M.R_ISM = R_ISM;
  for (int q = 0; q < 3; ++q) ctx->M.centre_ISM[q] = centre_ISM ? centre_ISM[q] : 0.0;
  return MCGPU_OK;
}

extern "C" int mcgpu_set_sed_bins(mcgpu_ctx* ctx, int N_thet, int N_phi, int l_sym_centrale, int l_sym_axiale) {
  if (!ctx || N_thet < 1 || N_phi < 1) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_sed_bins: bad argument");
  ctx->M.N_thet = N_thet; ctx->M.N_phi = N_phi;
  ctx->M.sym_c = l_sym_centrale ? 1 : 0; ctx->M.sym_a = l_sym_axiale ? 1 : 0;
  ctx->have_sed = true;
  return MCGPU_OK;
}

static size_t n_sed(const DevModel& M) { return (size_t)MCGPU_N_SED_TYPES * M.n_lambda * M.N_thet * M.N_phi; }

static int ensure_accum(mcgpu_ctx* ctx) {
  const DevModel& M = ctx->M;
  const size_t n = (size_t)M.n_cells + n_sed(M) + M.n_lambda + MCGPU_N_COUNTERS;
  if (ctx->d_accum && ctx->n_accum == n) return MCGPU_OK;
  if (ctx->d_accum) hipFree(ctx->d_accum);
  ctx->d_accum = nullptr;
  HIPCHK(hipMalloc((void**)&ctx->d_accum, n * sizeof(double)));
  HIPCHK(hipMemset(ctx->d_accum, 0, n * sizeof(double)));
  ctx->n_accum = n;
  return MCGPU_OK;
}

static int ready(mcgpu_ctx* ctx) {
  if (!ctx) return MCGPU_ERR_ARG;
  if (!(ctx->have_grid && ctx->have_stars && ctx->have_opacity && ctx->have_scatt && ctx->have_thermal &&
        ctx->have_sed))
    return fail(ctx, MCGPU_ERR_STATE, "model incomplete: call every mcgpu_set_* first");
  if (ctx->reemission_pending)
    return fail(ctx, MCGPU_ERR_STATE, "the re-emission tables were left to mcgpu_init_reemission: call it first");
  return MCGPU_OK;
}

extern "C" int mcgpu_set_E_prior(mcgpu_ctx* ctx, const double* E_prior) {
  if (!ctx || !E_prior || !ctx->have_grid) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_E_prior: bad argument");
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->d_E_prior) HIPCHK(hipMalloc((void**)&ctx->d_E_prior, (size_t)ctx->M.n_cells * sizeof(double)));
  HIPCHK(hipMemcpy(ctx->d_E_prior, E_prior, (size_t)ctx->M.n_cells * sizeof(double), hipMemcpyHostToDevice));
  return MCGPU_OK;
}

// the kernels themselves live in the kern_*.hip translation units; mc_kernels.h hands out their handles

// The hand-over threshold of this launch.  The kernels that can hand packets over run their bulk ~8 % slower (more
// spilled registers: 160 against 116 bytes of scratch on the Pascucci instance), and the tail kernel only pays where
// packets get trapped: Pascucci's launch has no tail at all (T(N) linear through 5 ms), ref4.1's has 73 ms, a thick
// disk's seconds.  Automatic: from the model alone -- the radial optical depth of the midplane at the most opaque
// wavelength (> 1000) --, so that two launches on the same model follow the same schedule whatever ran before
// (round 3 also looked at the last launch's interactions per packet: a plan that depended on the call history).
static int tail_threshold(const mcgpu_ctx* ctx) {
  if (ctx->opt_tail >= 0) return ctx->opt_tail;
  return ctx->tau_midplane > 1000.0 ? 48 : 0;
}

// the record buffers a role kernel hands unfinished packets over in (chunks of a binned run: two, used in turns; the
// tail kernel: one): per workgroup its records, a packet per lane and a batch of work items per wave
static int carry_prepare(mcgpu_ctx* ctx, int n_wg, int n_rec, int threads) {
  const size_t cap = (size_t)n_wg * ((size_t)n_rec + threads + (size_t)(threads / 64) * PK_BATCH);
  if (ctx->carry_cap < cap || !ctx->d_carry[0]) {
    for (int i = 0; i < 2; ++i) { if (ctx->d_carry[i]) hipFree(ctx->d_carry[i]); ctx->d_carry[i] = nullptr; }
    for (int i = 0; i < 2; ++i) HIPCHK(hipMalloc(&ctx->d_carry[i], cap * sizeof(Rec<true>)));
    ctx->carry_cap = cap;
  }
  if (!ctx->d_carry_n) HIPCHK(hipMalloc((void**)&ctx->d_carry_n, 2 * sizeof(unsigned int)));
  if (!ctx->d_tail_next) HIPCHK(hipMalloc((void**)&ctx->d_tail_next, sizeof(unsigned int)));
  return MCGPU_OK;
}

// ---- the tail's last packets on the host (mc_tail.hip.h "The last packets on the host"; host_tail.cpp) ----------------
// One launch's hand-over: the host copies of the model and of the launch's arguments the host threads run on.  Heap
// object, made by launch_tail, consumed and freed by the stream's callback.
struct HostTailCall {
  mcgpu_ctx* ctx;
  DevModel Mh;
  RunArgs Ah;
  mcgpu_host::TailJob job;
  const unsigned int* h_ctl;   // [1]: records k_tail wrote
  int* h_err;
};

static void host_tail_callback(void* p) {   // (runs on a thread of the HIP runtime, in stream order: no HIP calls here)
  HostTailCall* c = static_cast<HostTailCall*>(p);
  mcgpu_ctx* ctx = c->ctx;
  unsigned int n = c->h_ctl[1];
  if (n > ctx->tail_out_cap) n = ctx->tail_out_cap;   // (k_tail reported error 17)
  c->job.n = (*c->h_err == 0) ? n : 0u;
  mcgpu_host::run_tail(&c->job);
  ctx->host_tail_ms = c->job.ms; ctx->host_tail_packets = c->job.n; ctx->host_tail_threads = c->job.threads_used;
  ctx->host_tail_events = c->job.events;
  delete c;
}

static inline size_t arena_align(size_t x) { return (x + 63) & ~(size_t)63; }

// Can the host finish this launch's last packets?  (One dust class, no radiation-field extras -- what k_tail itself runs --
// and an emission table small enough to copy per launch.)
static bool host_tail_applicable(const mcgpu_ctx* ctx, const RunArgs& A) {
  const DevModel& M = ctx->M;
  if (ctx->opt_tail_where == 1) return false;
  if (M.n_classes || M.m1 || M.grid_sph || ctx->voro || A.xN_abs || A.xJ_abs) return false;
  if (M.prob_E_cell && (size_t)(M.n_cells + 1) * M.n_lambda * sizeof(double) > ((size_t)64 << 20)) return false;
  return true;
}

// k_tail (mc_tail.hip.h) on the packets in `carry`: one packet per wave, as many 256-thread workgroups as the tables'
// LDS footprint lets reside.  With the host behind it (option "tail_where"): k_tail leaves the last packets, and the
// stream continues with [copies to the host | the host threads, as a stream callback | copies back] -- asynchronous like
// every launch.
static int launch_tail(mcgpu_ctx* ctx, const RunArgs& A_in, const void* carry, const unsigned int* carry_n, bool l3d, bool mrw) {
  const DevModel& M = ctx->M;
  RunArgs A = A_in;
  const bool pola = ctx->lsepar_pola != 0, dark = M.dark != nullptr;
  const size_t lds = (lds_bytes(M) + 7) / 8 * 8;
  int per_cu = (int)((160 * 1024 - 512) / (lds ? lds : 1));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  const int blocks = ctx->prop.multiProcessorCount * per_cu;
  const void* fn = kpick_tail(l3d, pola, dark, mrw);
  HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  HIPCHK(hipMemsetAsync(ctx->d_tail_next, 0, sizeof(unsigned int), ctx->stream));
  const bool host = host_tail_applicable(ctx, A);
  ctx->tail_on_host = host;
  int n_threads = 0;
  unsigned int host_max = 0u;
  if (host) {
    int n_dev = 1;
    (void)hipGetDeviceCount(&n_dev);
    n_threads = ctx->opt_host_threads > 0 ? ctx->opt_host_threads : mcgpu_host::default_threads(n_dev);
    mcgpu_host::prepare_threads(n_threads);
    host_max = ctx->opt_tail_host_max > 0 ? (unsigned int)ctx->opt_tail_host_max : 8u * (unsigned int)n_threads;
    if (!ctx->d_tail_ctl) HIPCHK(hipMalloc((void**)&ctx->d_tail_ctl, 2 * sizeof(unsigned int)));
    if (ctx->tail_out_cap < host_max) {
      HIPCHK(hipStreamSynchronize(ctx->stream));   // (an earlier launch may still write the old buffer)
      if (ctx->d_tail_out) hipFree(ctx->d_tail_out);
      ctx->d_tail_out = nullptr; ctx->tail_out_cap = 0;
      HIPCHK(hipMalloc(&ctx->d_tail_out, (size_t)host_max * sizeof(Rec<true>)));
      ctx->tail_out_cap = host_max;
    }
    if (!ctx->side_stream) {
      HIPCHK(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&ctx->ev_side_in, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&ctx->ev_side_out, hipEventDisableTiming));
    }
    HIPCHK(hipMemsetAsync(ctx->d_tail_ctl, 0, 2 * sizeof(unsigned int), ctx->stream));
    A.tail_host_max = host_max; A.tail_done = ctx->d_tail_ctl; A.tail_out = ctx->d_tail_out; A.tail_out_n = ctx->d_tail_ctl + 1;
  }
  HIPCHK(hipEventRecord(ctx->ev_tail, ctx->stream));   // (what follows is the launch's tail: mcgpu_get_info "tail_ms")
  ctx->tail_launched = true;
  void* args[] = {(void*)&M, (void*)&A, (void*)&carry, (void*)&carry_n, (void*)&ctx->d_tail_next};
  HIPCHK(hipLaunchKernel(fn, dim3(blocks), dim3(MCGPU_TAIL_BLOCK), args, lds, ctx->stream));
  if (!host) return MCGPU_OK;

  // ---- the host's copies: one pinned arena, laid out per launch ---------------------------------------------------------
  HostTailCall* call = new HostTailCall();
  call->ctx = ctx;
  call->Mh = M;
  call->Ah = A;
  struct Seg { const void* dev; size_t off, bytes; };
  std::vector<Seg> segs;
  size_t total = 0;
  std::vector<std::pair<const void**, size_t>> fix;   // pointer slots of Mh / Ah and their arena offsets
  auto table = [&](const void** slot, size_t bytes) {
    if (!*slot) return;
    segs.push_back(Seg{*slot, total, bytes});
    fix.push_back({slot, total});
    total += arena_align(bytes);
  };
#define HT_TAB(field, count) table((const void**)&call->Mh.field, (size_t)(count) * sizeof(*call->Mh.field))
  DevModel& H = call->Mh;
  HT_TAB(r_lim_2, H.n_rad + 1); HT_TAB(zmax, H.n_rad); HT_TAB(ch, H.n_rad); HT_TAB(tan_phi_lim, H.n_az);
  HT_TAB(volume, H.n_cells); HT_TAB(star_xyzr, 4 * H.n_stars); HT_TAB(star_cell, 4 * H.n_stars);
  HT_TAB(kappa, H.n_lambda); HT_TAB(kappa_abs, H.n_lambda); HT_TAB(albedo, H.n_lambda);
  HT_TAB(kappa_factor, (size_t)H.n_cells + 1); HT_TAB(dark, H.n_cells);
  const size_t nt = (size_t)(H.nang + 1) * H.n_lambda;
  HT_TAB(prob_s11, nt); HT_TAB(s12, nt); HT_TAB(s22, nt); HT_TAB(s33, nt); HT_TAB(s34, nt); HT_TAB(s44, nt);
  HT_TAB(tab_g, H.n_lambda); HT_TAB(cos_tab, H.nang + 1);
  HT_TAB(log_Qcool, H.n_T); HT_TAB(cdf, (size_t)H.n_T * H.n_lambda); HT_TAB(spec_cum, H.n_lambda + 1);
  HT_TAB(frac_E_stars, H.n_lambda); HT_TAB(frac_E_disk, H.n_lambda); HT_TAB(CDF_E_star, (size_t)H.n_lambda * (H.n_stars + 1));
  HT_TAB(prob_E_cell, (size_t)(H.n_cells + 1) * H.n_lambda);
  if (H.mrw) {
    HT_TAB(mrw_zeta, H.mrw_n_zeta); HT_TAB(mrw_chi, H.n_T); HT_TAB(mrw_kdep, H.n_T); HT_TAB(mrw_ext, H.n_T);
    HT_TAB(mrw_guide, MRW_GUIDE + 1); HT_TAB(mrw_exit_cdf, (size_t)H.n_T * H.n_lambda); HT_TAB(r_lim, H.n_rad + 1);
    HT_TAB(sin_phi, H.n_az); HT_TAB(cos_phi, H.n_az);
  } else {
    H.mrw_zeta = H.mrw_chi = H.mrw_kdep = H.mrw_ext = H.mrw_exit_cdf = H.r_lim = H.sin_phi = H.cos_phi = nullptr; H.mrw_guide = nullptr;
  }
#undef HT_TAB
  // (tables of paths k_tail never runs: a stray access must fault, not read device memory)
  H.tan_theta_lim = H.theta_lim = H.r_lim_3 = nullptr; H.cell_class = nullptr;
  H.v_kappa = H.v_kabs = H.v_lq = H.v_cdf = nullptr; H.v_albedo = nullptr; H.v_kk = nullptr;
  H.v_prob = H.v_g = H.v_s11 = H.v_s12 = H.v_s22 = H.v_s33 = H.v_s34 = H.v_s44 = nullptr;
  const size_t n_tables = segs.size();
  RunArgs& Ha = call->Ah;
  if (Ha.frozen) table((const void**)&Ha.E_prior, (size_t)H.n_cells * sizeof(double)); else Ha.E_prior = nullptr;
  const size_t n_side = segs.size();   // (tables and the prior: constant during the launch -> the side stream)
  const size_t off_accum = total; total += arena_align(ctx->n_accum * sizeof(double));
  const size_t off_cnt = total; total += arena_align(CNT_SLOTS * sizeof(unsigned long long));
  const size_t off_err = total; total += arena_align(sizeof(int));
  const size_t off_ctl = total; total += arena_align(2 * sizeof(unsigned int));
  const size_t rec_bytes = pola ? sizeof(Rec<true>) : sizeof(Rec<false>);
  const size_t off_rec = total; total += arena_align((size_t)host_max * rec_bytes);
  (void)n_tables;
  if (ctx->h_arena_bytes < total) {
    hipError_t e = hipStreamSynchronize(ctx->stream);   // (an earlier launch's callback may still read the old arena)
    if (e == hipSuccess && ctx->h_arena) e = hipHostFree(ctx->h_arena);
    ctx->h_arena = nullptr; ctx->h_arena_bytes = 0;
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->h_arena, total + (total >> 2), hipHostMallocDefault);
    if (e != hipSuccess) { delete call; ctx->err = std::string("host tail: pinned arena: ") + hipGetErrorString(e); return MCGPU_ERR_HIP; }
    ctx->h_arena_bytes = total + (total >> 2);
  }
  char* const base = ctx->h_arena;
  for (auto& f : fix) *f.first = base + f.second;
  Ha.E_abs = reinterpret_cast<double*>(base + off_accum);
  Ha.sed = Ha.E_abs + M.n_cells;
  Ha.n_sent = Ha.sed + n_sed(M);
  Ha.counters = reinterpret_cast<unsigned long long*>(base + off_cnt);
  Ha.next_packet = nullptr;
  Ha.err = reinterpret_cast<int*>(base + off_err);
  Ha.xN_abs = nullptr; Ha.xJ_abs = nullptr;
  Ha.carry_in = nullptr; Ha.carry_in_n = nullptr; Ha.carry_out = nullptr; Ha.carry_out_n = nullptr;
  Ha.tail_host_max = 0u; Ha.tail_done = nullptr; Ha.tail_out = nullptr; Ha.tail_out_n = nullptr;
  std::memset(&Ha.bin, 0, sizeof(Ha.bin));
  call->h_ctl = reinterpret_cast<const unsigned int*>(base + off_ctl);
  call->h_err = Ha.err;
  call->job.model = &call->Mh; call->job.args = &call->Ah; call->job.recs = base + off_rec; call->job.n = 0;
  call->job.l3d = l3d ? 1 : 0; call->job.pola = pola ? 1 : 0; call->job.dark = dark ? 1 : 0; call->job.mrw = mrw ? 1 : 0;
  call->job.n_threads = n_threads;
#define HT_CHK(callexpr) do { hipError_t e_ = (callexpr); if (e_ != hipSuccess) { delete call; ctx->err = std::string(#callexpr) + ": " + hipGetErrorString(e_); return MCGPU_ERR_HIP; } } while (0)
  // the side stream copies the tables while k_tail runs (it starts where the stream stood in front of k_tail: the previous
  // launch's callback has finished with the arena by then)
  HT_CHK(hipStreamWaitEvent(ctx->side_stream, ctx->ev_tail, 0));
  for (size_t i = 0; i < n_side; ++i)
    HT_CHK(hipMemcpyAsync(base + segs[i].off, segs[i].dev, segs[i].bytes, hipMemcpyDeviceToHost, ctx->side_stream));
  HT_CHK(hipEventRecord(ctx->ev_side_out, ctx->side_stream));
  // behind k_tail: what it left, and the sums the host adds to
  HT_CHK(hipMemcpyAsync(base + off_ctl, ctx->d_tail_ctl, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
  HT_CHK(hipMemcpyAsync(base + off_rec, ctx->d_tail_out, (size_t)host_max * rec_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HT_CHK(hipMemcpyAsync(base + off_accum, ctx->d_accum, ctx->n_accum * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HT_CHK(hipMemcpyAsync(base + off_cnt, ctx->d_counters, CNT_SLOTS * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  HT_CHK(hipMemcpyAsync(base + off_err, ctx->d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HT_CHK(hipStreamWaitEvent(ctx->stream, ctx->ev_side_out, 0));
  HT_CHK(hipLaunchHostFunc(ctx->stream, host_tail_callback, call));
  // (from here on the callback owns `call`)
  HIPCHK(hipMemcpyAsync(ctx->d_accum, base + off_accum, ctx->n_accum * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_counters, base + off_cnt, CNT_SLOTS * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_err, base + off_err, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
#undef HT_CHK
  return MCGPU_OK;
}

// the persistent packet kernel of the cylindrical grids (mc_roles.hip.h, mc_device.hip.h)
static int launch_mega(mcgpu_ctx* ctx, const RunArgs& A, bool use_lds, int grid_blocks, int block_threads) {
  const DevModel& M = ctx->M;
  const size_t lds = lds_bytes(M);
  const size_t lds_cap = 160 * 1024;
  const size_t lds_k = use_lds ? lds + (size_t)M.n_cells * sizeof(double) : lds;
  const int max_threads = use_lds ? MCGPU_LDS_BLOCK : 256;
  const int threads = (block_threads > 0 && block_threads <= max_threads) ? block_threads : max_threads;
  if (threads % 64) return fail(ctx, MCGPU_ERR_ARG, "block_threads must be a multiple of 64");
  int blocks = grid_blocks;
  if (blocks <= 0) {
    // persistent grid: as many workgroups as the LDS footprint lets reside
    int per_cu = (int)(lds_cap / (lds_k > 0 ? lds_k : 1));
    const int cap = 2048 / threads;  // waves
    if (per_cu < 1) per_cu = 1;
    if (per_cu > cap) per_cu = cap;
    blocks = ctx->prop.multiProcessorCount * per_cu;
    const unsigned long long need = (A.n_packets + threads - 1) / threads;
    if ((unsigned long long)blocks > need) blocks = (int)(need ? need : 1);
  }
  const bool pola = ctx->lsepar_pola != 0, dark = M.dark != nullptr, l3d = M.l3D != 0;
  hipError_t e;
  // Waves with roles and LDS packet queues (mc_roles.hip.h): the default wherever the queues fit next to the
  // tables.  MCGPU_ROLES: -1 = the single-role kernel (thermal_body), 0..7 = that
  // many fixed flyer waves, 100+f = every wave picks its role per round (flyer when f lanes can fly), 200 = per
  // round, the role in which more of its lanes have work (default).
  if (M.mrw && ctx->mrw_classes != M.n_classes)
    return fail(ctx, MCGPU_ERR_STATE, "the random walk's tables belong to another set of dust classes: mcgpu_set_mrw after mcgpu_set_variable_dust");
  if (M.grid_sph && (M.n_classes || dark)) {  // the spherical grid with a dark zone and / or dust classes (round 5)
    if (M.mrw) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "spherical grid: the random walk runs without dark zone and dust classes");
    if (M.n_classes && dark) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "spherical grid: a dark zone or dust classes, not both (unverified: the CPU restatement has no such case)");
    const void* fn = kpick_thermal_sph_ext(l3d, pola, dark, use_lds, M.n_classes != 0);
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k));
    void* args[] = {(void*)&M, (void*)&A};
    HIPCHK(hipLaunchKernel(fn, dim3(blocks), dim3(threads), args, lds_k, ctx->stream));
    return MCGPU_OK;
  }
  if (M.n_classes) {  // lvariable_dust
    const void* fn;
    // the role schedule (k_thermal_roles_var) wherever its records fit; option "schedule" = 1 or the radiation-field
    // extras: the HBM-gather variant of the single-role kernel below
    {
      const int rthreads = (block_threads > 0 && block_threads <= MCGPU_ROLES_BLOCK) ? block_threads : MCGPU_ROLES_BLOCK;
      if (rthreads % 64) return fail(ctx, MCGPU_ERR_ARG, "block_threads must be a multiple of 64");
      const size_t lds_t = (lds_k + 7) / 8 * 8;
      int n_rec = lds_t < lds_cap ? rq_records_that_fit(pola, lds_cap - lds_t) : 0;
      if (n_rec > 2 * rthreads) n_rec = 2 * rthreads > RQ_MIN_REC ? 2 * rthreads : RQ_MIN_REC;
      if (ctx->opt_schedule != 1 && !A.xN_abs && !A.xJ_abs && n_rec > 0 && !M.mrw) {  // (the walk: single-role kernel)
        if (!ctx->vkk_valid) {  // (kappa kappa_factor, kappa_abs_LTE) per (cell, wavelength): what a flight reads per cell
          const size_t n = ((size_t)M.n_cells + 1) * M.n_lambda;
          if (ctx->d_vkk) hipFree(ctx->d_vkk);
          ctx->d_vkk = nullptr;
          HIPCHK(hipMalloc((void**)&ctx->d_vkk, n * sizeof(double2)));
          hipLaunchKernelGGL(k_build_vkk, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, M, ctx->d_vkk);
          HIPCHK(hipGetLastError());
          ctx->M.v_kk = ctx->d_vkk;
          ctx->vkk_valid = true;
        }
        DevModel Mv = ctx->M;
        const size_t lds_r = lds_t + rq_lds_bytes(pola, n_rec);
        int rblocks = grid_blocks > 0 ? grid_blocks : ctx->prop.multiProcessorCount;
        const unsigned long long need = (A.n_packets + rthreads - 1) / rthreads;
        if (grid_blocks <= 0 && (unsigned long long)rblocks > need) rblocks = (int)(need ? need : 1);
        int n_srv_pref = (rthreads / 64 + 3) / 4, k_short = 2, fly_iters = 16, fly_idle = 32, emit_qmax = 128;
        fn = kpick_roles_var(l3d, pola, dark, use_lds);
        HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
        void* rargs[] = {(void*)&Mv, (void*)&A, (void*)&n_rec, (void*)&n_srv_pref, (void*)&k_short, (void*)&fly_iters, (void*)&fly_idle, (void*)&emit_qmax};
        HIPCHK(hipLaunchKernel(fn, dim3(rblocks), dim3(rthreads), rargs, lds_r, ctx->stream));
        return MCGPU_OK;
      }
    }
    fn = kpick_thermal_var(l3d, pola, dark, use_lds, M.mrw != 0);
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k));
    void* args[] = {(void*)&M, (void*)&A};
    HIPCHK(hipLaunchKernel(fn, dim3(blocks), dim3(threads), args, lds_k, ctx->stream));
    return MCGPU_OK;
  }
  if (M.grid_sph) {  // the spherical grid runs the single-role kernel with its own grid operators
    const size_t lds_k2 = lds_k;
    const void* fn = kpick_thermal_sph(l3d, pola, use_lds, M.mrw != 0);
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k2));
    void* args[] = {(void*)&M, (void*)&A};
    HIPCHK(hipLaunchKernel(fn, dim3(blocks), dim3(threads), args, lds_k2, ctx->stream));
    return MCGPU_OK;
  }
  // Waves with roles and LDS packet records (mc_roles.hip.h): the default wherever enough records fit next to the
  // tables; otherwise (or with option "schedule" = 1) the single-role kernel below.
  {
    const int rthreads = (block_threads > 0 && block_threads <= MCGPU_ROLES_BLOCK) ? block_threads : MCGPU_ROLES_BLOCK;
    if (rthreads % 64) return fail(ctx, MCGPU_ERR_ARG, "block_threads must be a multiple of 64");
    const size_t lds_t = (lds_k + 7) / 8 * 8;
    // (the kernels that hand packets over hold 256 bytes of static LDS: a request of the full 160 KB is refused there)
    const int tail_thr = tail_threshold(ctx);
    const size_t lds_cap_r = lds_cap - ((!l3d && tail_thr > 0) ? 512 : 0);
    int n_rec = lds_t < lds_cap_r ? rq_records_that_fit(pola, lds_cap_r - lds_t) : 0;
    // (more records than twice the lanes buy nothing; small models keep their LDS footprint small)
    if (n_rec > 2 * rthreads) n_rec = 2 * rthreads > RQ_MIN_REC ? 2 * rthreads : RQ_MIN_REC;
    // (the optional radiation-field accumulators are kept by the single-role kernel)
    // (the random walk on a 3D grid runs in the single-role kernel: the role schedule's walk is 2D)
    if (tune("MCGPU_ROLES", (ctx->opt_schedule == 1 || A.xN_abs || A.xJ_abs) ? 0 : 1, 0, 1) && n_rec > 0 && !(M.mrw && l3d)) {
      const size_t lds_r = lds_t + rq_lds_bytes(pola, n_rec);
      int rblocks = grid_blocks > 0 ? grid_blocks : ctx->prop.multiProcessorCount;
      {
        const unsigned long long need = (A.n_packets + rthreads - 1) / rthreads;
        if (grid_blocks <= 0 && (unsigned long long)rblocks > need) rblocks = (int)(need ? need : 1);
      }
      int n_srv_pref = tune("MCGPU_N_SRV", (rthreads / 64 + 3) / 4, 1, 1016);  // serving waves to start with (adaptive; 1000 + n: fixed)
      // crossings between two visits of the rings: 16 where packets interact (ref4.1: 247 / 252 / 257 ms per 1e8 packets
      // at 16 / 24 / 32), 32 on thin models whose flights are a packet's whole life (Pascucci: 131.7 / 128.7 / 128.6 / 131.4
      // ms at 16 / 24 / 32 / 48) -- told apart like the tail hand-over, by the model's midplane optical depth
      int k_short = tune("MCGPU_K_SHORT", 2, 0, 64), fly_iters = tune("MCGPU_FLY_ITERS", ctx->tau_midplane > 1000.0 ? 16 : 32, 1, 256);
      int fly_idle = tune("MCGPU_FLY_IDLE", 32, 1, 65), emit_qmax = tune("MCGPU_EMIT_QMAX", 128, 0, 1 << 20);
      const void* fn = kpick_roles(l3d, pola, dark, use_lds, M.mrw != 0);   // (the walk: 2D; 3D was sent to the single-role kernel above)
      // 2D grids: the launch's last packets go to the tail kernel (one packet per wave, mc_tail.hip.h) once a
      // workgroup has no more than opt_tail of them left
      RunArgs At = A;
      const bool tail = !l3d && tail_thr > 0;
      if (tail) {
        int rc = carry_prepare(ctx, rblocks, n_rec, rthreads);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(ctx->d_carry_n, 0, 2 * sizeof(unsigned int), ctx->stream));
        At.carry_out = ctx->d_carry[0]; At.carry_out_n = ctx->d_carry_n; At.carry_cap = (unsigned int)ctx->carry_cap;
        At.tail_threshold = tail_thr;
        fn = kpick_roles_tail(pola, dark, use_lds, M.mrw != 0);
      }
      // option "crossing" = 1: the flying waves cross with the flight-parametric form (fly_step_2d_param, mc_roles.hip.h) --
      // not the reference's arithmetic (statistical parity only), so never by default; 2D, LDS deposits, no dark zone, no walk
      if (ctx->opt_crossing == 1 && !l3d && !dark && use_lds && !M.mrw) fn = kpick_roles_param(pola, tail);
      HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
      void* args[] = {(void*)&M, (void*)&At, (void*)&n_rec, (void*)&n_srv_pref, (void*)&k_short, (void*)&fly_iters, (void*)&fly_idle, (void*)&emit_qmax};
      HIPCHK(hipLaunchKernel(fn, dim3(rblocks), dim3(rthreads), args, lds_r, ctx->stream));
      if (tail) return launch_tail(ctx, A, ctx->d_carry[0], ctx->d_carry_n, false, M.mrw != 0);
      return MCGPU_OK;
    }
  }
  {
    const void* kern = kpick_thermal(use_lds, l3d, pola, dark, M.mrw != 0);
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k);
    void* args[] = {(void*)&M, (void*)&A};
    if (e == hipSuccess) e = hipLaunchKernel(kern, dim3(blocks), dim3(threads), args, lds_k, ctx->stream);
  }
  if (e != hipSuccess) {
    ctx->err = std::string("kernel launch: ") + hipGetErrorString(e);
    return MCGPU_ERR_HIP;
  }
  return MCGPU_OK;
}

// the Voronoi-grid kernel (mc_voronoi.hip.h).  Default: one 1024-thread workgroup per CU with a
// hashed deposit cache in the LDS left over by the tables; MCGPU_DEPOSIT=hbm: 256-thread
// workgroups depositing straight to HBM.
static int launch_voro(mcgpu_ctx* ctx, const RunArgs& A, int grid_blocks, int block_threads) {
  const DevModel& M = ctx->M;
  const size_t lds_t = (lds_bytes(M) + 7) / 8 * 8;
  const size_t lds_cap = 160 * 1024;
  // Option "schedule" = 2: the role schedule of mc_roles.hip.h (serving and flying waves over packet records in LDS)
  // with the deposit cache in what the tables and at least 256 records leave.  Not the default on this grid:
  // measured at 100 000 sites 2.5e7 packets/s against 4.7e7 for the single-role kernel below -- the stand-in disk's
  // packets interact 119 times for 162 crossings, so almost all work is serving work, and the serving lanes are
  // limited by the records that fit into LDS (512 for 1024 lanes).
  if (M.n_classes) {  // lvariable_dust (what a multi-grain SPH dump gives: p_n_cells = n_cells): the single-role kernel
    const bool pola = ctx->lsepar_pola != 0;   // with the deposit cache, the class's tables from HBM
    int log_ns = ctx->opt_cache_log_slots;
    while (log_ns > 6 && lds_t + ((size_t)12 << log_ns) > lds_cap) --log_ns;
    if (lds_t + ((size_t)12 << log_ns) > lds_cap) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "wavelength tables exceed the LDS of one CU");
    const size_t lds = lds_t + ((size_t)12 << log_ns);
    const int threads = (block_threads > 0 && block_threads <= 768) ? block_threads : 768;
    if (threads % 64) return fail(ctx, MCGPU_ERR_ARG, "block_threads must be a multiple of 64");
    const void* fn = kpick_voro_var(pola, M.mrw != 0);
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int blocks = grid_blocks;
    if (blocks <= 0) {
      int occ = 1;
      HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, threads, lds));
      if (occ < 1) occ = 1;
      blocks = ctx->prop.multiProcessorCount * occ;
      const unsigned long long need = (A.n_packets + threads - 1) / threads;
      if ((unsigned long long)blocks > need) blocks = (int)(need ? need : 1);
    }
    void* args[] = {(void*)&M, (void*)&A, (void*)&ctx->V, (void*)&log_ns};
    HIPCHK(hipLaunchKernel(fn, dim3(blocks), dim3(threads), args, lds, ctx->stream));
    return MCGPU_OK;
  }
  // The pool schedule (mc_voronoi_pool.hip.h; option "schedule" = 3): packet records in HBM / L2, queues by phase and by
  // neighbour-list length in LDS, one phase per wave pass.  For the plain thermal step: one dust class, no random walk, no
  // radiation-field extras, cached deposits.  Opt-in: measured at 1e6 sites it runs 0.55 of its lanes (0.29 for the kernel
  // below) on 30 % fewer vector instructions, and is 0.75x as fast -- its scattered 16-byte record and neighbour loads keep
  // the CU's address unit 80 % busy (profiles/r05_voro_pool_*).
  const bool pool = ctx->opt_schedule == 3 && !M.mrw && !M.m1 && ctx->opt_deposit != 1 && !A.xN_abs && !A.xJ_abs;
  if (pool) {
    const bool pola = ctx->lsepar_pola != 0;
    int log_rec = ctx->opt_pool_log_rec;
    int log_ns = ctx->opt_cache_log_slots < 12 ? ctx->opt_cache_log_slots : 12;   // (the queues want the LDS more than the cache does)
    auto lds_of = [&](int lr, int ln) { return lds_t + (((size_t)12 << ln) + 7) / 8 * 8 + vp_lds_bytes(lr); };
    while (lds_of(log_rec, log_ns) > lds_cap && log_ns > 9) --log_ns;
    while (lds_of(log_rec, log_ns) > lds_cap && log_rec > 8) --log_rec;
    while (lds_of(log_rec, log_ns) > lds_cap && log_ns > 6) --log_ns;
    if (lds_of(log_rec, log_ns) <= lds_cap) {
      const size_t lds_p = lds_of(log_rec, log_ns);
      const int pthreads = (block_threads > 0 && block_threads <= 1024) ? block_threads : 1024;
      if (pthreads % 64) return fail(ctx, MCGPU_ERR_ARG, "block_threads must be a multiple of 64");
      int pblocks = grid_blocks > 0 ? grid_blocks : ctx->prop.multiProcessorCount;
      const unsigned long long need = (A.n_packets + pthreads - 1) / pthreads;
      if (grid_blocks <= 0 && (unsigned long long)pblocks > need) pblocks = (int)(need ? need : 1);
      const size_t want = (size_t)pblocks * ((size_t)sizeof(PRec) << log_rec);
      if (ctx->pool_bytes < want) {
        if (ctx->d_pool) hipFree(ctx->d_pool);
        ctx->d_pool = nullptr; ctx->pool_bytes = 0;
        HIPCHK(hipMalloc(&ctx->d_pool, want));
        ctx->pool_bytes = want;
      }
      PoolArgs PA;
      PA.recs = reinterpret_cast<PRec*>(ctx->d_pool); PA.log_rec = log_rec; PA.cache_log_ns = log_ns;
      if (!ctx->d_pool_blob) HIPCHK(hipMalloc((void**)&ctx->d_pool_blob, sizeof(VpBlob)));
      ctx->h_pool_blob.M = M; ctx->h_pool_blob.A = A; ctx->h_pool_blob.G = ctx->V;
      // (pageable host memory: a synchronous copy -- behind the stream's earlier work -- so that a second launch enqueued
      // right after this one cannot overwrite the blob before it has been read)
      HIPCHK(hipStreamSynchronize(ctx->stream));
      HIPCHK(hipMemcpy(ctx->d_pool_blob, &ctx->h_pool_blob, sizeof(VpBlob), hipMemcpyHostToDevice));
      const VpBlob* blob = ctx->d_pool_blob;
      const void* fn = kpick_voro_pool(pola, pthreads);
      HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
      void* args[] = {(void*)&M, (void*)&A, (void*)&ctx->V, (void*)&PA, (void*)&blob};
      HIPCHK(hipLaunchKernel(fn, dim3(pblocks), dim3(pthreads), args, lds_p, ctx->stream));
      return MCGPU_OK;
    }
  }
  const bool voro_roles = ctx->opt_schedule == 2 && ctx->opt_deposit != 1 && !A.xN_abs && !A.xJ_abs;
  if (voro_roles) {
    const bool pola = ctx->lsepar_pola != 0;
    int log_ns = ctx->opt_cache_log_slots, n_rec = 0;
    for (; log_ns >= 6; --log_ns) {
      const size_t used = lds_t + (((size_t)12 << log_ns) + 7) / 8 * 8;
      n_rec = used < lds_cap ? rq_records_that_fit(pola, lds_cap - used) : 0;
      if (n_rec >= 256) break;
    }
    if (log_ns >= 6 && n_rec >= 256) {
      const int rthreads = (block_threads > 0 && block_threads <= MCGPU_ROLES_BLOCK) ? block_threads : MCGPU_ROLES_BLOCK;
      if (rthreads % 64) return fail(ctx, MCGPU_ERR_ARG, "block_threads must be a multiple of 64");
      if (n_rec > RQ_CAP) n_rec = RQ_CAP;
      const size_t lds_r = lds_t + (((size_t)12 << log_ns) + 7) / 8 * 8 + rq_lds_bytes(pola, n_rec);
      int rblocks = grid_blocks > 0 ? grid_blocks : ctx->prop.multiProcessorCount;
      const unsigned long long need = (A.n_packets + rthreads - 1) / rthreads;
      if (grid_blocks <= 0 && (unsigned long long)rblocks > need) rblocks = (int)(need ? need : 1);
      int n_srv_pref = tune("MCGPU_N_SRV", (rthreads / 64 + 1) / 2, 1, 1016);  // (this grid's packets interact as often as they cross)
      int k_short = tune("MCGPU_K_SHORT", 2, 0, 64), fly_iters = tune("MCGPU_FLY_ITERS", 16, 1, 256);
      int fly_idle = tune("MCGPU_FLY_IDLE", 32, 1, 65), emit_qmax = tune("MCGPU_EMIT_QMAX", 128, 0, 1 << 20);
      const void* fn = kpick_voro_roles(pola, M.mrw != 0);
      HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
      void* args[] = {(void*)&M, (void*)&A, (void*)&ctx->V, (void*)&log_ns, (void*)&n_rec, (void*)&n_srv_pref, (void*)&k_short,
                      (void*)&fly_iters, (void*)&fly_idle, (void*)&emit_qmax};
      HIPCHK(hipLaunchKernel(fn, dim3(rblocks), dim3(rthreads), args, lds_r, ctx->stream));
      return MCGPU_OK;
    }
  }
  if (M.mrw) {  // the random walk: the single-role kernel with HBM deposits (schedule 2: the role kernel above, where it fits)
    const bool pola = ctx->lsepar_pola != 0;
    const int threads = (block_threads > 0 && block_threads <= 256) ? block_threads : 256;
    if (threads % 64) return fail(ctx, MCGPU_ERR_ARG, "block_threads must be a multiple of 64");
    const void* fn = kpick_voro_mrw(pola);
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
    int blocks = grid_blocks;
    if (blocks <= 0) {
      int occ = 1;
      HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, threads, lds_t));
      if (occ < 1) occ = 1;
      blocks = ctx->prop.multiProcessorCount * occ;
      const unsigned long long need = (A.n_packets + threads - 1) / threads;
      if ((unsigned long long)blocks > need) blocks = (int)(need ? need : 1);
    }
    void* args[] = {(void*)&M, (void*)&A, (void*)&ctx->V};
    HIPCHK(hipLaunchKernel(fn, dim3(blocks), dim3(threads), args, lds_t, ctx->stream));
    return MCGPU_OK;
  }
  bool cache = ctx->opt_deposit != 1;
  int log_ns = 0;
  if (cache) {
    log_ns = ctx->opt_cache_log_slots;
    while (log_ns > 6 && lds_t + ((size_t)12 << log_ns) > lds_cap) --log_ns;
    if (lds_t + ((size_t)12 << log_ns) > lds_cap) cache = false;
  }
  const size_t lds = cache ? lds_t + ((size_t)12 << log_ns) : lds_t;
  const int max_threads = cache ? VORO_CACHE_BLOCK : 256;
  // default 768 threads = 3 waves/SIMD at 168 VGPRs and no scratch; the 1024-thread build (4 waves/SIMD at 128 VGPRs,
  // 51-66 spilled) measured 3 % faster at most, within the run-to-run noise
  const int threads = (block_threads > 0 && block_threads <= max_threads) ? block_threads : (cache ? 768 : max_threads);
  if (threads % 64) return fail(ctx, MCGPU_ERR_ARG, "block_threads must be a multiple of 64");
  const bool pola = ctx->lsepar_pola != 0;
  // (which register budget the workgroup was compiled for: 512 / 768 / 1024 threads)
  const void* fn = cache ? kpick_voro_cache(pola, threads) : kpick_voro(pola);
  HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int blocks = grid_blocks;
  if (blocks <= 0) {
    int occ = 1;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, threads, lds));
    if (occ < 1) occ = 1;
    blocks = ctx->prop.multiProcessorCount * occ;
    const unsigned long long need = (A.n_packets + threads - 1) / threads;
    if ((unsigned long long)blocks > need) blocks = (int)(need ? need : 1);
  }
  {
    void* args[] = {(void*)&M, (void*)&A, (void*)&ctx->V, (void*)&log_ns};   // (k_thermal_voro takes the first three)
    hipLaunchKernel(fn, dim3(blocks), dim3(threads), args, lds, ctx->stream);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ctx->err = std::string("kernel launch: ") + hipGetErrorString(e);
    return MCGPU_ERR_HIP;
  }
  return MCGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// Binned deposits (mc_binned.hip.h): 3D cylindrical grids -- the absorbed-energy array does not fit in LDS
// ---------------------------------------------------------------------------------------------
// can this context's thermal step run with binned deposits?  (cylindrical 3D grids, one dust class -- with the random
// walk since round 4: k_thermal_roles_bin<..., MRW> --; the optional radiation-field arrays are kept by the single-role kernel)
static bool bin_applicable(const mcgpu_ctx* ctx, const RunArgs& A) {
  const DevModel& M = ctx->M;
  if (ctx->voro || M.grid_sph || M.n_classes || !M.l3D || A.xN_abs || A.xJ_abs) return false;
  if (ctx->opt_schedule == 1) return false;
  return true;
}

// bucket width: about 48 buckets, slices of at most 2^14 cells (128 KB of LDS in the fold)
static int bin_shift_for(int n_cells) {
  int s = 6;
  while (s < 14 && ((n_cells + (1 << s) - 1) >> s) > 48) ++s;
  return s;
}

// Does the binned path fit this model at all?  (bucket count, and the staging buckets + tables + the least number of
// packet records in the LDS of one CU.)  The automatic deposit mode falls back to HBM atomics where it does not.
static bool bin_fits(const mcgpu_ctx* ctx) {
  const DevModel& M = ctx->M;
  const int shift = bin_shift_for(M.n_cells);
  const int nb = (M.n_cells + (1 << shift) - 1) >> shift;
  if (nb > BIN_MAX_BUCKETS) return false;
  const size_t lds_cap = 160 * 1024 - 512;
  const size_t lds_t = (lds_bytes(M) + 7) / 8 * 8, lds_b = (bin_lds_bytes(nb) + 7) / 8 * 8;
  return lds_t + lds_b < lds_cap && rq_records_that_fit(ctx->lsepar_pola != 0, lds_cap - lds_t - lds_b) > 0;
}

// The deposit log of this launch.  Automatic size: what the packets asked for are expected to deposit -- n_packets x
// deposits per packet (measured by the context's first launch; 400 before that) x 1.5 of slack, 12 bytes each --,
// at most 64 GiB (89 M blocks = 5.7e9 deposits per chunk; round 5: 24 GiB before -- config 3's step in 10 chunks instead of
// 19, 473 -> 457 ms; 128 GiB buys nothing more: profiles/r05_bin_log_size.log) and at most a quarter of the device's free
// memory (other contexts of the process, xJ_abs, xI_scatt want theirs); an existing log is kept while it is large enough.
static int bin_prepare(mcgpu_ctx* ctx, int n_parts, uint64_t n_packets) {
  const DevModel& M = ctx->M;
  const int shift = bin_shift_for(M.n_cells);
  const int nb = (M.n_cells + (1 << shift) - 1) >> shift;
  if (nb > BIN_MAX_BUCKETS) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "binned deposits: too many cells");
  const size_t block_bytes = (size_t)BIN_H * (sizeof(double) + sizeof(unsigned int));
  const unsigned long long least = (unsigned long long)nb * n_parts * 2;
  const double dep_pp = ctx->bin_dep_per_packet > 0.0 ? ctx->bin_dep_per_packet : 400.0;
  double want_b = (double)n_packets * dep_pp * 1.5 * (double)(sizeof(double) + sizeof(unsigned int));
  if (want_b < 64.0 * 1048576.0) want_b = 64.0 * 1048576.0;
  if (want_b > 64.0 * 1073741824.0) want_b = 64.0 * 1073741824.0;
  if (ctx->opt_log_mb > 0) want_b = (double)((size_t)ctx->opt_log_mb << 20);
  const bool same = ctx->bin.keys && ctx->bin.n_buckets == nb && ctx->bin.shift == shift && ctx->bin_max_parts >= n_parts;
  if (same && (ctx->opt_log_mb > 0 || (double)ctx->bin_total_blocks * (double)block_bytes >= 0.999 * want_b ||
               ctx->bin_log_capped)) return MCGPU_OK;
  bin_release(ctx);
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  size_t bytes = (size_t)want_b;
  ctx->bin_log_capped = false;
  if (ctx->opt_log_mb <= 0 && bytes > free_b / 4) { bytes = free_b / 4; ctx->bin_log_capped = true; }
  unsigned long long blocks = bytes / block_bytes;
  if (blocks < least) blocks = least;
  if (blocks > 0xFFFFFFFFull) blocks = 0xFFFFFFFFull;  // (block indices are 32-bit)
  HIPCHK(hipMalloc((void**)&ctx->bin.keys, blocks * BIN_H * sizeof(unsigned int)));
  HIPCHK(hipMalloc((void**)&ctx->bin.vals, blocks * BIN_H * sizeof(double)));
  HIPCHK(hipMalloc((void**)&ctx->bin.count, (size_t)nb * n_parts * sizeof(unsigned int)));
  HIPCHK(hipMalloc((void**)&ctx->bin.stats, 2 * sizeof(unsigned long long)));
  HIPCHK(hipMalloc((void**)&ctx->d_bin_off, nb * sizeof(unsigned int)));
  HIPCHK(hipMalloc((void**)&ctx->d_bin_cap, nb * sizeof(unsigned int)));
  HIPCHK(hipMalloc((void**)&ctx->d_bin_want, nb * sizeof(double)));
  HIPCHK(hipMemset(ctx->bin.count, 0, (size_t)nb * n_parts * sizeof(unsigned int)));
  HIPCHK(hipMemset(ctx->bin.stats, 0, 2 * sizeof(unsigned long long)));
  ctx->bin.off = ctx->d_bin_off; ctx->bin.cap = ctx->d_bin_cap;
  ctx->bin.n_buckets = nb; ctx->bin.shift = shift; ctx->bin.n_parts = n_parts;
  ctx->bin_total_blocks = blocks;
  ctx->bin_max_parts = n_parts;
  return MCGPU_OK;
}

// The thermal step in chunks: [plan the log's regions] -> packet kernel (deposits to the log) -> fold, all asynchronous
// on the context's stream.  The chunks grow by factors of four from a short first one (whose packets see E = 0 like
// the reference's first packets) up to what the log holds; see bin_energy_scale for the in-flight temperature.
static int launch_binned(mcgpu_ctx* ctx, RunArgs A, const mcgpu_run_opts* o) {
  const DevModel& M = ctx->M;
  const bool pola = ctx->lsepar_pola != 0, dark = M.dark != nullptr;
  const size_t lds_cap = 160 * 1024 - 512;  // (a request of exactly 160 KB minus a few bytes is refused)
  const size_t lds_t = (lds_bytes(M) + 7) / 8 * 8;
  const int rthreads = (o->block_threads > 0 && o->block_threads <= MCGPU_ROLES_BIN_BLOCK) ? o->block_threads : MCGPU_ROLES_BIN_BLOCK;
  if (rthreads % 64) return fail(ctx, MCGPU_ERR_ARG, "block_threads must be a multiple of 64");
  const int n_cu = ctx->prop.multiProcessorCount;
  const int max_parts = o->grid_blocks > n_cu ? o->grid_blocks : n_cu;
  int rc = bin_prepare(ctx, max_parts, A.n_packets);
  if (rc) return rc;
  const size_t lds_b = (bin_lds_bytes(ctx->bin.n_buckets) + 7) / 8 * 8;
  int n_rec = lds_t + lds_b < lds_cap ? rq_records_that_fit(pola, lds_cap - lds_t - lds_b) : 0;
  if (n_rec > 2 * rthreads) n_rec = 2 * rthreads > RQ_MIN_REC ? 2 * rthreads : RQ_MIN_REC;
  if (n_rec <= 0) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "binned deposits: the staging buckets and the packet records do not fit in LDS");
  const size_t lds_r = lds_t + lds_b + rq_lds_bytes(pola, n_rec);
  const void* fn = kpick_roles_bin(pola, dark, M.mrw != 0);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r) != hipSuccess) {
    ctx->err = "binned deposits: LDS request of " + std::to_string(lds_r) + " bytes refused (tables " + std::to_string(lds_t) +
               ", staging " + std::to_string(lds_b) + ", records " + std::to_string(n_rec) + ")";
    return MCGPU_ERR_HIP;
  }
  const size_t fold_lds = sizeof(double) << ctx->bin.shift;
  HIPCHK(hipFuncSetAttribute((const void*)k_fold_bins, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fold_lds));
  int n_srv_pref = tune("MCGPU_N_SRV", (rthreads / 64 + 3) / 4, 1, 1016);
  int k_short = tune("MCGPU_K_SHORT", 2, 0, 64), fly_iters = tune("MCGPU_FLY_ITERS", 16, 1, 256);
  int fly_idle = tune("MCGPU_FLY_IDLE", 32, 1, 65), emit_qmax = tune("MCGPU_EMIT_QMAX", 128, 0, 1 << 20);
  if ((rc = carry_prepare(ctx, max_parts, n_rec, rthreads))) return rc;
  // (the 3D kernel hands packets over anyway -- between chunks --: its last chunk always may; automatic = 48)
  const int tail_thr = ctx->opt_tail >= 0 ? ctx->opt_tail : 48;
  HIPCHK(hipMemsetAsync(ctx->d_carry_n, 0, 2 * sizeof(unsigned int), ctx->stream));

  // chunk sizes: the log holds total_blocks * 64 deposits; a chunk uses at most 60 % of it (the regions carry half as
  // much again as slack: the deposits per packet grow while the disk warms up)
  const double dep_pp = ctx->bin_dep_per_packet > 0.0 ? ctx->bin_dep_per_packet : 400.0;
  double c_max = 0.6 * (double)ctx->bin_total_blocks * BIN_H / dep_pp;
  if (c_max < 1024.0) c_max = 1024.0;
  const uint64_t n_total = A.n_packets, first0 = A.first_packet;
  const double folded0 = o->accumulate ? ctx->accum_packets : 0.0;
  uint64_t done = 0, chunk = 65536, last_chunk = 0;
  if ((double)chunk > c_max) chunk = (uint64_t)c_max;
  int last_parts = 0;
  ctx->bin_chunks = 0;
  const int split = 8;
  while (done < n_total) {
    uint64_t c = chunk < n_total - done ? chunk : n_total - done;
    // (a remainder smaller than a quarter of a chunk rides with this one when the log has room for it)
    if (n_total - done - c < c / 4 && (double)(n_total - done) <= c_max) c = n_total - done;
    int rblocks = o->grid_blocks > 0 ? o->grid_blocks : n_cu;
    const unsigned long long need = (c + rthreads - 1) / rthreads;
    if (o->grid_blocks <= 0 && (unsigned long long)rblocks > need) rblocks = (int)(need ? need : 1);
    if (last_chunk == 0) {
      hipLaunchKernelGGL(k_plan_uniform, dim3(1), dim3(128), 0, ctx->stream, ctx->d_bin_off, ctx->d_bin_cap, ctx->bin.n_buckets,
                         ctx->bin_total_blocks, rblocks);
    } else {
      BinLog Lp = ctx->bin;
      Lp.n_parts = last_parts;
      hipLaunchKernelGGL(k_plan_bins, dim3(1), dim3(128), 0, ctx->stream, Lp, ctx->d_bin_off, ctx->d_bin_cap,
                         (unsigned long long)ctx->bin_total_blocks, (double)c / (double)last_chunk, rblocks, ctx->d_bin_want);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(ctx->d_counters + WORK_SLOT, 0, sizeof(unsigned long long), ctx->stream));
    A.first_packet = first0 + done;
    A.n_packets = c;
    A.n_folded = A.frozen ? 0.0 : folded0 + (double)done;
    A.bin = ctx->bin;
    A.bin.n_parts = rblocks;
    {
      // the packets chunk i leaves unfinished are the first work items of chunk i + 1; the last chunk finishes all
      const int in = ctx->bin_chunks & 1, out = in ^ 1;
      const bool last = done + c >= n_total;
      // (the last chunk hands ITS last packets to the tail kernel instead, once a workgroup has few of them left)
      const bool to_tail = last && tail_thr > 0;
      A.carry_in = ctx->d_carry[in]; A.carry_in_n = ctx->d_carry_n + in;
      A.carry_out = (last && !to_tail) ? nullptr : ctx->d_carry[out]; A.carry_out_n = ctx->d_carry_n + out;
      A.carry_cap = (unsigned int)ctx->carry_cap;
      A.tail_threshold = to_tail ? tail_thr : 0;
      if (!last || to_tail) HIPCHK(hipMemsetAsync(ctx->d_carry_n + out, 0, sizeof(unsigned int), ctx->stream));
    }
    void* args[] = {(void*)&M, (void*)&A, (void*)&n_rec, (void*)&n_srv_pref, (void*)&k_short, (void*)&fly_iters, (void*)&fly_idle, (void*)&emit_qmax};
    HIPCHK(hipLaunchKernel(fn, dim3(rblocks), dim3(rthreads), args, lds_r, ctx->stream));
    hipLaunchKernelGGL(k_fold_bins, dim3(ctx->bin.n_buckets * split), dim3(1024), fold_lds, ctx->stream, A.bin, A.E_abs, M.n_cells, split);
    HIPCHK(hipGetLastError());
    if (done + c >= n_total && tail_thr > 0) {  // behind the fold: E_abs is complete but for these packets' own deposits
      RunArgs At = A;
      At.n_folded = 0.0;
      if ((rc = launch_tail(ctx, At, A.carry_out, A.carry_out_n, true, M.mrw != 0))) return rc;
    }
    done += c;
    last_chunk = c;
    last_parts = rblocks;
    ctx->bin_chunks++;
    if ((double)chunk * 4.0 <= c_max) chunk *= 4; else chunk = (uint64_t)c_max;
  }
  // the counts of the last chunk are cleared for the next launch
  if (last_parts > 0) {
    HIPCHK(hipMemsetAsync(ctx->bin.count, 0, (size_t)ctx->bin.n_buckets * ctx->bin_max_parts * sizeof(unsigned int), ctx->stream));
  }
  return MCGPU_OK;
}

extern "C" int mcgpu_launch_thermal(mcgpu_ctx* ctx, const mcgpu_run_opts* o) {
  int rc = ready(ctx);
  if (rc) return rc;
  if (!o) return fail(ctx, MCGPU_ERR_ARG, "null options");
  if (o->frozen && !ctx->d_E_prior) return fail(ctx, MCGPU_ERR_STATE, "frozen mode needs mcgpu_set_E_prior");
  HIPCHK(hipSetDevice(ctx->device));
  if ((rc = ensure_accum(ctx))) return rc;
  const DevModel& M = ctx->M;
  const size_t lds = lds_bytes(M);
  if (lds > (size_t)ctx->prop.sharedMemPerBlock && lds > 160 * 1024)
    return fail(ctx, MCGPU_ERR_UNSUPPORTED, "wavelength tables exceed the LDS of one CU");
  if (!o->accumulate) {
    HIPCHK(hipMemsetAsync(ctx->d_accum, 0, ctx->n_accum * sizeof(double), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_counters, 0, CNT_SLOTS * sizeof(unsigned long long), ctx->stream));
  } else {
    HIPCHK(hipMemsetAsync(ctx->d_counters + WORK_SLOT, 0, sizeof(unsigned long long), ctx->stream));
  }
  HIPCHK(hipMemsetAsync(ctx->d_err, 0, sizeof(int), ctx->stream));
  ctx->tail_launched = false;
  ctx->tail_on_host = false;
  RunArgs A;
  std::memset(&A, 0, sizeof(A));
  A.seed = o->seed; A.first_packet = o->first_packet; A.n_packets = o->n_packets;
  A.qscale = o->n_replicas >= 1.0 ? o->n_replicas : 1.0;
  A.frozen = o->frozen ? 1 : 0;
  A.E_prior = ctx->d_E_prior;
  A.E_abs = ctx->d_accum;
  A.sed = ctx->d_accum + M.n_cells;
  A.n_sent = ctx->d_accum + M.n_cells + n_sed(M);
  A.counters = ctx->d_counters;
  A.next_packet = ctx->d_counters + WORK_SLOT;
  A.err = ctx->d_err;
  A.inner_iters = tune("MCGPU_INNER_ITERS", 64, 1, 4096);
  A.flush_every = tune("MCGPU_FLUSH_EVERY", 16, 1, 1000000);
  // leave the crossing loop when fewer than this many of 64 live lanes still fly; Voronoi packets interact every 1.4
  // crossings, so their loop is worth one crossing per round (measured: 48 is 5 % ahead of 32)
  A.min_active = tune("MCGPU_MIN_ACTIVE", ctx->voro ? 48 : 32, 0, 64);
  A.flags = tune("MCGPU_DIAG_FLAGS", 0, 0, 0x7FFFFFFF);  // (diagnostic builds only)
  if (ctx->opt_radiation_field & 1) {
    if (!ctx->d_xN) { HIPCHK(hipMalloc((void**)&ctx->d_xN, (size_t)M.n_cells * sizeof(unsigned long long))); HIPCHK(hipMemset(ctx->d_xN, 0, (size_t)M.n_cells * sizeof(unsigned long long))); }
    if (!o->accumulate) HIPCHK(hipMemsetAsync(ctx->d_xN, 0, (size_t)M.n_cells * sizeof(unsigned long long), ctx->stream));
    A.xN_abs = ctx->d_xN;
  }
  if (ctx->opt_radiation_field & 2) {
    const size_t nj = (size_t)M.n_cells * M.n_lambda;
    if (!ctx->d_xJ) { HIPCHK(hipMalloc((void**)&ctx->d_xJ, nj * sizeof(double))); HIPCHK(hipMemset(ctx->d_xJ, 0, nj * sizeof(double))); }
    if (!o->accumulate) HIPCHK(hipMemsetAsync(ctx->d_xJ, 0, nj * sizeof(double), ctx->stream));
    A.xJ_abs = ctx->d_xJ;
  }
  if (ctx->voro) {
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    int rcv = launch_voro(ctx, A, o->grid_blocks, o->block_threads);
    if (rcv) return rcv;
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->launched = true;
    return MCGPU_OK;
  }
  // Deposit mode: a private absorbed-energy grid in LDS when it fits next to the tables
  // (2D grids), HBM atomics otherwise; mcgpu_set_option("deposit") overrides.
  const size_t lds_e = lds + (size_t)M.n_cells * sizeof(double);
  const size_t lds_cap = 160 * 1024;
  bool use_lds = lds_e <= lds_cap;
  if (ctx->opt_deposit == 1) use_lds = false;
  else if (ctx->opt_deposit == 2) {
    if (lds_e > lds_cap) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "deposit = lds: the grid does not fit in LDS");
    use_lds = true;
  }
  // Grids that do not fit in LDS: binned deposits (mc_binned.hip.h) where they are built, HBM atomics otherwise
  bool use_bin = false;
  if (ctx->opt_deposit == 3) {
    if (!bin_applicable(ctx, A)) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "deposit = binned: 3D cylindrical grids, one dust class, role schedule");
    use_bin = true; use_lds = false;
  } else if (ctx->opt_deposit == 0 && !use_lds && bin_applicable(ctx, A) && bin_fits(ctx)) {
    // automatic mode: binned deposits where they fit -- and where the log can be had; otherwise the HBM atomics every
    // such grid ran on before round 3 (only the explicit deposit = 3 reports why the binned path cannot run)
    const int n_cu = ctx->prop.multiProcessorCount;
    use_bin = bin_prepare(ctx, o->grid_blocks > n_cu ? o->grid_blocks : n_cu, A.n_packets) == MCGPU_OK;
    if (!use_bin) { (void)hipGetLastError(); bin_release(ctx); ctx->err.clear(); }
  }
  HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
  int rc3 = use_bin ? launch_binned(ctx, A, o) : launch_mega(ctx, A, use_lds, o->grid_blocks, o->block_threads);
  if (rc3) return rc3;
  HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
  ctx->launched = true;
  ctx->accum_packets = (o->accumulate ? ctx->accum_packets : 0.0) + (double)o->n_packets;
  return MCGPU_OK;
}

extern "C" int mcgpu_sync(mcgpu_ctx* ctx, double* kernel_ms) {
  if (!ctx) return MCGPU_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (kernel_ms) {
    *kernel_ms = 0.0;
    if (ctx->launched) {
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
      *kernel_ms = ms;
    }
  }
  if (ctx->launched && ctx->d_counters) {  // what the next launch plans with: chunk sizes (binned), the tail hand-over
    unsigned long long c[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
    HIPCHK(hipMemcpy(c, ctx->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    if (c[0] > 1000ull) {
      // (sticky: the first launch's measurement sizes the chunks of every later one, so that launches 2, 3, ... of a
      // context follow one plan; the first runs on the guess of 400 deposits per packet)
      if (ctx->bin.keys && !(ctx->bin_dep_per_packet > 0.0)) ctx->bin_dep_per_packet = (double)c[1] / (double)c[0];
      ctx->last_inter_pp = (double)(c[3] + c[4]) / (double)c[0];
    }
  }
  int herr = 0;
  HIPCHK(hipMemcpy(&herr, ctx->d_err, sizeof(int), hipMemcpyDeviceToHost));
  if (herr) {
    ctx->err = "kernel error code " + std::to_string(herr) + " (12 = emission source outside the engine's scope, 13 = a packet of more than 2e8 "
               "crossings was dropped, 15 = a scheduling watchdog ended the launch (a logic error: please report), 16 / 17 = a hand-over "
               "buffer overflowed, 18 = the xI log overflowed; include/mcgpu.h)";
    return MCGPU_ERR_KERNEL;
  }
  return MCGPU_OK;
}

extern "C" int mcgpu_device_accumulators(mcgpu_ctx* ctx, void** accum_dev, uint64_t* n_doubles, void** counters_dev) {
  int rc = ready(ctx);
  if (rc) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  if ((rc = ensure_accum(ctx))) return rc;
  if (accum_dev) *accum_dev = ctx->d_accum;
  if (n_doubles) *n_doubles = ctx->n_accum;
  if (counters_dev) *counters_dev = ctx->d_counters;
  return MCGPU_OK;
}

extern "C" int mcgpu_fetch(mcgpu_ctx* ctx, double* E_abs, double* sed, double* n_sent, uint64_t* counters) {
  int rc = ready(ctx);
  if (rc) return rc;
  if (!ctx->d_accum) return fail(ctx, MCGPU_ERR_STATE, "nothing launched yet");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const DevModel& M = ctx->M;
  if (E_abs) HIPCHK(hipMemcpy(E_abs, ctx->d_accum, (size_t)M.n_cells * sizeof(double), hipMemcpyDeviceToHost));
  if (sed) HIPCHK(hipMemcpy(sed, ctx->d_accum + M.n_cells, n_sed(M) * sizeof(double), hipMemcpyDeviceToHost));
  if (n_sent)
    HIPCHK(hipMemcpy(n_sent, ctx->d_accum + M.n_cells + n_sed(M), (size_t)M.n_lambda * sizeof(double),
                     hipMemcpyDeviceToHost));
  if (counters) HIPCHK(hipMemcpy(counters, ctx->d_counters, MCGPU_N_COUNTERS * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return MCGPU_OK;
}

// The MCGPU_N_COUNTERS event counters ride in the tail of the fused accumulator as doubles (exact below 2^53), so that a
// multi-GPU host reduces ONE buffer per temperature iteration.
__global__ void k_counters_to_accum(const unsigned long long* cnt, double* tail) {
  if (threadIdx.x < MCGPU_N_COUNTERS) tail[threadIdx.x] = (double)cnt[threadIdx.x];
}
__global__ void k_counters_from_accum(unsigned long long* cnt, const double* tail) {
  if (threadIdx.x < MCGPU_N_COUNTERS) cnt[threadIdx.x] = (unsigned long long)(tail[threadIdx.x] + 0.5);
}

// xN_abs(1:n_cells,1) and xJ_abs(1:n_cells,1:n_lambda) of the last thermal launch(es), summed over "threads"
// (radiation_field.f90:20-24, 54-55); needs mcgpu_set_option("radiation_field", bits) before the launch
extern "C" int mcgpu_fetch_radiation_field(mcgpu_ctx* ctx, double* xN_abs, double* xJ_abs) {
  int rc = ready(ctx);
  if (rc) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const DevModel& M = ctx->M;
  if (xN_abs) {
    if (!ctx->d_xN) return fail(ctx, MCGPU_ERR_STATE, "xN_abs was not accumulated (option radiation_field bit 0)");
    std::vector<unsigned long long> h(M.n_cells);
    HIPCHK(hipMemcpy(h.data(), ctx->d_xN, (size_t)M.n_cells * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < M.n_cells; ++i) xN_abs[i] = (double)h[i];
  }
  if (xJ_abs) {
    if (!ctx->d_xJ) return fail(ctx, MCGPU_ERR_STATE, "xJ_abs was not accumulated (option radiation_field bit 1)");
    HIPCHK(hipMemcpy(xJ_abs, ctx->d_xJ, (size_t)M.n_cells * M.n_lambda * sizeof(double), hipMemcpyDeviceToHost));
  }
  return MCGPU_OK;
}

extern "C" int mcgpu_counters_to_accum(mcgpu_ctx* ctx) {
  int rc = ready(ctx);
  if (rc) return rc;
  if (!ctx->d_accum) return fail(ctx, MCGPU_ERR_STATE, "nothing launched yet");
  HIPCHK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_counters_to_accum, dim3(1), dim3(64), 0, ctx->stream, ctx->d_counters,
                     ctx->d_accum + (ctx->n_accum - MCGPU_N_COUNTERS));
  HIPCHK(hipGetLastError());
  return MCGPU_OK;
}

extern "C" int mcgpu_counters_from_accum(mcgpu_ctx* ctx) {
  int rc = ready(ctx);
  if (rc) return rc;
  if (!ctx->d_accum) return fail(ctx, MCGPU_ERR_STATE, "nothing launched yet");
  HIPCHK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_counters_from_accum, dim3(1), dim3(64), 0, ctx->stream, ctx->d_counters,
                     ctx->d_accum + (ctx->n_accum - MCGPU_N_COUNTERS));
  HIPCHK(hipGetLastError());
  return MCGPU_OK;
}

extern "C" int mcgpu_run_thermal(mcgpu_ctx* ctx, const mcgpu_run_opts* opts, double* E_abs, double* sed,
                                 double* n_sent, uint64_t* counters, double* kernel_ms) {
  int rc = mcgpu_launch_thermal(ctx, opts);
  if (rc) return rc;
  if ((rc = mcgpu_sync(ctx, kernel_ms))) return rc;
  return mcgpu_fetch(ctx, E_abs, sed, n_sent, counters);
}

// ---------------------------------------------------------------------------------------------
// repartition_energie(lambda) (thermal_emission.f90:1771-1949), LTE grains: the SED step's emission tables of one
// wavelength, built on the device from the dust temperature
// ---------------------------------------------------------------------------------------------
extern "C" int mcgpu_repartition_energie(mcgpu_ctx* ctx, int lambda, double wl_um, double E_star, double E_ISM, const float* Tdust,
                                         const float* weight_proba_emission, double* frac_E_stars, double* frac_E_disk,
                                         double* E_disk, double* prob_E_cell) {
  if (!ctx || !ctx->have_grid || !ctx->have_opacity || !Tdust) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_repartition_energie: grid, opacity and Tdust are needed");
  const DevModel& M = ctx->M;
  if (lambda < 1 || lambda > M.n_lambda || !(wl_um > 0.0)) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_repartition_energie: bad wavelength");
  const int n = M.n_cells;
  // lweight_emission biases the cells' emission probabilities with weight_proba_emission (thermal_emission.f90:1895-1899) and
  // compensates with the packet's weight, Stokes(1) *= correct_E_emission(icell) (dust_transfer.f90:1140-1142).  The
  // reference's generator of both arrays is commented out (thermal_emission.f90:2078-2135: they stay 1.0, :2147-2148), so
  // there is no compensating weight to restate; a biased table with unit packet weights would be a silently biased SED.
  if (weight_proba_emission)
    for (int i = 0; i < n; ++i)
      if (weight_proba_emission[i] != 1.0f)
        return fail(ctx, MCGPU_ERR_UNSUPPORTED, "mcgpu_repartition_energie: weight_proba_emission other than 1 (lweight_emission: the "
                                               "reference never builds such weights, thermal_emission.f90:2078-2135; the packet-weight half, "
                                               "dust_transfer.f90:1140-1142, is not built)");
  HIPCHK(hipSetDevice(ctx->device));
  DevBuf<float> d_T, d_w;
  DevBuf<double> d_E, d_Ec, d_tot;
  HIPCHK(hipMalloc((void**)&d_T.p, (size_t)n * sizeof(float)));
  HIPCHK(hipMalloc((void**)&d_E.p, (size_t)n * sizeof(double)));
  HIPCHK(hipMalloc((void**)&d_Ec.p, (size_t)n * sizeof(double)));
  HIPCHK(hipMalloc((void**)&d_tot.p, 2 * sizeof(double)));
  HIPCHK(hipMemcpyAsync(d_T.p, Tdust, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  if (weight_proba_emission) {
    HIPCHK(hipMalloc((void**)&d_w.p, (size_t)n * sizeof(float)));
    HIPCHK(hipMemcpyAsync(d_w.p, weight_proba_emission, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  }
  if (!ctx->d_prob_E) HIPCHK(hipMalloc((void**)&ctx->d_prob_E, ((size_t)n + 1) * sizeof(double)));
  const double wl = wl_um * (double)1.e-6f;  // (:1804: the default-real literal 1.e-6)
  hipLaunchKernelGGL(k_repart_E_cell, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, M, lambda, wl, d_T.p, d_w.p, d_E.p, d_Ec.p);
  hipLaunchKernelGGL(k_cumsum_in_order, dim3(1), dim3(SCAN_TILE), 0, ctx->stream, d_Ec.p, d_E.p, n, ctx->d_prob_E, d_tot.p);
  hipLaunchKernelGGL(k_cumsum_normalise, dim3((n + 256) / 256), dim3(256), 0, ctx->stream, ctx->d_prob_E, d_tot.p, n);
  HIPCHK(hipGetLastError());
  double tot[2] = {0.0, 0.0};
  HIPCHK(hipMemcpyAsync(tot, d_tot.p, sizeof(tot), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const double Ed = tot[1];
  ctx->prob_E_lambda = 0;
  if (E_star + Ed + E_ISM < 2.2250738585072014e-308) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_repartition_energie: no energy at this wavelength");  // (:1899-1903)
  const double fs = E_star / (E_star + Ed + E_ISM), fd = (E_star + Ed) / (E_star + Ed + E_ISM);
  if (frac_E_stars) *frac_E_stars = fs;
  if (frac_E_disk) *frac_E_disk = fd;
  if (E_disk) *E_disk = Ed;
  if (prob_E_cell) HIPCHK(hipMemcpy(prob_E_cell, ctx->d_prob_E, ((size_t)n + 1) * sizeof(double), hipMemcpyDeviceToHost));
  ctx->prob_E_lambda = lambda; ctx->prob_E_fstar = fs; ctx->prob_E_fdisk = fd;
  return MCGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// SED mode (mc_mono.hip.h)
// ---------------------------------------------------------------------------------------------
extern "C" int mcgpu_set_rt1(mcgpu_ctx* ctx, int RT_n_incl, int RT_n_az, const double* tab_u_rt,
                             const double* tab_v_rt, const double* tab_w_rt, int n_az_rt, int n_theta_rt,
                             int N_type_flux, int lsepar_contrib, const float* tab_s11_pos, int n_lambda_pos) {
  if (!ctx || RT_n_incl < 1 || RT_n_az < 1 || !tab_u_rt || !tab_v_rt || !tab_w_rt || n_az_rt < 1 ||
      n_theta_rt < 1 || N_type_flux < 1 || !tab_s11_pos || n_lambda_pos < 1)
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_rt1: bad argument");
  if (!ctx->have_scatt) return fail(ctx, MCGPU_ERR_STATE, "set the scattering tables before mcgpu_set_rt1");
  const bool pola = ctx->lsepar_pola != 0;
  const int want = pola ? (lsepar_contrib ? 8 : 4) : (lsepar_contrib ? 5 : 1);  // init_mcfost.f90:1603-1616
  if (N_type_flux != want) return fail(ctx, MCGPU_ERR_ARG, "N_type_flux inconsistent with lsepar_pola / lsepar_contrib");
  if ((ctx->M.l3D || ctx->voro) ? (n_az_rt != 1 || n_theta_rt != 1) : (n_theta_rt != 2))
    return fail(ctx, MCGPU_ERR_UNSUPPORTED, "n_az_rt / n_theta_rt must follow dust_ray_tracing.f90:91-98");
  HIPCHK(hipSetDevice(ctx->device));
  int rc;
  if ((rc = upload(ctx, tab_u_rt, (size_t)RT_n_incl * RT_n_az, &ctx->d_rt_u))) return rc;
  if ((rc = upload(ctx, tab_v_rt, (size_t)RT_n_incl * RT_n_az, &ctx->d_rt_v))) return rc;
  if ((rc = upload(ctx, tab_w_rt, (size_t)RT_n_incl, &ctx->d_rt_w))) return rc;
  if ((rc = upload(ctx, tab_s11_pos, (size_t)(ctx->M.nang + 1) * n_lambda_pos, &ctx->d_tab_s11))) return rc;
  ctx->RT_n_incl = RT_n_incl; ctx->RT_n_az = RT_n_az; ctx->n_az_rt = n_az_rt; ctx->n_theta_rt = n_theta_rt;
  ctx->N_type_flux = N_type_flux; ctx->lsepar_contrib = lsepar_contrib ? 1 : 0; ctx->n_lambda_pos = n_lambda_pos;
  ctx->have_rt1 = true;
  return MCGPU_OK;
}

template <bool SCOUT>
static int launch_mono(mcgpu_ctx* ctx, const MonoArgs& A, int grid_blocks, int block_threads, bool log = false) {
  const DevModel& M = ctx->M;
  const bool pola = ctx->lsepar_pola != 0, dark = M.dark != nullptr, l3d = M.l3D != 0;
  const void* fn;
  // (the commit pass of a context with default-real xI_scatt runs the F32 variant of the deposit code)
  constexpr bool kCommit = !SCOUT;
  const bool f32 = kCommit && A.rt1 && ctx->xI_bytes == 4;
  if (ctx->voro) fn = kpick_mono_voro(pola, SCOUT, f32);
  else if (M.grid_sph) fn = kpick_mono_sph(l3d, pola, SCOUT, f32);
  else fn = kpick_mono(l3d, pola, dark, SCOUT, f32, log);
  // workgroup size: the one that keeps the most wavefronts on a CU, up to 8 (2 per SIMD).  The r02 build needs only
  // 101-169 VGPRs here, so the registers would admit 3-4 waves per SIMD -- measured slower (bench sed 4.03e7 against
  // 4.87e7 packets/s): this mode is bound by the xI_scatt atomics, and more waves in flight only deepen their queues.
  // The LDS of a workgroup is the shared tables plus the per-lane ray-tracing scratch, which grows with the observers.
  // (the kernel that logs its deposits makes no atomics, needs 146 VGPRs and almost no LDS: three waves per SIMD)
  // (so does a scout pass of the cylindrical / spherical kernels: no deposits, no per-lane results, no tiles -- round 6;
  // until then it was launched with the commit pass's LDS and two waves per SIMD)
  const bool lean = log || (SCOUT && !ctx->voro);
  // (the commit pass with default-real records and one dust class on cylindrical / spherical grids keeps the flights'
  // deposit weights as rows in the sub-bin's order: deposit_rt1_wave_row)
  const int rowf = (f32 && !lean && !ctx->voro && !M.n_classes) ? xi32_row_floats(A.xi, A.nRT) : 0;
#ifndef MCGPU_MONO_CU_THREADS
#define MCGPU_MONO_CU_THREADS 512   // (A/B builds)
#endif
  const int cu_threads = lean ? 768 : MCGPU_MONO_CU_THREADS;
  int threads = 0;
  const bool slim = true;                   // (mono_lds_bytes)
  const int max_threads = 512;              // (__launch_bounds__ of the kernels)
  if (block_threads > 0 && block_threads <= max_threads && block_threads % 64 == 0) {
    threads = block_threads;
  } else {
    int best_waves = 0;
    for (int th = lean ? 256 : max_threads; th >= 64; th -= 64) {
      const size_t l = mono_lds_bytes(M, A.nRT, th, pola, slim, lean, 0, rowf);
      if (l > 160 * 1024) continue;
      int per_cu = (int)((160 * 1024) / l);
      if (per_cu > cu_threads / th) per_cu = cu_threads / th;
      const int waves = per_cu * th / 64;
      if (waves > best_waves) { best_waves = waves; threads = th; }
    }
    if (!threads) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "too many ray-tracing directions for the LDS of one CU");
  }
  size_t lds = mono_lds_bytes(M, A.nRT, threads, pola, slim, lean, 0, rowf);
  if (lds > 160 * 1024) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "too many ray-tracing directions for the LDS of one CU");
  // The commit pass's one global load per crossing, kappa_factor, from LDS where the grid is small enough to ride along
  // without costing a workgroup its place on the CU (2D grids with few observers): see mono_lds_bytes.
  MonoArgs A2 = A;
  A2.kf_lds = 0;
  A2.rowf = rowf;
  if (kCommit && !lean && (A.rt1 || A.rt2) && !ctx->voro && !M.n_classes) {
    const size_t with_kf = mono_lds_bytes(M, A.nRT, threads, pola, slim, lean, M.n_cells, rowf);
    const int per_cu_max = cu_threads / threads > 0 ? cu_threads / threads : 1;
    int per_cu = (int)((160 * 1024) / lds), per_cu_kf = with_kf <= 160 * 1024 ? (int)((160 * 1024) / with_kf) : 0;
    if (per_cu > per_cu_max) per_cu = per_cu_max;
    if (per_cu_kf > per_cu_max) per_cu_kf = per_cu_max;
    if (per_cu_kf >= per_cu && per_cu_kf > 0) { A2.kf_lds = 1; lds = with_kf; }
  }
  HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int blocks = grid_blocks;
  if (blocks <= 0) {
    int occ = 1;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, threads, lds));
    if (occ < 1) occ = 1;
    blocks = ctx->prop.multiProcessorCount * occ;
    const unsigned long long need = (A.n_items + threads - 1) / threads;
    if ((unsigned long long)blocks > need) blocks = (int)(need ? need : 1);
  }
  void* args[] = {(void*)&M, (void*)&A2, (void*)&ctx->V};  // the Voronoi kernels take the grid as 3rd argument
  HIPCHK(hipLaunchKernel(fn, dim3(blocks), dim3(threads), args, lds, ctx->stream));
  return MCGPU_OK;
}

// Ray tracing method 2 (lscatt_ray_tracing2): the accumulators of save_radiation_field's branch radiation_field.f90:91-129
extern "C" int mcgpu_set_rt2(mcgpu_ctx* ctx, int n_theta_I, int n_phi_I, int N_type_flux, int lsepar_contrib) {
  if (!ctx) return MCGPU_ERR_ARG;
  if (!ctx->have_grid) return fail(ctx, MCGPU_ERR_STATE, "set the grid first");
  const int n_Stokes = ctx->lsepar_pola ? 4 : 1;
  if (n_theta_I < 1 || n_phi_I < 1 || n_theta_I > 1024 || n_phi_I > 1024 || N_type_flux != n_Stokes + (lsepar_contrib ? 4 : 0))
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_rt2: N_type_flux = n_Stokes (+ 4 with lsepar_contrib)");
  if (ctx->M.l3D || ctx->voro) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "ray tracing method 2 is for 2D grids (cylindrical or spherical)");
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->d_I_spec) hipFree(ctx->d_I_spec);
  if (ctx->d_I_spec_star) hipFree(ctx->d_I_spec_star);
  ctx->d_I_spec = ctx->d_I_spec_star = nullptr; ctx->have_rt2 = false;
  const size_t n = (size_t)ctx->M.n_cells * n_phi_I * n_theta_I * XI_LINE;
  HIPCHK(hipMalloc((void**)&ctx->d_I_spec, n * sizeof(double)));
  HIPCHK(hipMalloc((void**)&ctx->d_I_spec_star, (size_t)ctx->M.n_cells * sizeof(double)));
  HIPCHK(hipMemset(ctx->d_I_spec, 0, n * sizeof(double)));
  HIPCHK(hipMemset(ctx->d_I_spec_star, 0, (size_t)ctx->M.n_cells * sizeof(double)));
  ctx->n_theta_I = n_theta_I; ctx->n_phi_I = n_phi_I; ctx->rt2_N_type_flux = N_type_flux; ctx->rt2_contrib = lsepar_contrib ? 1 : 0;
  ctx->have_rt2 = true;
  return MCGPU_OK;
}

// I_spec(N_type_flux, n_theta_I, n_phi_I, n_cells) and I_spec_star(n_cells) in the reference's layout: default real
// and / or the FP64 sums the device holds
extern "C" int mcgpu_fetch_I_spec(mcgpu_ctx* ctx, float* I_spec, double* I_spec_f64, float* I_spec_star, double* I_spec_star_f64) {
  if (!ctx || !ctx->have_rt2) return fail(ctx, MCGPU_ERR_STATE, "mcgpu_set_rt2 first");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const int nt = ctx->n_theta_I, np = ctx->n_phi_I, ntf = ctx->rt2_N_type_flux, nc = ctx->M.n_cells;
  std::vector<double> t((size_t)nc * np * nt * XI_LINE), st((size_t)nc);
  HIPCHK(hipMemcpy(t.data(), ctx->d_I_spec, t.size() * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(st.data(), ctx->d_I_spec_star, st.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int c = 0; c < nc; ++c)
    for (int p = 0; p < np; ++p)
      for (int th = 0; th < nt; ++th)
        for (int f = 0; f < ntf; ++f) {
          const double v = t[((((size_t)c * np + p) * nt) + th) * XI_LINE + f];
          const size_t o = (size_t)f + (size_t)ntf * ((size_t)th + (size_t)nt * ((size_t)p + (size_t)np * c));
          if (I_spec) I_spec[o] = (float)v;
          if (I_spec_f64) I_spec_f64[o] = v;
        }
  for (int c = 0; c < nc; ++c) {
    if (I_spec_star) I_spec_star[c] = (float)st[c];
    if (I_spec_star_f64) I_spec_star_f64[c] = st[c];
  }
  return MCGPU_OK;
}

// the reference's I_spec / I_spec_star handed to the device (e.g. sums over several processes): see include/mcgpu.h
extern "C" int mcgpu_set_I_spec(mcgpu_ctx* ctx, const double* I_spec, const double* I_spec_star) {
  if (!ctx || !ctx->have_rt2) return fail(ctx, MCGPU_ERR_STATE, "mcgpu_set_rt2 first");
  if (!I_spec || !I_spec_star) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_I_spec: null argument");
  HIPCHK(hipSetDevice(ctx->device));
  const int nt = ctx->n_theta_I, np = ctx->n_phi_I, ntf = ctx->rt2_N_type_flux, nc = ctx->M.n_cells;
  const size_t n = (size_t)ntf * nt * np * nc;
  DevBuf<double> d_in;
  HIPCHK(d_in.alloc(n)); HIPCHK(d_in.put(I_spec, n));
  HIPCHK(hipMemsetAsync(ctx->d_I_spec, 0, (size_t)nc * np * nt * XI_LINE * sizeof(double), ctx->stream));
  hipLaunchKernelGGL(k_I_spec_put, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_I_spec, d_in.p, ntf, nt, np, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(ctx->d_I_spec_star, I_spec_star, (size_t)nc * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MCGPU_OK;
}

// init_dust_source_fct2 (dust_ray_tracing.f90:717-806) of one inclination on the device: see include/mcgpu.h
extern "C" int mcgpu_rt2_source(mcgpu_ctx* ctx, const mcgpu_rt_opts* o, int p_lambda, int ibin, const float* Tdust,
                                const double* r_grid, const double* z_grid, int nang_ray_tracing,
                                int nang_ray_tracing_star, float* eps_dust2, float* eps_dust2_star, double* kernel_ms) {
  int rc = ready(ctx);
  if (rc) return rc;
  if (!o || !Tdust || !r_grid || !z_grid || nang_ray_tracing < 1 || nang_ray_tracing_star < 1 ||
      nang_ray_tracing > 4096 || nang_ray_tracing_star > 65536)
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_rt2_source: bad argument");
  DevModel& M = ctx->M;
  if (!ctx->have_rt2 || !ctx->have_rt1) return fail(ctx, MCGPU_ERR_STATE, "mcgpu_rt2_source needs mcgpu_set_rt1 (the observers, tab_s11_pos) and mcgpu_set_rt2");
  if (M.l3D || ctx->voro) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "ray tracing method 2 is 2D only");
  if (M.n_classes && !M.v_s11) return fail(ctx, MCGPU_ERR_STATE, "variable dust: tab_s11_pos per class is missing (mcgpu_opacity or mcgpu_set_variable_dust_s11)");
  if (o->lambda < 1 || o->lambda > M.n_lambda || p_lambda < 1 || p_lambda > M.n_lambda || ibin < 1 || ibin > ctx->RT_n_incl ||
      !(o->wl_um > 0.0) || !(o->n_sent_photons > 0.0))
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_rt2_source: bad option");
  if (!M.n_classes && p_lambda > ctx->n_lambda_pos) return fail(ctx, MCGPU_ERR_ARG, "p_lambda out of range");
  HIPCHK(hipSetDevice(ctx->device));
  const int nt = ctx->n_theta_I, np = ctx->n_phi_I, na = nang_ray_tracing, ns = nang_ray_tracing_star, nang = M.nang;
  const bool pola = ctx->lsepar_pola != 0;
  const int n_Stokes = pola ? 4 : 1, ntf = ctx->rt2_N_type_flux;
  // the observer's direction (tab_w_rt(ibin), tab_uv_rt(ibin) = sin(incl)): dust_ray_tracing.f90:280-289
  std::vector<double> h_w(ctx->RT_n_incl);
  HIPCHK(hipMemcpy(h_w.data(), ctx->d_rt_w, h_w.size() * sizeof(double), hipMemcpyDeviceToHost));
  const double w0 = h_w[ibin - 1], uv0 = std::sqrt(1.0 - w0 * w0);
  // ---- the directions where Inu * s11 is evaluated (:973-1072)
  const size_t ntab = (size_t)2 * na * np * nt;
  std::vector<int> tab_k(ntab * RT2_NSUP2);
  std::vector<float> tab_sin(ntab * RT2_NSUP2);
  std::vector<double> tab_cosw(ntab, 0.0), tab_sinw(ntab, 0.0);
  const double PI_ = 3.14159265358979323846;
  auto angle_index = [&](float cos_scatt) {
    const float ac = (float)std::acos((double)cos_scatt);
    if (ac != ac) return nang;
    return (int)std::llrint(std::floor((double)(ac * (float)nang) / PI_ + 0.5));
  };
  for (int dir = 0; dir <= 1; ++dir)
    for (int iscatt = 1; iscatt <= na; ++iscatt) {
      const float phi_scatt = (float)(2 * PI_ * (double)((float)iscatt / (float)na));
      const double ur = uv0 * std::sin((double)phi_scatt), vr = -uv0 * std::cos((double)phi_scatt), wr = w0;
      for (int phi_I = 1; phi_I <= np; ++phi_I)
        for (int theta_I = 1; theta_I <= nt; ++theta_I) {
          const size_t b = (((size_t)dir * na + (iscatt - 1)) * np + (phi_I - 1)) * nt + (theta_I - 1);
          float sum_sin = 0.f;
          for (int i2 = 1; i2 <= RT2_N_SUPER; ++i2)
            for (int i1 = 1; i1 <= RT2_N_SUPER; ++i1) {
              const float f1 = (float)i1 / (float)(RT2_N_SUPER + 1), f2 = (float)i2 / (float)(RT2_N_SUPER + 1);
              const double w = (2.0 * (((double)theta_I - (double)f1) / (double)nt) - 1.0) * (double)(2 * dir - 1);
              const double phi = 2 * PI_ * ((double)phi_I - (double)f2) / (double)np;
              const double w02 = std::sqrt(1.0 - w * w), u = w02 * std::sin(phi), v = -w02 * std::cos(phi);
              const float cos_scatt = (float)(ur * u + vr * v + wr * w);
              int k = angle_index(cos_scatt);
              if (k > nang) k = nang;
              if (k < 0) k = 0;
              const float sin_scatt = (float)std::sqrt(1.0 - (double)cos_scatt * (double)cos_scatt);
              tab_k[b * RT2_NSUP2 + (i1 - 1) + RT2_N_SUPER * (i2 - 1)] = k;
              tab_sin[b * RT2_NSUP2 + (i1 - 1) + RT2_N_SUPER * (i2 - 1)] = sin_scatt;
              sum_sin = sum_sin + sin_scatt;
            }
          for (int t = 0; t < RT2_NSUP2; ++t) tab_sin[b * RT2_NSUP2 + t] = tab_sin[b * RT2_NSUP2 + t] / sum_sin;
          if (pola) {
            const double w = (2.0 * (((double)theta_I - (double)0.5f) / (double)nt) - 1.0) * (double)(2 * dir - 1);
            const double phi = 2 * PI_ * ((double)phi_I - (double)0.5f) / (double)np;
            const double w02 = std::sqrt(1.0 - w * w), u = w02 * std::sin(phi), v = -w02 * std::cos(phi);
            double v1pi, v1pj, v1pk;
            host_rotation(u, v, w, -ur, -vr, -wr, v1pi, v1pj, v1pk);
            const double xnyp = std::sqrt(v1pk * v1pk + v1pj * v1pj);
            const double costhet = (xnyp < 1e-10) ? 1.0 : v1pj / xnyp;
            double theta = std::acos(costhet);
            if (theta >= PI_) theta = 0.0;
            double omega = 2.0 * theta;
            if (v1pk < 0.0) omega = -1.0 * omega;
            double cosw = std::cos(omega), sinw = std::sin(omega);
            if (std::fabs(cosw) < 1e-06) cosw = 0.0;
            if (std::fabs(sinw) < 1e-06) sinw = 0.0;
            tab_cosw[b] = cosw; tab_sinw[b] = sinw;
          }
        }
    }
  DevBuf<int> d_k;
  DevBuf<float> d_sin, d_T;
  DevBuf<double> d_cw, d_sw, d_J, d_rg, d_zg;
  HIPCHK(d_k.alloc(tab_k.size())); HIPCHK(d_k.put(tab_k.data(), tab_k.size()));
  HIPCHK(d_sin.alloc(tab_sin.size())); HIPCHK(d_sin.put(tab_sin.data(), tab_sin.size()));
  HIPCHK(d_cw.alloc(ntab)); HIPCHK(d_cw.put(tab_cosw.data(), ntab));
  HIPCHK(d_sw.alloc(ntab)); HIPCHK(d_sw.put(tab_sinw.data(), ntab));
  HIPCHK(d_T.alloc(M.n_cells)); HIPCHK(d_T.put(Tdust, M.n_cells));
  HIPCHK(d_J.alloc(M.n_cells));
  HIPCHK(d_rg.alloc(M.n_cells)); HIPCHK(d_rg.put(r_grid, M.n_cells));
  HIPCHK(d_zg.alloc(M.n_cells)); HIPCHK(d_zg.put(z_grid, M.n_cells));
  const size_t n_eps = (size_t)ntf * na * 2 * M.n_cells, n_eps_s = (size_t)n_Stokes * ns * 2 * M.n_cells;
  // (the result stays in HBM for mcgpu_rt2_dust_map / mcgpu_rt2_image)
  if (ctx->d_eps2) hipFree(ctx->d_eps2);
  if (ctx->d_eps2_star) hipFree(ctx->d_eps2_star);
  if (ctx->d_rt2_zgrid) hipFree(ctx->d_rt2_zgrid);
  ctx->d_eps2 = ctx->d_eps2_star = nullptr; ctx->d_rt2_zgrid = nullptr; ctx->rt2_src_ibin = 0;
  HIPCHK(hipMalloc((void**)&ctx->d_eps2, n_eps * sizeof(float)));
  HIPCHK(hipMalloc((void**)&ctx->d_eps2_star, n_eps_s * sizeof(float)));
  HIPCHK(hipMalloc((void**)&ctx->d_rt2_zgrid, (size_t)M.n_cells * sizeof(double)));
  HIPCHK(hipMemcpyAsync(ctx->d_rt2_zgrid, z_grid, (size_t)M.n_cells * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  Rt2Args A;
  std::memset(&A, 0, sizeof(A));
  A.lambda = o->lambda; A.p_lambda = p_lambda; A.n_theta_I = nt; A.n_phi_I = np; A.nang_rt = na; A.nang_star = ns;
  A.n_Stokes = n_Stokes; A.N_type_flux = ntf; A.contrib = ctx->rt2_contrib; A.pola = pola ? 1 : 0;
  const double AU_to_cm = 149597870700.0 * 100.0;
  A.photon_energy = o->E_src * o->wl_um * 1.0e-6 / (o->n_sent_photons * AU_to_cm * M_PI);
  A.uv0 = uv0; A.w0 = w0;
  A.I_spec = ctx->d_I_spec; A.I_spec_star = ctx->d_I_spec_star; A.J_th = d_J.p; A.r_grid = d_rg.p; A.z_grid = d_zg.p;
  A.tab_k = d_k.p; A.tab_sin = d_sin.p; A.tab_cosw = d_cw.p; A.tab_sinw = d_sw.p;
  A.s11_single = M.n_classes ? nullptr : ctx->d_tab_s11 + (size_t)(M.nang + 1) * (p_lambda - 1);
  A.eps_dust2 = ctx->d_eps2; A.eps_dust2_star = ctx->d_eps2_star;
  HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
  hipLaunchKernelGGL(k_calc_Jth, dim3((M.n_cells + 255) / 256), dim3(256), 0, ctx->stream, M, o->lambda, o->wl_um * 1.e-6, d_T.p, d_J.p);
  const size_t n1 = (size_t)M.n_cells * 2 * na, n2 = (size_t)M.n_cells * 2 * ns;
  if (pola) {
    hipLaunchKernelGGL(k_rt2_source<true>, dim3((unsigned)((n1 + 127) / 128)), dim3(128), 0, ctx->stream, M, A);
    hipLaunchKernelGGL(k_rt2_source_star<true>, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, ctx->stream, M, A);
  } else {
    hipLaunchKernelGGL(k_rt2_source<false>, dim3((unsigned)((n1 + 127) / 128)), dim3(128), 0, ctx->stream, M, A);
    hipLaunchKernelGGL(k_rt2_source_star<false>, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, ctx->stream, M, A);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (kernel_ms) { float ms = 0.f; hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1); *kernel_ms = ms; }
  ctx->rt2_src_ibin = ibin; ctx->rt2_src_lambda = o->lambda; ctx->rt2_src_nang = na; ctx->rt2_src_nang_star = ns;
  if (eps_dust2) HIPCHK(hipMemcpy(eps_dust2, ctx->d_eps2, n_eps * sizeof(float), hipMemcpyDeviceToHost));
  if (eps_dust2_star) HIPCHK(hipMemcpy(eps_dust2_star, ctx->d_eps2_star, n_eps_s * sizeof(float), hipMemcpyDeviceToHost));
  return MCGPU_OK;
}

// ---- the commit pass with its deposits as a log (mc_mono.hip.h "The deposits as a log"; mc_xilog.hip.h) ---------------
// Does this context's commit pass log its xI_scatt deposits?  (Default-real records, one dust class, cylindrical grid.)
// Automatic (option "xi_log" = 1): only where a crossing's atomics touch at least four lines -- with fewer observers the
// atomics are cheaper than the sort and the fold whatever the flights' length (ref4.1 at 60 um, 36 crossings per flight:
// 3 observers 207 ms with atomics against 333 with the log, 6 observers 296 against 349, 10 observers 384 against 367).
static bool xi_log_applicable(const mcgpu_ctx* ctx, bool rt1) {
  const DevModel& M = ctx->M;
  if (!(rt1 && ctx->opt_xi_log != 0 && ctx->xI_bytes == 4 && !M.n_classes && !ctx->voro && !M.grid_sph)) return false;
  return ctx->opt_xi_log == 2 || xi32_lines_touched(xi_layout_of(ctx), ctx->RT_n_incl * ctx->RT_n_az) >= 4;
}

// The log's buffers: the launch's records and its flights' rows, the sorted copy of the records, the sort's scratch.
// Sized for launches of up to 2^28 records / 2^26 flights (2 x 3.2 + 10.7 GB with ten observers and Stokes tracking), less
// for a run that cannot fill them; a wavelength's commit pass runs in as many launches as that takes.
// (Measured and dropped, profiles/r06_xi_log_ab.log: two buffer sets with the sort and the fold of launch i on a second
// stream under launch i + 1's transport -- the persistent transport kernel leaves them a wave per SIMD, the sort ran 4x
// slower, and the sum stayed what it was.)
static int xi_log_prepare(mcgpu_ctx* ctx, unsigned long long n_items, int nRT, bool pola) {
  const size_t row_floats = (size_t)nRT * (pola ? 4 : 1);
  size_t cap = (size_t)1 << 28, rows = (size_t)1 << 26;
  // (a run of few packets -- the tests' -- does not need gigabytes: ~4096 crossings and 1024 flights per packet are provided for)
  while (cap > ((size_t)1 << 22) && (double)cap > 4096.0 * (double)n_items + 8.0e6) cap >>= 1;
  while (rows > ((size_t)1 << 20) && (double)rows > 1024.0 * (double)n_items + 2.0e6) rows >>= 1;
  if (ctx->xlog_cap >= cap && ctx->xlog_rows_cap >= rows && ctx->xlog_row_floats == row_floats) return MCGPU_OK;
  if (ctx->xlog_cap > cap) cap = ctx->xlog_cap;
  if (ctx->xlog_rows_cap > rows && ctx->xlog_row_floats == row_floats) rows = ctx->xlog_rows_cap;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 2; ++i) {
    if (ctx->d_xlog_keys[i]) hipFree(ctx->d_xlog_keys[i]);
    if (ctx->d_xlog_vals[i]) hipFree(ctx->d_xlog_vals[i]);
    ctx->d_xlog_keys[i] = nullptr; ctx->d_xlog_vals[i] = nullptr;
  }
  if (ctx->d_xlog_rows) hipFree(ctx->d_xlog_rows);
  if (ctx->d_xlog_temp) hipFree(ctx->d_xlog_temp);
  ctx->d_xlog_rows = nullptr; ctx->d_xlog_temp = nullptr; ctx->xlog_cap = 0; ctx->xlog_rows_cap = 0;
  for (int i = 0; i < 2; ++i) {   // [0]: the launch's log; [1]: the sorted copy
    HIPCHK(hipMalloc((void**)&ctx->d_xlog_keys[i], cap * sizeof(unsigned int)));
    HIPCHK(hipMalloc((void**)&ctx->d_xlog_vals[i], cap * sizeof(unsigned long long)));
  }
  HIPCHK(hipMalloc((void**)&ctx->d_xlog_rows, rows * row_floats * sizeof(float)));
  if (!ctx->d_xlog_ctl) HIPCHK(hipMalloc((void**)&ctx->d_xlog_ctl, 2 * sizeof(unsigned long long)));
  ctx->xlog_temp_bytes = xi_sort_temp_bytes(cap, 31);
  HIPCHK(hipMalloc(&ctx->d_xlog_temp, ctx->xlog_temp_bytes ? ctx->xlog_temp_bytes : 16));
  ctx->xlog_cap = cap; ctx->xlog_rows_cap = rows; ctx->xlog_row_floats = row_floats;
  return MCGPU_OK;
}

// Below this many logged crossings per flight the commit pass deposits with atomics: a flight costs its row -- 160 bytes
// written once and gathered once per crossing -- whatever its length.  Measured at ten observers against the atomics in the
// packed layout (4 lines per crossing; profiles/r06_xi_log_ab.log): 36 crossings per flight (ref4.1 at 60 um) 367 against
// 384 ms; 6 per flight (0.3 um) 633 against 534 ms; 2.2 per flight (1 um) 1.70 against 1.30 s.
constexpr double XI_LOG_MIN_CROSSINGS_PER_FLIGHT = 20.0;

// One commit pass (the work items [0, A.n_items) of `A`): plainly, or -- `*mode` = 1: with its deposits logged, in launches
// sized for the log, each followed by the sort and the fold of what it logged.  The first launch is short and measures
// records and flights per packet; the others take what 70 % of the buffers hold at that rate -- or, where the flights
// turn out too short for the log to pay, the rest of the pass (and of the call: *mode = 2) runs with atomics.
static int commit_mono(mcgpu_ctx* ctx, MonoArgs A, int grid_blocks, int block_threads, int* mode) {
  if (*mode != 1) return launch_mono<false>(ctx, A, grid_blocks, block_threads);
  const DevModel& M = ctx->M;
  const bool pola = ctx->lsepar_pola != 0;
  const unsigned long long n_total = A.n_items;
  int rc = xi_log_prepare(ctx, n_total, A.nRT, pola);
  if (rc) return rc;
  const unsigned int n_bins = (unsigned int)((size_t)M.n_cells * A.n_theta_rt * A.n_az_rt);
  int end_bit = 1;
  while (end_bit < 31 && (1u << end_bit) <= n_bins) ++end_bit;   // the unused entries' key, 2^end_bit - 1 >= n_bins, sorts last
  if ((1u << end_bit) - 1u < n_bins) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "xI log: too many sub-bins for a 31-bit key");
  A.log_keys = ctx->d_xlog_keys[0]; A.log_vals = ctx->d_xlog_vals[0]; A.log_rows = ctx->d_xlog_rows; A.log_ctl = ctx->d_xlog_ctl;
  A.log_cap = ctx->xlog_cap; A.rows_cap = ctx->xlog_rows_cap; A.log_sentinel = (1u << end_bit) - 1u;
  const int nv = pola ? 4 : 1;
  // (the first launch: room for 1024 records and 256 flights per packet -- ref4.1 has 60-160 and 2-60, by wavelength)
  unsigned long long done = 0, chunk = 1000000ull;
  if (chunk > ctx->xlog_cap / 1024) chunk = ctx->xlog_cap / 1024;
  if (chunk > ctx->xlog_rows_cap / 256) chunk = ctx->xlog_rows_cap / 256;
  double rpp = 0.0, fpp = 0.0;   // records / flights reserved per packet, as measured
  while (done < n_total) {
    if (rpp > 0.0) {
      double c = 0.7 * (double)ctx->xlog_cap / rpp;
      if (fpp > 0.0 && 0.7 * (double)ctx->xlog_rows_cap / fpp < c) c = 0.7 * (double)ctx->xlog_rows_cap / fpp;
      chunk = c < 1024.0 ? 1024ull : (unsigned long long)c;
    }
    const unsigned long long c = chunk < n_total - done ? chunk : n_total - done;
    HIPCHK(hipMemsetAsync(ctx->d_xlog_ctl, 0, 2 * sizeof(unsigned long long), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_counters + WORK_SLOT, 0, sizeof(unsigned long long), ctx->stream));
    A.item_lo = done; A.n_items = c;
    if ((rc = launch_mono<false>(ctx, A, grid_blocks, block_threads, true))) return rc;
    unsigned long long ctl[2] = {0ull, 0ull};
    int dev_err = 0;
    HIPCHK(hipMemcpyAsync(ctl, ctx->d_xlog_ctl, sizeof(ctl), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(&dev_err, ctx->d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));   // (the host needs the record count for the sort)
    if (dev_err == 18 || ctl[0] > ctx->xlog_cap || ctl[1] > ctx->xlog_rows_cap) {
      ctx->err = "xI log: a launch of " + std::to_string(c) + " packets logged " + std::to_string(ctl[0]) + " records and " + std::to_string(ctl[1]) +
                 " flights, more than the buffers hold (" + std::to_string(ctx->xlog_cap) + ", " + std::to_string(ctx->xlog_rows_cap) +
                 "): mcgpu_set_option(ctx, \"xi_log\", 0) runs this model with atomics";
      return MCGPU_ERR_KERNEL;
    }
    if (dev_err) { ctx->err = "device error " + std::to_string(dev_err) + " in the commit pass"; return MCGPU_ERR_KERNEL; }
    const int e = xi_sort_fold(ctx->stream, ctx->d_xlog_keys[0], ctx->d_xlog_vals[0], ctx->d_xlog_keys[1], ctx->d_xlog_vals[1], (size_t)ctl[0],
                               end_bit, ctx->d_xlog_temp, ctx->xlog_temp_bytes, ctx->d_xlog_rows, A.nRT, nv, A.contrib, n_bins,
                               reinterpret_cast<float*>(ctx->d_xI), A.xi);
    if (e != (int)hipSuccess) { ctx->err = std::string("xI log: sort / fold: ") + hipGetErrorString((hipError_t)e); return MCGPU_ERR_HIP; }
    rpp = (double)ctl[0] / (double)c; fpp = (double)ctl[1] / (double)c;
    ctx->xlog_chunks++; ctx->xlog_records += ctl[0]; ctx->xlog_flights += ctl[1];
    done += c;
    if (ctx->opt_xi_log == 1 && done < n_total && ctl[1] > 0 && (double)ctl[0] / (double)ctl[1] < XI_LOG_MIN_CROSSINGS_PER_FLIGHT) {
      *mode = 2;   // short flights: the rest with atomics (the sums do not care who adds them)
      A.item_lo = done; A.n_items = n_total - done;
      A.log_keys = nullptr; A.log_vals = nullptr; A.log_rows = nullptr; A.log_ctl = nullptr;
      HIPCHK(hipMemsetAsync(ctx->d_counters + WORK_SLOT, 0, sizeof(unsigned long long), ctx->stream));
      return launch_mono<false>(ctx, A, grid_blocks, block_threads);
    }
  }
  return MCGPU_OK;
}

extern "C" int mcgpu_run_mono(mcgpu_ctx* ctx, const mcgpu_mono_opts* o, double frac_E_stars, double frac_E_disk,
                              const double* prob_E_cell, uint64_t* n_sent_chunk, double* kernel_ms) {
  int rc = ready(ctx);
  if (rc) return rc;
  if (!o) return fail(ctx, MCGPU_ERR_ARG, "null options");
  DevModel& M = ctx->M;
  if (M.n_classes) {  // lvariable_dust: the tables of every class, with a column per wavelength (p_lambda = lambda in SED mode)
    if (M.v_scatt && M.aniso_method == 1 && M.p_lambda_fixed)
      return fail(ctx, MCGPU_ERR_STATE, "SED mode with variable dust needs prob_s11_pos per wavelength: set the scattering tables with p_lambda_fixed = 0");
    if (!M.v_scatt) return fail(ctx, MCGPU_ERR_STATE, "SED mode with variable dust needs the per-class scattering tables");
    if (o->rt1 == 1 && !M.v_s11) return fail(ctx, MCGPU_ERR_STATE, "rt1 deposits with variable dust need tab_s11_pos per class (mcgpu_opacity or mcgpu_set_variable_dust_s11)");
    if (M.m1) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "scattering method 1 is for the thermal step (ray tracing forces method 2, init_mcfost.f90:1659)");
  }
  if (o->lambda < 1 || o->lambda > M.n_lambda || o->n_chunks < 1 || o->n_chunks > (1 << 22) || o->capt_sup < 1 ||
      o->first_chunk < 0 || (long long)o->first_chunk + o->n_chunks > (1 << 23))
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_run_mono: bad option");
  const bool rt2 = o->rt1 == 2, rt1 = o->rt1 != 0 && !rt2;   // opts->rt1: 0 none, 1 lscatt_ray_tracing1, 2 lscatt_ray_tracing2
  if (rt1 && !ctx->have_rt1) return fail(ctx, MCGPU_ERR_STATE, "rt1 deposits need mcgpu_set_rt1");
  if (rt2 && !ctx->have_rt2) return fail(ctx, MCGPU_ERR_STATE, "rt2 deposits need mcgpu_set_rt2");
  if (rt2 && (M.l3D || ctx->voro)) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "ray tracing method 2 is 2D only (radiation_field.f90:91)");
  if (M.grid_sph && M.dark)   // (the reference never has one there: `if (lspherical.or.l3D) call no_dark_zone()`, dust_transfer.f90:734, 916)
    return fail(ctx, MCGPU_ERR_UNSUPPORTED, "SED mode on a spherical grid: no dark zone (the reference defines none there)");
  const int n_pos = ctx->have_rt1 ? ctx->n_lambda_pos : M.n_lambda;
  if (o->p_lambda < 1 || o->p_lambda > n_pos || o->p_lambda > M.n_lambda) return fail(ctx, MCGPU_ERR_ARG, "p_lambda out of range");
  // prob_E_cell = NULL: the table mcgpu_repartition_energie left on the device for this wavelength
  if (frac_E_stars < 1.0 && frac_E_disk > frac_E_stars && !prob_E_cell && ctx->prob_E_lambda != o->lambda)
    return fail(ctx, MCGPU_ERR_ARG, "disk emission needs prob_E_cell (or mcgpu_repartition_energie of this wavelength first)");
  if (frac_E_disk < 1.0 && !(M.R_ISM > 0.0)) return fail(ctx, MCGPU_ERR_ARG, "frac_E_disk < 1 needs mcgpu_set_ism");
  HIPCHK(hipSetDevice(ctx->device));
  if ((rc = ensure_accum(ctx))) return rc;
  const int nc = o->n_chunks;
  if (ctx->mono_chunks < nc) {
    if (ctx->d_mono_u64) hipFree(ctx->d_mono_u64);
    if (ctx->d_mono_i32) hipFree(ctx->d_mono_i32);
    ctx->d_mono_u64 = nullptr; ctx->d_mono_i32 = nullptr; ctx->mono_chunks = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_mono_u64, ((size_t)5 * nc + 1) * sizeof(unsigned long long)));
    HIPCHK(hipMalloc((void**)&ctx->d_mono_i32, (size_t)2 * nc * sizeof(int)));
    ctx->mono_chunks = nc;
  }
  unsigned long long *d_need = ctx->d_mono_u64, *d_sent = d_need + nc, *d_base = d_sent + nc;
  unsigned long long *d_start = d_base + nc + 1, *d_hitcnt = d_start + nc;
  int *d_active = ctx->d_mono_i32, *d_done = d_active + nc;
  if (prob_E_cell) {
    if (!ctx->d_prob_E) HIPCHK(hipMalloc((void**)&ctx->d_prob_E, ((size_t)M.n_cells + 1) * sizeof(double)));
    HIPCHK(hipMemcpyAsync(ctx->d_prob_E, prob_E_cell, ((size_t)M.n_cells + 1) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ctx->prob_E_lambda = 0;
  }
  const int nRT = ctx->have_rt1 ? ctx->RT_n_incl * ctx->RT_n_az : 0;
  // n_xI: elements of the reference's array; the device keeps XI_LINE doubles per (cell, sub-bin, observer)
  const size_t n_xI = rt1 ? (size_t)ctx->n_az_rt * ctx->n_theta_rt * ctx->N_type_flux * nRT * (size_t)M.n_cells : 0;
  const size_t xi_bytes = rt1 ? xi_dev_bytes(ctx) : 0;
  if (rt1 && ctx->N_type_flux > XI_LINE) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "N_type_flux > 8");
  if (rt1 && ctx->n_xI != n_xI) {
    if (ctx->d_xI) hipFree(ctx->d_xI);
    ctx->d_xI = nullptr; ctx->n_xI = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_xI, xi_bytes));
    ctx->n_xI = n_xI;
    HIPCHK(hipMemsetAsync(ctx->d_xI, 0, xi_bytes, ctx->stream));
  } else if (rt1 && !o->accumulate) {
    HIPCHK(hipMemsetAsync(ctx->d_xI, 0, xi_bytes, ctx->stream));
  }
  if (rt2 && !o->accumulate) {
    HIPCHK(hipMemsetAsync(ctx->d_I_spec, 0, (size_t)M.n_cells * ctx->n_phi_I * ctx->n_theta_I * XI_LINE * sizeof(double), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_I_spec_star, 0, (size_t)M.n_cells * sizeof(double), ctx->stream));
  }
  // (the E_abs part of the fused accumulator belongs to the thermal step: a host may run the SED Monte Carlo and
  // then call mcgpu_temp_finale(ctx, NULL, ...) on the device's own absorbed-energy grid)
  if (!o->accumulate) {
    HIPCHK(hipMemsetAsync(ctx->d_accum + M.n_cells, 0, (ctx->n_accum - M.n_cells) * sizeof(double), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_counters, 0, CNT_SLOTS * sizeof(unsigned long long), ctx->stream));
  }
  HIPCHK(hipMemsetAsync(ctx->d_err, 0, sizeof(int), ctx->stream));

  MonoArgs A;
  std::memset(&A, 0, sizeof(A));
  A.seed = o->seed; A.lambda = o->lambda; A.p_lambda = o->p_lambda; A.capt_sup = o->capt_sup; A.rt1 = rt1 ? 1 : 0;
  A.frac_E_stars = frac_E_stars; A.frac_E_disk = frac_E_disk;
  A.prob_E_cell = (prob_E_cell || ctx->prob_E_lambda == o->lambda) ? ctx->d_prob_E : nullptr;
  A.n_chunks = nc; A.first_chunk = o->first_chunk;
  A.RT_n_incl = ctx->have_rt1 ? ctx->RT_n_incl : 1; A.nRT = rt1 ? nRT : 0;
  A.rt_u = ctx->d_rt_u; A.rt_v = ctx->d_rt_v; A.rt_w = ctx->d_rt_w;
  A.n_az_rt = ctx->n_az_rt; A.n_theta_rt = ctx->n_theta_rt; A.N_type_flux = ctx->N_type_flux; A.contrib = ctx->lsepar_contrib;
  A.s11 = ctx->have_rt1 ? ctx->d_tab_s11 + (size_t)(M.nang + 1) * (o->p_lambda - 1) : nullptr;
  A.xI = ctx->d_xI; A.xI_f32 = ctx->xI_bytes == 4 ? 1 : 0; A.xi = xi_layout_of(ctx);
  if (rt2) {
    A.rt2 = 1; A.n_theta_I = ctx->n_theta_I; A.n_phi_I = ctx->n_phi_I; A.I_spec = ctx->d_I_spec; A.I_spec_star = ctx->d_I_spec_star;
    A.N_type_flux = ctx->rt2_N_type_flux; A.contrib = ctx->rt2_contrib;
  }
  A.sed = ctx->d_accum + M.n_cells;
  A.n_sent = ctx->d_accum + M.n_cells + n_sed(M);
  A.counters = ctx->d_counters; A.next_item = ctx->d_counters + WORK_SLOT; A.err = ctx->d_err;
  A.inner_iters = tune("MCGPU_INNER_ITERS", 64, 1, 4096);
  A.min_active = tune("MCGPU_MIN_ACTIVE", 32, 0, 64);
  A.flags = tune("MCGPU_DIAG_FLAGS", 0, 0, 255);  // (diagnostic builds only)
  int xlog_mode = xi_log_applicable(ctx, rt1) ? 1 : 0;   // 1: the commit passes log their xI_scatt deposits (mc_xilog.hip.h folds them)
  ctx->xlog_chunks = 0; ctx->xlog_records = 0; ctx->xlog_flights = 0;

  // ---- SCOUT: find every stream's stopping index (dust_transfer.f90:526-553) ----------------
  double lim_d = std::ceil((double)o->n_phot_lim);
  if (!(lim_d >= 0.0)) lim_d = 0.0;
  const unsigned long long lim = lim_d > 9.0e18 ? ~0ull : (unsigned long long)lim_d;
  // Speculative commit: after a short first scout batch has measured the rate at which packets land in capt_sup,
  // most of every stream -- as many packets as can be sent with (statistical) certainty before the stopping packet,
  // 7 sigma short of it -- is committed directly, with its hits counted, and only the remainder is scouted.  Without
  // it every packet is transported twice.  Should a stream reach its count inside the speculative range after all,
  // the accumulators are cleared and the call starts over without speculation (so: only when the call owns them).
  bool speculate = !o->accumulate && ctx->opt_speculation != 0;
restart:
  std::vector<unsigned long long> need(nc, o->n_photons2), sent(nc, 0ull), start(nc, 0ull);
  std::vector<int> active, done(nc, 0);
  bool probed = false;
  if (o->n_photons2 >= lim) {
    // a stream cannot collect n_photons2 packets in capt_sup out of fewer than n_photons2 sent: every stream runs
    // to n_phot_lim and nothing has to be scouted (image mode, run_image_mc: p_nnfot2 => nnfot2, :507-508, 711-713)
    for (int c = 0; c < nc; ++c) sent[c] = lim;
  } else if (o->n_photons2 > 0 && lim > 0) {
    for (int c = 0; c < nc; ++c) active.push_back(c);
  }
  HIPCHK(hipMemcpyAsync(d_need, need.data(), nc * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d_sent, sent.data(), nc * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
  double rate = 1.0 / (double)M.N_thet;  // first guess: an isotropic source fills the bins evenly
  unsigned long long scout_packets = 0, hits_found = 0;
  while (!active.empty()) {
    const int na = (int)active.size();
    unsigned long long max_need = 0;
    for (int c : active) if (need[c] > max_need) max_need = need[c];
    double b = 1.25 * (double)max_need / rate + 64.0;
    if (speculate && !probed) b = 0.05 * (double)max_need / rate + 64.0;  // the probe that measures the rate
    const double bmax = 2.0e9 / (double)na;
    if (b > bmax) b = bmax;
    unsigned long long max_left = 0;  // no stream can use more than what n_phot_lim leaves it
    for (int c : active) if (lim - sent[c] > max_left) max_left = lim - sent[c];
    if (b > (double)max_left) b = (double)max_left;
    if (b < 64.0) b = 64.0;
    const unsigned long long batch = ((unsigned long long)b + 63ull) / 64ull * 64ull;
    const size_t nh = (size_t)na * batch;
    if (ctx->hits_cap < nh) {
      if (ctx->d_hits) hipFree(ctx->d_hits);
      ctx->d_hits = nullptr; ctx->hits_cap = 0;
      HIPCHK(hipMalloc((void**)&ctx->d_hits, nh));
      ctx->hits_cap = nh;
    }
    HIPCHK(hipMemsetAsync(ctx->d_hits, 0, nh, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_active, active.data(), na * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_counters + WORK_SLOT, 0, sizeof(unsigned long long), ctx->stream));
    A.active = d_active; A.seq0 = d_sent; A.batch = batch; A.hits = ctx->d_hits; A.n_items = nh;
    if ((rc = launch_mono<true>(ctx, A, o->grid_blocks, o->block_threads))) return rc;
    hipLaunchKernelGGL(k_mono_scan, dim3(na), dim3(64), 0, ctx->stream, d_active, na, batch, ctx->d_hits, d_need, d_sent,
                       lim, d_done);
    HIPCHK(hipGetLastError());
    std::vector<unsigned long long> need2(nc), sent2(nc);
    HIPCHK(hipMemcpyAsync(need2.data(), d_need, nc * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(sent2.data(), d_sent, nc * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(done.data(), d_done, nc * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    int dev_err = 0;
    HIPCHK(hipMemcpy(&dev_err, ctx->d_err, sizeof(int), hipMemcpyDeviceToHost));
    if (dev_err) { ctx->err = "device error " + std::to_string(dev_err) + " in the scout pass"; return MCGPU_ERR_KERNEL; }
    std::vector<int> still;
    for (int c : active) {
      scout_packets += sent2[c] - sent[c];
      hits_found += need[c] - need2[c];
      if (!done[c]) still.push_back(c);
    }
    need = need2; sent = sent2;
    active.swap(still);
    if (scout_packets > 0 && hits_found > 0) rate = (double)hits_found / (double)scout_packets;
    else rate *= 0.25;  // nothing landed in capt_sup yet: widen the next batch

    if (speculate && !probed) {
      probed = true;
      std::vector<unsigned long long> extra(nc, 0ull), cnt(nc, 0ull);
      unsigned long long total_extra = 0;
      if (hits_found >= 100)
        for (int c : active) {
          const double nr = (double)need[c];
          if (nr < 400.0) continue;
          const double margin = 7.0 / std::sqrt(nr) + 0.02 + 3.0 / std::sqrt((double)hits_found);
          double n1 = std::floor(nr * (1.0 - margin) / rate);
          if (n1 < 0.0) n1 = 0.0;
          if (n1 > (double)(lim - sent[c])) n1 = (double)(lim - sent[c]);
          extra[c] = (unsigned long long)n1;
          total_extra += extra[c];
        }
      if (total_extra > 0) {
        std::vector<unsigned long long> base(nc + 1, 0ull);
        for (int c = 0; c < nc; ++c) { cnt[c] = sent[c] + extra[c]; base[c + 1] = base[c] + cnt[c]; }
        HIPCHK(hipMemcpyAsync(d_base, base.data(), ((size_t)nc + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemsetAsync(d_hitcnt, 0, nc * sizeof(unsigned long long), ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->d_counters + WORK_SLOT, 0, sizeof(unsigned long long), ctx->stream));
        A.item_base = d_base; A.n_items = base[nc]; A.active = nullptr; A.seq0 = nullptr; A.hits = nullptr; A.batch = 0;
        A.hit_count = d_hitcnt;
        if ((rc = commit_mono(ctx, A, o->grid_blocks, o->block_threads, &xlog_mode))) return rc;
        A.hit_count = nullptr;
        std::vector<unsigned long long> hc(nc);
        HIPCHK(hipMemcpyAsync(hc.data(), d_hitcnt, nc * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipMemcpy(&dev_err, ctx->d_err, sizeof(int), hipMemcpyDeviceToHost));
        if (dev_err) { ctx->err = "device error " + std::to_string(dev_err) + " in the commit pass"; return MCGPU_ERR_KERNEL; }
        bool overshoot = false;
        for (int c : active) if (extra[c] > 0 && hc[c] >= o->n_photons2) overshoot = true;
        if (overshoot) {  // (a 7-sigma event) clear what this call accumulated and run it the plain way
          speculate = false;
          HIPCHK(hipMemsetAsync(ctx->d_accum + M.n_cells, 0, (ctx->n_accum - M.n_cells) * sizeof(double), ctx->stream));
          HIPCHK(hipMemsetAsync(ctx->d_counters, 0, CNT_SLOTS * sizeof(unsigned long long), ctx->stream));
          if (rt1) HIPCHK(hipMemsetAsync(ctx->d_xI, 0, xi_bytes, ctx->stream));
          if (rt2) {
            HIPCHK(hipMemsetAsync(ctx->d_I_spec, 0, (size_t)M.n_cells * ctx->n_phi_I * ctx->n_theta_I * XI_LINE * sizeof(double), ctx->stream));
            HIPCHK(hipMemsetAsync(ctx->d_I_spec_star, 0, (size_t)M.n_cells * sizeof(double), ctx->stream));
          }
          goto restart;
        }
        for (int c = 0; c < nc; ++c) start[c] = cnt[c];   // committed so far: [0, start)
        for (int c : active)
          if (extra[c] > 0) { need[c] = o->n_photons2 - hc[c]; sent[c] += extra[c]; }
        HIPCHK(hipMemcpyAsync(d_need, need.data(), nc * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(d_sent, sent.data(), nc * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
      }
    }
  }

  // ---- COMMIT: exactly the packets s < K of every stream, with deposits -----------------------
  std::vector<unsigned long long> base(nc + 1, 0ull);
  for (int c = 0; c < nc; ++c) base[c + 1] = base[c] + (sent[c] - start[c]);
  if (n_sent_chunk) for (int c = 0; c < nc; ++c) n_sent_chunk[c] = sent[c];
  HIPCHK(hipMemcpyAsync(d_base, base.data(), ((size_t)nc + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d_start, start.data(), nc * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_counters + WORK_SLOT, 0, sizeof(unsigned long long), ctx->stream));
  A.item_base = d_base; A.n_items = base[nc]; A.active = nullptr; A.seq0 = d_start; A.hits = nullptr; A.batch = 0;
  if (A.n_items > 0 && (rc = commit_mono(ctx, A, o->grid_blocks, o->block_threads, &xlog_mode))) return rc;
  HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
  ctx->launched = true;
  if ((rc = mcgpu_sync(ctx, kernel_ms))) return rc;
  return MCGPU_OK;
}

extern "C" int mcgpu_device_xI(mcgpu_ctx* ctx, void** xI_dev, uint64_t* n_doubles) {
  if (!ctx || !ctx->d_xI) return fail(ctx, MCGPU_ERR_STATE, "no xI_scatt accumulated yet");
  if (xI_dev) *xI_dev = ctx->d_xI;
  if (n_doubles) *n_doubles = xi_dev_values(ctx);
  return MCGPU_OK;
}

extern "C" int mcgpu_set_xI_precision(mcgpu_ctx* ctx, int bytes_per_value) {
  if (!ctx) return MCGPU_ERR_ARG;
  if (bytes_per_value != 4 && bytes_per_value != 8) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_xI_precision: 4 or 8");
  if (bytes_per_value != ctx->xI_bytes) {
    hipSetDevice(ctx->device);
    if (ctx->d_xI) hipFree(ctx->d_xI);  // the layout changes with the type: what was accumulated is dropped
    ctx->d_xI = nullptr; ctx->n_xI = 0;
    ctx->xI_bytes = bytes_per_value;
  }
  return MCGPU_OK;
}

extern "C" int mcgpu_get_xI_precision(mcgpu_ctx* ctx) { return ctx ? ctx->xI_bytes : 0; }

extern "C" int mcgpu_set_xI(mcgpu_ctx* ctx, const double* xI_scatt) {
  if (!ctx || !xI_scatt) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_set_xI: null argument");
  if (!ctx->have_rt1) return fail(ctx, MCGPU_ERR_STATE, "mcgpu_set_xI needs mcgpu_set_rt1");
  if (ctx->N_type_flux > XI_LINE) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "N_type_flux > 8");
  HIPCHK(hipSetDevice(ctx->device));
  const int nRT = ctx->RT_n_incl * ctx->RT_n_az;
  const size_t n = (size_t)ctx->n_az_rt * ctx->n_theta_rt * ctx->N_type_flux * nRT * (size_t)ctx->M.n_cells;
  const size_t xi_bytes = xi_dev_bytes(ctx);
  if (ctx->n_xI != n) {
    if (ctx->d_xI) hipFree(ctx->d_xI);
    ctx->d_xI = nullptr; ctx->n_xI = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_xI, xi_bytes));
    ctx->n_xI = n;
  }
  HIPCHK(hipMemsetAsync(ctx->d_xI, 0, xi_bytes, ctx->stream));
  double* d_in = nullptr;
  HIPCHK(hipMalloc((void**)&d_in, n * sizeof(double)));
  hipError_t e = hipMemcpyAsync(d_in, xI_scatt, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_xI_put, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_xI, d_in,
                       ctx->n_az_rt, ctx->n_theta_rt, ctx->N_type_flux, nRT, n, ctx->xI_bytes == 4 ? 1 : 0, xi_layout_of(ctx), ctx->lsepar_pola ? 4 : 1);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(d_in);
  HIPCHK(e);
  return MCGPU_OK;
}

extern "C" int mcgpu_fetch_xI(mcgpu_ctx* ctx, float* xI_scatt_f32, double* xI_scatt_f64) {
  if (!ctx || !ctx->d_xI) return fail(ctx, MCGPU_ERR_STATE, "no xI_scatt accumulated yet");
  HIPCHK(hipSetDevice(ctx->device));
  const size_t n = ctx->n_xI;
  float* d32 = nullptr;
  double* d64 = nullptr;
  hipError_t e = hipSuccess;
  if (xI_scatt_f32) e = hipMalloc((void**)&d32, n * sizeof(float));
  if (e == hipSuccess && xI_scatt_f64) e = hipMalloc((void**)&d64, n * sizeof(double));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_xI_fetch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_xI, d32, d64,
                       ctx->n_az_rt, ctx->n_theta_rt, ctx->N_type_flux, ctx->RT_n_incl * ctx->RT_n_az, n,
                       ctx->xI_bytes == 4 ? 1 : 0, xi_layout_of(ctx), ctx->lsepar_pola ? 4 : 1);
    e = hipGetLastError();
  }
  if (e == hipSuccess && d32) e = hipMemcpyAsync(xI_scatt_f32, d32, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && d64) e = hipMemcpyAsync(xI_scatt_f64, d64, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (d32) hipFree(d32);
  if (d64) hipFree(d64);
  HIPCHK(e);
  return MCGPU_OK;
}

extern "C" int mcgpu_temp_finale(mcgpu_ctx* ctx, const double* E_abs, float* Tdust) {
  int rc = ready(ctx);
  if (rc) return rc;
  if (!Tdust) return fail(ctx, MCGPU_ERR_ARG, "null Tdust");
  HIPCHK(hipSetDevice(ctx->device));
  const DevModel& M = ctx->M;
  double* d_E = nullptr;
  const double* src = ctx->d_accum;
  if (E_abs) {
    HIPCHK(hipMalloc((void**)&d_E, (size_t)M.n_cells * sizeof(double)));
    HIPCHK(hipMemcpyAsync(d_E, E_abs, (size_t)M.n_cells * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    src = d_E;
  } else if (!src) {
    return fail(ctx, MCGPU_ERR_STATE, "no accumulator to reduce");
  }
  float* d_T = nullptr;
  HIPCHK(hipMalloc((void**)&d_T, (size_t)M.n_cells * sizeof(float)));
  const int threads = 256, blocks = (M.n_cells + threads - 1) / threads;
  hipLaunchKernelGGL(k_temp_finale, dim3(blocks), dim3(threads), 0, ctx->stream, M, src, ctx->d_tab_Temp,
                     ctx->T_min, d_T);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(Tdust, d_T, (size_t)M.n_cells * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  hipFree(d_T);
  if (d_E) hipFree(d_E);
  return MCGPU_OK;
}

// ---- probes ---------------------------------------------------------------

// Temp_approx_diffusion_vertical (diffusion.f90:292-374, called at dust_transfer.f90:316,659 after Temp_finale when
// the model has a dark zone): refills the temperature of the dark zone and of delta_cell_dark_zone cells around it.
extern "C" int mcgpu_temp_approx_diffusion_vertical(mcgpu_ctx* ctx, const double* tab_lambda, const double* tab_delta_lambda,
                                                    int ri_in_dark_zone, int ri_out_dark_zone, const int* zj_sup_dark_zone,
                                                    float* Tdust, int* n_iterations) {
  int rc = ready(ctx);
  if (rc) return rc;
  if (!tab_lambda || !tab_delta_lambda || !zj_sup_dark_zone || !Tdust) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_temp_approx_diffusion_vertical: null argument");
  const DevModel& M = ctx->M;
  if (ctx->voro || M.l3D || M.grid_sph) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "diffusion approximation: 2D cylindrical grids only");
  if (ri_in_dark_zone < 1 || ri_out_dark_zone > M.n_rad) return fail(ctx, MCGPU_ERR_ARG, "dark-zone radii out of range");
  for (int i = 0; i < M.n_rad; ++i)
    if (zj_sup_dark_zone[i] < 0 || zj_sup_dark_zone[i] > M.nz) return fail(ctx, MCGPU_ERR_ARG, "zj_sup_dark_zone out of range");
  HIPCHK(hipSetDevice(ctx->device));
  DevBuf<double> d_lam, d_dl;
  DevBuf<int> d_zj, d_it;
  DevBuf<float> d_T;
  HIPCHK(d_lam.alloc(M.n_lambda)); HIPCHK(d_lam.put(tab_lambda, M.n_lambda));
  HIPCHK(d_dl.alloc(M.n_lambda)); HIPCHK(d_dl.put(tab_delta_lambda, M.n_lambda));
  HIPCHK(d_zj.alloc(M.n_rad)); HIPCHK(d_zj.put(zj_sup_dark_zone, M.n_rad));
  HIPCHK(d_it.alloc(1)); HIPCHK(hipMemset(d_it.p, 0, sizeof(int)));
  HIPCHK(d_T.alloc(M.n_cells)); HIPCHK(d_T.put(Tdust, M.n_cells));
  HIPCHK(hipMemsetAsync(ctx->d_err, 0, sizeof(int), ctx->stream));
  if (ri_out_dark_zone >= ri_in_dark_zone) {
    hipLaunchKernelGGL(k_clean_dark_temperature, dim3(ri_out_dark_zone - ri_in_dark_zone + 1), dim3(64), 0, ctx->stream,
                       M.n_rad, ri_in_dark_zone, ri_out_dark_zone, d_zj.p, ctx->T_min, d_T.p);
    HIPCHK(hipGetLastError());
  }
  const int i_lo = ri_in_dark_zone - DELTA_CELL_DARK_ZONE > 3 ? ri_in_dark_zone - DELTA_CELL_DARK_ZONE : 3;
  const int i_hi = ri_out_dark_zone + DELTA_CELL_DARK_ZONE < M.n_rad - 2 ? ri_out_dark_zone + DELTA_CELL_DARK_ZONE : M.n_rad - 2;
  if (i_hi >= i_lo) {
    const size_t lds = (size_t)3 * (M.nz + 2) * sizeof(double);
    hipLaunchKernelGGL(k_diffusion_vertical, dim3(i_hi - i_lo + 1), dim3(128), lds, ctx->stream, M, d_lam.p, d_dl.p, i_lo, i_hi,
                       d_zj.p, d_T.p, d_it.p, ctx->d_err);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  int herr = 0;
  HIPCHK(hipMemcpy(&herr, ctx->d_err, sizeof(int), hipMemcpyDeviceToHost));
  if (herr) return fail(ctx, MCGPU_ERR_KERNEL, "diffusion approximation did not converge");
  HIPCHK(d_T.get(Tdust, M.n_cells));
  if (n_iterations) HIPCHK(d_it.get(n_iterations, 1));
  return MCGPU_OK;
}


// ---------------------------------------------------------------------------------------------
// RT1 ray-traced dust SED (mc_raytrace.hip.h)
// ---------------------------------------------------------------------------------------------
// common part of the two entry points: checks, J_th, the RtArgs both kernels share
struct Rt1Job {
  DevBuf<float> d_T, d_az;
  DevBuf<double> d_J;
  RtArgs A;
  bool method2 = false;
};

static int rt1_prepare(mcgpu_ctx* ctx, const mcgpu_rt_opts* o, const float* tab_RT_az, const float* Tdust, Rt1Job& J,
                       const char* who) {
  int rc = ready(ctx);
  if (rc) return rc;
  if (ctx->voro && J.method2) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "ray tracing on a Voronoi grid: method 1");
  if (ctx->M.grid_sph && ctx->M.dark)
    return fail(ctx, MCGPU_ERR_UNSUPPORTED, "ray tracing on a spherical grid: no dark zone (the reference defines none there)");
  if (!o || !tab_RT_az || !Tdust) return fail(ctx, MCGPU_ERR_ARG, "RT1 ray tracing: null argument");
  if (!ctx->have_rt1 || (!ctx->d_xI && !J.method2))
    return fail(ctx, MCGPU_ERR_STATE, "RT1 ray tracing needs the xI_scatt of mcgpu_run_mono(rt1=1) or mcgpu_set_xI");
  if (J.method2 && (!ctx->d_eps2 || !ctx->rt2_src_ibin || ctx->rt2_src_lambda != (o ? o->lambda : 0)))
    return fail(ctx, MCGPU_ERR_STATE, "method 2 ray tracing needs mcgpu_rt2_source of this wavelength first");
  if (J.method2 && ctx->RT_n_az != 1) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "method 2 is 2D and knows one observer azimuth (RT_n_az = 1)");
  const DevModel& M = ctx->M;
  if (o->lambda < 1 || o->lambda > M.n_lambda || !(o->wl_um > 0.0) || !(o->n_sent_photons > 0.0) ||
      !(o->distance > 0.0) || !(o->Rmin > 0.0) || !(o->Rmax > o->Rmin))
    return fail(ctx, MCGPU_ERR_ARG, who);
  HIPCHK(hipSetDevice(ctx->device));
  const int nRT = ctx->RT_n_incl * ctx->RT_n_az;
  HIPCHK(J.d_T.alloc(M.n_cells)); HIPCHK(J.d_T.put(Tdust, M.n_cells));
  HIPCHK(J.d_az.alloc(ctx->RT_n_az)); HIPCHK(J.d_az.put(tab_RT_az, ctx->RT_n_az));
  HIPCHK(J.d_J.alloc(M.n_cells));
  RtArgs& A = J.A;
  std::memset(&A, 0, sizeof(A));
  A.lambda = o->lambda; A.RT_n_incl = ctx->RT_n_incl; A.nRT = nRT; A.n_az_rt = ctx->n_az_rt; A.n_theta_rt = ctx->n_theta_rt;
  A.N_type_flux = ctx->N_type_flux; A.contrib = ctx->lsepar_contrib; A.l_sym_ima = o->l_sym_ima ? 1 : 0;
  A.wl = o->wl_um * 1.e-6;
  const double AU_to_cm = 149597870700.0 * 100.0, pc_to_AU = 648000.0 / M_PI;
  A.photon_energy = o->E_src * o->wl_um * 1.0e-6 / (o->n_sent_photons * AU_to_cm * M_PI);
  A.pix_scale = 1.0 / (o->distance * pc_to_AU);
  A.ang_disque = o->ang_disque; A.tau_dark_zone_obs = o->tau_dark_zone_obs;
  A.rmin_RT = 0.01 * o->Rmin;
  const double rmax_RT = 2.0 * o->Rmax;
  A.fact_r = std::exp((1.0 / ((double)RT_N_RAD - 1)) * std::log(rmax_RT / A.rmin_RT));
  A.fact_A = std::sqrt(M_PI * (A.fact_r - 1.0 / A.fact_r) / RT_N_PHI);
  A.cst_phi = (o->l_sym_ima ? M_PI : 2 * M_PI) / (double)RT_N_PHI;
  A.l_far = 10. * o->Rmax;
  A.rt_u = ctx->d_rt_u; A.rt_v = ctx->d_rt_v; A.rt_w = ctx->d_rt_w; A.rt_az = J.d_az.p;
  A.xI = ctx->d_xI; A.J_th = J.d_J.p;
  A.xI_f32 = ctx->xI_bytes == 4 ? 1 : 0; A.xi = xi_layout_of(ctx);
  if (J.method2) {
    A.method2 = 1; A.q_only = ctx->rt2_src_ibin - 1; A.nang_rt = ctx->rt2_src_nang; A.nang_star = ctx->rt2_src_nang_star;
    A.eps2 = ctx->d_eps2; A.eps2_star = ctx->d_eps2_star; A.z_grid = ctx->d_rt2_zgrid;
    A.N_type_flux = ctx->rt2_N_type_flux; A.contrib = ctx->rt2_contrib;
  }
  HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
  hipLaunchKernelGGL(k_calc_Jth, dim3((M.n_cells + 255) / 256), dim3(256), 0, ctx->stream, M, o->lambda, A.wl, J.d_T.p, J.d_J.p);
  HIPCHK(hipGetLastError());
  return MCGPU_OK;
}

template <bool IMAGE>
static int rt1_launch(mcgpu_ctx* ctx, const RtArgs& A, int blocks) {
  const size_t lds = lds_bytes(ctx->M, true);
  const bool pola = A.N_type_flux == 4 || A.N_type_flux == 8, l3d = ctx->M.l3D != 0;
#define RT1_GO(a, b) do {                                                                                          \
    const void* fn = IMAGE ? (const void*)k_rt1_image<a, b> : (const void*)k_rt1_dust_map<a, b>;                    \
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
    if (IMAGE) hipLaunchKernelGGL((k_rt1_image<a, b>), dim3(blocks), dim3(256), lds, ctx->stream, ctx->M, A);       \
    else hipLaunchKernelGGL((k_rt1_dust_map<a, b>), dim3(blocks), dim3(256), lds, ctx->stream, ctx->M, A);          \
  } while (0)
#define RT1_GO_VORO(b) do {                                                                                         \
    const void* fn = IMAGE ? (const void*)k_rt1_image_voro<b> : (const void*)k_rt1_dust_map_voro<b>;                 \
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
    if (IMAGE) hipLaunchKernelGGL((k_rt1_image_voro<b>), dim3(blocks), dim3(256), lds, ctx->stream, ctx->M, A, ctx->V);   \
    else hipLaunchKernelGGL((k_rt1_dust_map_voro<b>), dim3(blocks), dim3(256), lds, ctx->stream, ctx->M, A, ctx->V);      \
  } while (0)
  if (ctx->voro) { if (pola) RT1_GO_VORO(true); else RT1_GO_VORO(false); }
  else if (l3d) { if (pola) RT1_GO(true, true); else RT1_GO(true, false); }
  else { if (pola) RT1_GO(false, true); else RT1_GO(false, false); }
#undef RT1_GO_VORO
#undef RT1_GO
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MCGPU_OK;
}

extern "C" int mcgpu_rt1_dust_map(mcgpu_ctx* ctx, const mcgpu_rt_opts* o, const float* tab_RT_az, const float* Tdust,
                                  double* stokes, double* kernel_ms) {
  if (ctx && !stokes) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_rt1_dust_map: null argument");
  Rt1Job J;
  int rc = rt1_prepare(ctx, o, tab_RT_az, Tdust, J, "mcgpu_rt1_dust_map: bad option");
  if (rc) return rc;
  const size_t n_out = (size_t)J.A.nRT * ctx->N_type_flux;
  DevBuf<double> d_out;
  HIPCHK(d_out.alloc(n_out));
  HIPCHK(hipMemsetAsync(d_out.p, 0, n_out * sizeof(double), ctx->stream));
  J.A.out = d_out.p;
  const int n_rays = J.A.nRT * RT_N_RAD * RT_N_PHI;
  if ((rc = rt1_launch<false>(ctx, J.A, (n_rays + 255) / 256))) return rc;  // one ray per lane
  if (kernel_ms) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1)); *kernel_ms = ms; }
  HIPCHK(d_out.get(stokes, n_out));
  return MCGPU_OK;
}

// define_dark_zone (optical_depth.f90:1425-1651), 2D cylindrical grids: see include/mcgpu.h
extern "C" int mcgpu_define_dark_zone(mcgpu_ctx* ctx, int lambda, double tau_max_in, const double* r_lim, const double* r_grid,
                                      const double* z_grid, const double* z_lim, unsigned char* l_dark_zone, int* ri_in_dark_zone,
                                      int* ri_out_dark_zone, int* zj_sup_dark_zone) {
  int rc = ready(ctx);
  if (rc) return rc;
  if (!r_lim || !r_grid || !z_grid || !z_lim || !l_dark_zone || !ri_in_dark_zone || !ri_out_dark_zone || !zj_sup_dark_zone)
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_define_dark_zone: null argument");
  const DevModel& M = ctx->M;
  if (ctx->voro || M.l3D || M.grid_sph) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "define_dark_zone: 2D cylindrical grids");
  if (lambda < 1 || lambda > M.n_lambda) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_define_dark_zone: bad wavelength");
  HIPCHK(hipSetDevice(ctx->device));
  const int n_rad = M.n_rad, nz = M.nz;
  // kf: kappa(p_icell, lambda) * kappa_factor(icell) per cell (:1467: the cell's own class with lvariable_dust)
  std::vector<double> kf(M.n_cells);
  double kap = 0.0;
  HIPCHK(hipMemcpy(kf.data(), M.kappa_factor, (size_t)M.n_cells * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&kap, M.kappa + (lambda - 1), sizeof(double), hipMemcpyDeviceToHost));
  std::vector<double> kapc(M.n_cells, kap);
  if (M.n_classes) {
    std::vector<int> cls(M.n_cells);
    std::vector<double> vk((size_t)M.n_classes * M.n_lambda);
    HIPCHK(hipMemcpy(cls.data(), M.cell_class, cls.size() * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(vk.data(), M.v_kappa, vk.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int c = 0; c < M.n_cells; ++c) kapc[c] = vk[(size_t)cls[c] * M.n_lambda + (lambda - 1)];
  }
  const float tau_max = (float)tau_max_in;  // real, intent(in)
  // steps 1-3 (:1459-1500): the running sums are default reals
  int ri_in = n_rad, ri_out = 1;
  float total = 0.0f;
  for (int i = 1; i <= n_rad; ++i) {
    total = (float)((double)total + kapc[i - 1] * kf[i - 1] * (r_lim[i] - r_lim[i - 1]));
    if (total > tau_max) { ri_in = i; break; }
  }
  total = 0.0f;
  for (int i = n_rad; i >= 1; --i) {
    total = (float)((double)total + kapc[i - 1] * kf[i - 1] * (r_lim[i] - r_lim[i - 1]));
    if (total > tau_max) { ri_out = i; break; }
  }
  if (ri_out == n_rad) ri_out = n_rad - 1;
  std::vector<int> zj(n_rad, 0);
  for (int i = ri_in; i <= ri_out; ++i) {
    total = 0.0f;
    for (int j = nz; j >= 1; --j) {
      const double dzl = z_lim[(i - 1) + (size_t)n_rad * j] - z_lim[(i - 1) + (size_t)n_rad * (j - 1)];
      total = (float)((double)total + kapc[(i - 1) + (size_t)n_rad * (j - 1)] * kf[(i - 1) + (size_t)n_rad * (j - 1)] * dzl);
      if (total > tau_max) { zj[i - 1] = j; break; }
    }
  }
  // step 4 (:1522-1551) on the device: one ray per thread
  std::memset(l_dark_zone, 0, (size_t)M.n_cells);
  const int i_lo = ri_in > 2 ? ri_in : 2, i_hi = ri_out;
  if (i_hi >= i_lo) {
    DevBuf<int> d_zj;
    DevBuf<double> d_rg, d_zg;
    DevBuf<unsigned char> d_flag, d_now;
    HIPCHK(d_zj.alloc(n_rad)); HIPCHK(d_zj.put(zj.data(), n_rad));
    HIPCHK(d_rg.alloc(M.n_cells)); HIPCHK(d_rg.put(r_grid, M.n_cells));
    HIPCHK(d_zg.alloc(M.n_cells)); HIPCHK(d_zg.put(z_grid, M.n_cells));
    HIPCHK(d_flag.alloc(M.n_cells)); HIPCHK(d_now.alloc(M.n_cells));
    const long long n_rays = 11LL * (i_hi - i_lo + 1) * nz;
    const size_t lds = lds_bytes(M, true);
    HIPCHK(hipFuncSetAttribute((const void*)k_dark_zone_rays, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<unsigned char> flag(M.n_cells), next(M.n_cells);
    // The reference's loop is sequential in the column and reads the flags it has set so far (see k_dark_zone_rays): passes
    // with the previous pass's flags until nothing changes -- the flags only grow, column i depends on the columns before it.
    for (int pass = 0; pass <= i_hi - i_lo + 1; ++pass) {
      HIPCHK(d_now.put(l_dark_zone, M.n_cells));
      HIPCHK(hipMemsetAsync(d_flag.p, 0, M.n_cells, ctx->stream));
      hipLaunchKernelGGL(k_dark_zone_rays, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), lds, ctx->stream, M, lambda, tau_max, i_lo,
                         i_hi, d_zj.p, d_rg.p, d_zg.p, d_now.p, d_flag.p);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(ctx->stream));
      HIPCHK(d_flag.get(flag.data(), M.n_cells));
      // the first flagged cell from the top of the candidate range, and everything below it (:1541-1547)
      std::fill(next.begin(), next.end(), (unsigned char)0);
      for (int i = i_lo; i <= i_hi; ++i)
        for (int j = zj[i - 1]; j >= 1; --j)
          if (flag[(i - 1) + (size_t)n_rad * (j - 1)]) {
            for (int jj = 1; jj <= j; ++jj) next[(i - 1) + (size_t)n_rad * (jj - 1)] = 1;
            break;
          }
      const bool same = std::memcmp(next.data(), l_dark_zone, (size_t)M.n_cells) == 0;
      std::memcpy(l_dark_zone, next.data(), (size_t)M.n_cells);
      if (same) break;
    }
  }
  // (:1621-1628) the extent handed to the diffusion fill
  if (ri_in <= ri_out) {
    for (int i = 1; i < ri_in; ++i) zj[i - 1] = zj[ri_in - 1];
    for (int i = ri_out + 1; i <= n_rad; ++i) zj[i - 1] = zj[ri_out - 1];
  }
  *ri_in_dark_zone = ri_in; *ri_out_dark_zone = ri_out;
  for (int i = 0; i < n_rad; ++i) zj_sup_dark_zone[i] = zj[i];
  return MCGPU_OK;
}

// dust_map with method 2's source function (the inclination of the last mcgpu_rt2_source): see include/mcgpu.h
extern "C" int mcgpu_rt2_dust_map(mcgpu_ctx* ctx, const mcgpu_rt_opts* o, const float* tab_RT_az, const float* Tdust,
                                  double* stokes, double* kernel_ms) {
  if (ctx && !stokes) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_rt2_dust_map: null argument");
  Rt1Job J;
  J.method2 = true;
  int rc = rt1_prepare(ctx, o, tab_RT_az, Tdust, J, "mcgpu_rt2_dust_map: bad option");
  if (rc) return rc;
  const int ntf = ctx->rt2_N_type_flux;
  const size_t n_out = (size_t)J.A.nRT * ntf;
  DevBuf<double> d_out;
  HIPCHK(d_out.alloc(n_out));
  HIPCHK(hipMemsetAsync(d_out.p, 0, n_out * sizeof(double), ctx->stream));
  J.A.out = d_out.p;
  const int n_rays = RT_N_RAD * RT_N_PHI;
  if ((rc = rt1_launch<false>(ctx, J.A, (n_rays + 255) / 256))) return rc;
  if (kernel_ms) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1)); *kernel_ms = ms; }
  std::vector<double> all(n_out);
  HIPCHK(d_out.get(all.data(), n_out));
  for (int t = 0; t < ntf; ++t) stokes[t] = all[(size_t)J.A.q_only * ntf + t];
  return MCGPU_OK;
}

extern "C" int mcgpu_rt2_image(mcgpu_ctx* ctx, const mcgpu_rt_opts* o, const float* tab_RT_az, const float* Tdust,
                               int npix_x, int npix_y, double map_size, double zoom, double* image, uint64_t* n_rays,
                               double* kernel_ms) {
  if (ctx && (!image || npix_x < 1 || npix_y < 1 || npix_x > 32768 || npix_y > 32768 || !(map_size > 0.0) || !(zoom > 0.0)))
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_rt2_image: bad argument");
  Rt1Job J;
  J.method2 = true;
  int rc = rt1_prepare(ctx, o, tab_RT_az, Tdust, J, "mcgpu_rt2_image: bad option");
  if (rc) return rc;
  RtArgs& A = J.A;
  const int ntf = ctx->rt2_N_type_flux;
  A.npix_x = npix_x; A.npix_y = npix_y;
  A.npix_x_max = o->l_sym_ima ? npix_x / 2 + npix_x % 2 : npix_x;
  A.taille_pix = (map_size / zoom) / (double)(npix_x > npix_y ? npix_x : npix_y);
  const size_t n_all = (size_t)A.nRT * ntf * npix_x * npix_y;
  DevBuf<double> d_img;
  DevBuf<unsigned long long> d_rays;
  HIPCHK(d_img.alloc(n_all)); HIPCHK(d_rays.alloc(1));
  HIPCHK(hipMemsetAsync(d_img.p, 0, n_all * sizeof(double), ctx->stream));
  HIPCHK(hipMemsetAsync(d_rays.p, 0, sizeof(unsigned long long), ctx->stream));
  A.image = d_img.p; A.n_rays = d_rays.p;
  const long n_waves = (long)A.npix_x_max * npix_y;   // one wavefront per pixel
  long blocks = (n_waves + 3) / 4;
  if (blocks > 65536) blocks = 65536;
  if ((rc = rt1_launch<true>(ctx, A, (int)blocks))) return rc;
  if (kernel_ms) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1)); *kernel_ms = ms; }
  std::vector<double> all(n_all);
  HIPCHK(d_img.get(all.data(), n_all));
  // image(npix_x, npix_y, N_type_flux) of the inclination: the slice (ibin, iaz = 1) of the full layout
  const int n_az = A.nRT / A.RT_n_incl, ibin0 = A.q_only % A.RT_n_incl, iaz0 = A.q_only / A.RT_n_incl;
  for (int t = 0; t < ntf; ++t)
    std::memcpy(image + (size_t)t * npix_x * npix_y,
                all.data() + ((((size_t)t * n_az + iaz0) * A.RT_n_incl + ibin0) * npix_y) * npix_x, (size_t)npix_x * npix_y * sizeof(double));
  if (n_rays) { unsigned long long r = 0; HIPCHK(d_rays.get(&r, 1)); *n_rays = r; }
  return MCGPU_OK;
}

// compute_stars_map for the SED (dust_transfer.f90:1604-1854): see include/mcgpu.h
extern "C" int mcgpu_rt1_stars_map_sed(mcgpu_ctx* ctx, const mcgpu_rt_opts* o, const float* tab_RT_az, uint64_t seed,
                                       const double* star_flux, double* stars_flux) {
  int rc = ready(ctx);
  if (rc) return rc;
  if (!o || !tab_RT_az || !star_flux || !stars_flux) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_rt1_stars_map_sed: null argument");
  const DevModel& M = ctx->M;
  if (!ctx->have_rt1) return fail(ctx, MCGPU_ERR_STATE, "stars map: set the observers first (mcgpu_set_rt1)");
  if (o->lambda < 1 || o->lambda > M.n_lambda) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_rt1_stars_map_sed: bad option");
  HIPCHK(hipSetDevice(ctx->device));
  const int nRT = ctx->RT_n_incl * ctx->RT_n_az;
  DevBuf<float> d_az;
  DevBuf<double> d_flux, d_out;
  HIPCHK(d_az.alloc(ctx->RT_n_az)); HIPCHK(d_az.put(tab_RT_az, ctx->RT_n_az));
  HIPCHK(d_flux.alloc(M.n_stars)); HIPCHK(d_flux.put(star_flux, M.n_stars));
  HIPCHK(d_out.alloc(nRT)); HIPCHK(hipMemsetAsync(d_out.p, 0, nRT * sizeof(double), ctx->stream));
  RtArgs A;
  std::memset(&A, 0, sizeof(A));
  A.lambda = o->lambda; A.RT_n_incl = ctx->RT_n_incl; A.nRT = nRT; A.ang_disque = o->ang_disque;
  A.rt_u = ctx->d_rt_u; A.rt_v = ctx->d_rt_v; A.rt_w = ctx->d_rt_w; A.rt_az = d_az.p;
  DevBuf<VoroGrid> d_V;   // Voronoi grid: the kernels pick optical_length_tot_voro when they are handed the grid's record
  if (ctx->voro) { HIPCHK(d_V.alloc(1)); HIPCHK(d_V.put(&ctx->V, 1)); A.voro = d_V.p; }
  const size_t lds = lds_bytes(M, true);
  const unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
  if (M.l3D) {
    HIPCHK(hipFuncSetAttribute((const void*)k_stars_map_sed<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_stars_map_sed<true>, dim3(nRT * M.n_stars), dim3(512), lds, ctx->stream, M, A, k0, k1, d_flux.p, d_out.p);
  } else {
    HIPCHK(hipFuncSetAttribute((const void*)k_stars_map_sed<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_stars_map_sed<false>, dim3(nRT * M.n_stars), dim3(512), lds, ctx->stream, M, A, k0, k1, d_flux.p, d_out.p);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(d_out.get(stars_flux, nRT));
  return MCGPU_OK;
}

// compute_stars_map for images (resolved discs, limb darkening): see include/mcgpu.h
extern "C" int mcgpu_rt1_stars_map_image(mcgpu_ctx* ctx, const mcgpu_rt_opts* o, const float* tab_RT_az, uint64_t seed,
                                         const double* star_flux, int npix_x, int npix_y, double map_size, double zoom, int n_mu,
                                         const float* mu_limb_darkening, const float* limb_darkening,
                                         const float* pola_limb_darkening, double* stars_map, double* star_position) {
  int rc = ready(ctx);
  if (rc) return rc;
  if (!o || !tab_RT_az || !star_flux || !stars_map || npix_x < 1 || npix_y < 1 || npix_x > 32768 || npix_y > 32768 ||
      !(map_size > 0.0) || !(zoom > 0.0) || n_mu < 0 || (n_mu > 0 && (n_mu < 2 || !mu_limb_darkening || !limb_darkening)))
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_rt1_stars_map_image: bad argument");
  const DevModel& M = ctx->M;
  if (!ctx->have_rt1) return fail(ctx, MCGPU_ERR_STATE, "stars map: set the observers first (mcgpu_set_rt1)");
  if (o->lambda < 1 || o->lambda > M.n_lambda || !(o->distance > 0.0)) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_rt1_stars_map_image: bad option");
  HIPCHK(hipSetDevice(ctx->device));
  const int nRT = ctx->RT_n_incl * ctx->RT_n_az;
  const int n_maps = (n_mu > 0 && pola_limb_darkening) ? 3 : 1;
  const size_t n_out = (size_t)npix_x * npix_y * n_maps * nRT;
  DevBuf<float> d_az, d_mu, d_ld, d_pld;
  DevBuf<double> d_flux, d_map, d_pos;
  HIPCHK(d_az.alloc(ctx->RT_n_az)); HIPCHK(d_az.put(tab_RT_az, ctx->RT_n_az));
  HIPCHK(d_flux.alloc(M.n_stars)); HIPCHK(d_flux.put(star_flux, M.n_stars));
  HIPCHK(d_map.alloc(n_out)); HIPCHK(hipMemsetAsync(d_map.p, 0, n_out * sizeof(double), ctx->stream));
  HIPCHK(d_pos.alloc((size_t)M.n_stars * nRT * 2));
  if (n_mu > 0) {
    HIPCHK(d_mu.alloc(n_mu)); HIPCHK(d_mu.put(mu_limb_darkening, n_mu));
    HIPCHK(d_ld.alloc(n_mu)); HIPCHK(d_ld.put(limb_darkening, n_mu));
    if (pola_limb_darkening) { HIPCHK(d_pld.alloc(n_mu)); HIPCHK(d_pld.put(pola_limb_darkening, n_mu)); }
  }
  RtArgs A;
  std::memset(&A, 0, sizeof(A));
  A.lambda = o->lambda; A.RT_n_incl = ctx->RT_n_incl; A.nRT = nRT; A.ang_disque = o->ang_disque;
  A.rt_u = ctx->d_rt_u; A.rt_v = ctx->d_rt_v; A.rt_w = ctx->d_rt_w; A.rt_az = d_az.p;
  DevBuf<VoroGrid> d_V;   // Voronoi grid: the kernels pick optical_length_tot_voro when they are handed the grid's record
  if (ctx->voro) { HIPCHK(d_V.alloc(1)); HIPCHK(d_V.put(&ctx->V, 1)); A.voro = d_V.p; }
  StarsImageArgs I;
  I.npix_x = npix_x; I.npix_y = npix_y; I.n_maps = n_maps; I.n_mu = n_mu;
  I.taille_pix = (map_size / zoom) / (double)(npix_x > npix_y ? npix_x : npix_y);
  I.pix_size = (float)(map_size / zoom / (double)(npix_x > npix_y ? npix_x : npix_y));
  I.distance = o->distance;
  I.mu_ld = d_mu.p; I.ld = d_ld.p; I.pola_ld = (n_mu > 0 && pola_limb_darkening) ? d_pld.p : nullptr;
  I.map = d_map.p; I.star_position = d_pos.p;
  const size_t lds = lds_bytes(M, true);
  const unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
  if (M.l3D) {
    HIPCHK(hipFuncSetAttribute((const void*)k_stars_map_image<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_stars_map_image<true>, dim3(nRT * M.n_stars), dim3(512), lds, ctx->stream, M, A, I, k0, k1, d_flux.p);
  } else {
    HIPCHK(hipFuncSetAttribute((const void*)k_stars_map_image<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_stars_map_image<false>, dim3(nRT * M.n_stars), dim3(512), lds, ctx->stream, M, A, I, k0, k1, d_flux.p);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(d_map.get(stars_map, n_out));
  if (star_position) HIPCHK(d_pos.get(star_position, (size_t)M.n_stars * nRT * 2));
  return MCGPU_OK;
}

extern "C" int mcgpu_rt1_image(mcgpu_ctx* ctx, const mcgpu_rt_opts* o, const float* tab_RT_az, const float* Tdust,
                               int npix_x, int npix_y, double map_size, double zoom, double* image, uint64_t* n_rays,
                               double* kernel_ms) {
  if (ctx && (!image || npix_x < 1 || npix_y < 1 || npix_x > 32768 || npix_y > 32768 || !(map_size > 0.0) || !(zoom > 0.0)))
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_rt1_image: bad argument");
  Rt1Job J;
  int rc = rt1_prepare(ctx, o, tab_RT_az, Tdust, J, "mcgpu_rt1_image: bad option");
  if (rc) return rc;
  RtArgs& A = J.A;
  A.npix_x = npix_x; A.npix_y = npix_y;
  A.npix_x_max = o->l_sym_ima ? npix_x / 2 + npix_x % 2 : npix_x;               // dust_transfer.f90:1553-1557
  A.taille_pix = (map_size / zoom) / (double)(npix_x > npix_y ? npix_x : npix_y);  // (:1545)
  const size_t n_out = (size_t)A.nRT * ctx->N_type_flux * npix_x * npix_y;
  DevBuf<double> d_img;
  DevBuf<unsigned long long> d_rays;
  HIPCHK(d_img.alloc(n_out)); HIPCHK(d_rays.alloc(1));
  HIPCHK(hipMemsetAsync(d_img.p, 0, n_out * sizeof(double), ctx->stream));
  HIPCHK(hipMemsetAsync(d_rays.p, 0, sizeof(unsigned long long), ctx->stream));
  A.image = d_img.p; A.n_rays = d_rays.p;
  const long n_pix = (long)A.nRT * A.npix_x_max * npix_y;                        // one wavefront per pixel
  long blocks = (n_pix + 3) / 4;
  if (blocks > 65536) blocks = 65536;
  if ((rc = rt1_launch<true>(ctx, A, (int)blocks))) return rc;
  if (kernel_ms) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1)); *kernel_ms = ms; }
  HIPCHK(d_img.get(image, n_out));
  if (n_rays) { unsigned long long r = 0; HIPCHK(d_rays.get(&r, 1)); *n_rays = r; }
  return MCGPU_OK;
}

// compute_tau_map / compute_tau_surface_map: see include/mcgpu.h
extern "C" int mcgpu_tau_maps(mcgpu_ctx* ctx, const mcgpu_rt_opts* o, const float* tab_RT_az, int npix_x, int npix_y,
                              double map_size, double zoom, double tau_surface, float* tau_map, float* tau_surface_map,
                              double* kernel_ms) {
  int rc = ready(ctx);
  if (rc) return rc;
  if (!o || !tab_RT_az || (!tau_map && !tau_surface_map) || npix_x < 1 || npix_y < 1 || npix_x > 32768 || npix_y > 32768 ||
      !(map_size > 0.0) || !(zoom > 0.0) || (tau_surface_map && !(tau_surface > 0.0)))
    return fail(ctx, MCGPU_ERR_ARG, "mcgpu_tau_maps: bad argument");
  const DevModel& M = ctx->M;
  if (!ctx->have_rt1) return fail(ctx, MCGPU_ERR_STATE, "optical-depth maps: set the observers first (mcgpu_set_rt1)");
  if (o->lambda < 1 || o->lambda > M.n_lambda || !(o->Rmax > 0.0)) return fail(ctx, MCGPU_ERR_ARG, "mcgpu_tau_maps: bad option");
  HIPCHK(hipSetDevice(ctx->device));
  const int nRT = ctx->RT_n_incl * ctx->RT_n_az;
  const size_t n_pix = (size_t)npix_x * npix_y * nRT;
  DevBuf<float> d_az, d_tau, d_surf;
  HIPCHK(d_az.alloc(ctx->RT_n_az)); HIPCHK(d_az.put(tab_RT_az, ctx->RT_n_az));
  if (tau_map) HIPCHK(d_tau.alloc(n_pix));
  if (tau_surface_map) HIPCHK(d_surf.alloc(3 * n_pix));
  RtArgs A;
  std::memset(&A, 0, sizeof(A));
  A.lambda = o->lambda; A.RT_n_incl = ctx->RT_n_incl; A.nRT = nRT; A.ang_disque = o->ang_disque;
  A.rt_u = ctx->d_rt_u; A.rt_v = ctx->d_rt_v; A.rt_w = ctx->d_rt_w; A.rt_az = d_az.p;
  A.npix_x = npix_x; A.npix_y = npix_y;
  A.taille_pix = (map_size / zoom) / (double)(npix_x > npix_y ? npix_x : npix_y);  // (:2053, :2161)
  A.l_far = 10.0 * o->Rmax;                                                         // (:2046, :2154)
  const size_t lds = lds_bytes(M, true);
  long blocks = (long)((n_pix + 255) / 256);
  if (blocks > 65536) blocks = 65536;
  HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
  if (ctx->voro) {
    HIPCHK(hipFuncSetAttribute((const void*)k_tau_maps_voro, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_tau_maps_voro, dim3(blocks), dim3(256), lds, ctx->stream, M, A, ctx->V, (float)tau_surface, d_tau.p, d_surf.p);
  } else if (M.l3D) {
    HIPCHK(hipFuncSetAttribute((const void*)k_tau_maps<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_tau_maps<true>, dim3(blocks), dim3(256), lds, ctx->stream, M, A, (float)tau_surface, d_tau.p, d_surf.p);
  } else {
    HIPCHK(hipFuncSetAttribute((const void*)k_tau_maps<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_tau_maps<false>, dim3(blocks), dim3(256), lds, ctx->stream, M, A, (float)tau_surface, d_tau.p, d_surf.p);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (kernel_ms) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1)); *kernel_ms = ms; }
  if (tau_map) HIPCHK(d_tau.get(tau_map, n_pix));
  if (tau_surface_map) HIPCHK(d_surf.get(tau_surface_map, 3 * n_pix));
  return MCGPU_OK;
}

extern "C" int mcgpu_probe_cross_cell(mcgpu_ctx* ctx, int n, const double* x0, const double* y0, const double* z0,
                                      const double* u, const double* v, const double* w, const int* cell,
                                      double* x1, double* y1, double* z1, int* next_cell, double* l) {
  int rc = ready(ctx);
  if (rc) return rc;
  if (ctx->voro) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "cylindrical probe on a Voronoi grid");
  HIPCHK(hipSetDevice(ctx->device));
  const DevModel& M = ctx->M;
  DevBuf<double> in[6], out[4];
  DevBuf<int> dc, dn;
  const double* hin[6] = {x0, y0, z0, u, v, w};
  for (int q = 0; q < 6; ++q) { HIPCHK(in[q].alloc(n)); HIPCHK(in[q].put(hin[q], n)); }
  for (int q = 0; q < 4; ++q) HIPCHK(out[q].alloc(n));
  HIPCHK(dc.alloc(n)); HIPCHK(dc.put(cell, n)); HIPCHK(dn.alloc(n));
  const size_t lds = lds_bytes(M);
#define PROBE_CROSS(a, b) do {                                                                                       \
    hipFuncSetAttribute((const void*)k_probe_cross<a, b>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
    hipLaunchKernelGGL((k_probe_cross<a, b>), dim3(64), dim3(256), lds, ctx->stream, M, n, in[0].p, in[1].p, in[2].p,  \
                       in[3].p, in[4].p, in[5].p, ctx->d_cmi, ctx->d_cmj, ctx->d_cmk, dc.p, out[0].p, out[1].p,       \
                       out[2].p, dn.p, out[3].p);                                                                     \
  } while (0)
  if (M.grid_sph) { if (M.l3D) PROBE_CROSS(true, true); else PROBE_CROSS(false, true); }
  else { if (M.l3D) PROBE_CROSS(true, false); else PROBE_CROSS(false, false); }
#undef PROBE_CROSS
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(out[0].get(x1, n)); HIPCHK(out[1].get(y1, n)); HIPCHK(out[2].get(z1, n)); HIPCHK(out[3].get(l, n));
  HIPCHK(dn.get(next_cell, n));
  return MCGPU_OK;
}

extern "C" int mcgpu_probe_cross_voronoi(mcgpu_ctx* ctx, int n, const double* x0, const double* y0,
                                         const double* z0, const double* u, const double* v, const double* w,
                                         const int* cell, const int* previous_cell, double* x1, double* y1,
                                         double* z1, int* next_cell, double* l, double* l_contrib,
                                         double* l_void_before) {
  int rc = ready(ctx);
  if (rc) return rc;
  if (!ctx->voro) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "Voronoi probe on a cylindrical grid");
  HIPCHK(hipSetDevice(ctx->device));
  DevBuf<double> in[6], out[6];
  DevBuf<int> dc, dp, dn;
  const double* hin[6] = {x0, y0, z0, u, v, w};
  for (int q = 0; q < 6; ++q) { HIPCHK(in[q].alloc(n)); HIPCHK(in[q].put(hin[q], n)); }
  for (int q = 0; q < 6; ++q) HIPCHK(out[q].alloc(n));
  HIPCHK(dc.alloc(n)); HIPCHK(dc.put(cell, n));
  HIPCHK(dp.alloc(n)); HIPCHK(dp.put(previous_cell, n));
  HIPCHK(dn.alloc(n));
  hipLaunchKernelGGL(k_probe_cross_voro, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->M, ctx->V, n,
                     in[0].p, in[1].p, in[2].p, in[3].p, in[4].p, in[5].p, dc.p, dp.p, out[0].p, out[1].p, out[2].p,
                     dn.p, out[3].p, out[4].p, out[5].p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(out[0].get(x1, n)); HIPCHK(out[1].get(y1, n)); HIPCHK(out[2].get(z1, n)); HIPCHK(out[3].get(l, n));
  HIPCHK(out[4].get(l_contrib, n)); HIPCHK(out[5].get(l_void_before, n));
  HIPCHK(dn.get(next_cell, n));
  return MCGPU_OK;
}

extern "C" int mcgpu_probe_index_cell(mcgpu_ctx* ctx, int n, const double* x, const double* y, const double* z,
                                      int* icell) {
  int rc = ready(ctx);
  if (rc) return rc;
  if (ctx->voro) return fail(ctx, MCGPU_ERR_UNSUPPORTED, "cylindrical probe on a Voronoi grid");
  HIPCHK(hipSetDevice(ctx->device));
  const DevModel& M = ctx->M;
  DevBuf<double> in[3];
  DevBuf<int> dn;
  const double* hin[3] = {x, y, z};
  for (int q = 0; q < 3; ++q) { HIPCHK(in[q].alloc(n)); HIPCHK(in[q].put(hin[q], n)); }
  HIPCHK(dn.alloc(n));
  const size_t lds = lds_bytes(M);
#define PROBE_INDEX(a, b) do {                                                                                       \
    hipFuncSetAttribute((const void*)k_probe_index<a, b>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
    hipLaunchKernelGGL((k_probe_index<a, b>), dim3(64), dim3(256), lds, ctx->stream, M, n, in[0].p, in[1].p, in[2].p, dn.p); \
  } while (0)
  if (M.grid_sph) { if (M.l3D) PROBE_INDEX(true, true); else PROBE_INDEX(false, true); }
  else { if (M.l3D) PROBE_INDEX(true, false); else PROBE_INDEX(false, false); }
#undef PROBE_INDEX
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(dn.get(icell, n));
  return MCGPU_OK;
}

extern "C" int mcgpu_probe_philox(mcgpu_ctx* ctx, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  if (!ctx) return MCGPU_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  DevBuf<uint32_t> d;
  HIPCHK(d.alloc(4));
  hipLaunchKernelGGL(k_probe_philox, dim3(1), dim3(1), 0, ctx->stream, ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], d.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(d.get(out, 4));
  return MCGPU_OK;
}

extern "C" int mcgpu_probe_packet_rand(mcgpu_ctx* ctx, uint64_t seed, uint64_t packet, int n, float* out) {
  if (!ctx || n < 1) return MCGPU_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  DevBuf<float> d;
  HIPCHK(d.alloc(n));
  hipLaunchKernelGGL(k_probe_rand, dim3(1), dim3(1), 0, ctx->stream, seed, packet, n, d.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(d.get(out, n));
  return MCGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// Several GPUs behind ONE host thread (the reference's host is one OpenMP process: mcfost.f90,
// mcfost2phantom.f90:159).  Packets shard by id range, tables are replicated (the host calls the
// setters on every context), and ONE ncclAllReduce per temperature iteration sums the fused
// accumulator [E_abs | sed | n_sent | counters] over xGMI.  No other exchange.
// ---------------------------------------------------------------------------------------------
struct mcgpu_multi {
  int n_dev = 0;
  std::vector<int> devs;
  std::vector<mcgpu_ctx*> ctx;
  std::vector<ncclComm_t> comm;   // empty until the first collective (one device never needs them)
  bool shared = false;            // MCGPU_MULTI_SHARED_DEVICE: every context on ONE device, sums by k_sum_into (RCCL refuses duplicate devices)
  bool force_rccl = false;        // MCGPU_MULTI_FORCE_RCCL: the communicator and the all-reduce also with ONE device (a sum over one rank)
  std::vector<hipEvent_t> ev;     // shared mode: one event per context for the cross-stream ordering of the in-library sum
  unsigned long long n_reduce = 0; // collectives executed (either kind)
  bool reduced = false;           // the accumulators of every device hold the all-reduced totals of the last call
  int reduced_what = 0;           // ... and which of xI_scatt / I_spec do (MULTI_XI | MULTI_ISPEC)
  std::string err;
};

__global__ void k_scale_f64(double* a, size_t n, double f) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] *= f;
}
__global__ void k_scale_f32(float* a, size_t n, float f) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] *= f;
}

__global__ void k_sum_into(double* a, const double* b, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += b[i];
}
__global__ void k_sum_into_f32(float* a, const float* b, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += b[i];
}

// flags = MCGPU_MULTI_SHARED_DEVICE: the n_dev contexts all live on ONE device (devices[i] all equal; NULL: device 0)
// and the reduction is the library's own sum kernel -- RCCL refuses a communicator with a device twice.  This is how
// the sharding, rescaling, counter and error logic of n_dev > 1 is executed on a box with one GPU (tests, dry runs);
// a production host passes flags = 0 and distinct devices.
// flags = MCGPU_MULTI_FORCE_RCCL: open the RCCL communicator and run the grouped ncclAllReduce of every buffer also when
// n_dev = 1 (ncclCommInitAll over one rank; the sum over one rank leaves the buffers as they are), so that the
// collective path -- counters into the accumulator's tail, the group call, counters back -- executes on a box with one
// GPU.  Not with MCGPU_MULTI_SHARED_DEVICE (no communicator exists there).
extern "C" int mcgpu_multi_create_ex(int n_dev, const int* devices, unsigned int flags, mcgpu_multi** out) {
  if (!out || n_dev < 1 || (flags & ~(unsigned int)(MCGPU_MULTI_SHARED_DEVICE | MCGPU_MULTI_FORCE_RCCL))) return MCGPU_ERR_ARG;
  *out = nullptr;
  const bool shared = (flags & MCGPU_MULTI_SHARED_DEVICE) != 0;
  if (shared && (flags & MCGPU_MULTI_FORCE_RCCL)) return MCGPU_ERR_ARG;
  int n_have = 0;
  if (hipGetDeviceCount(&n_have) != hipSuccess || n_have <= 0) return MCGPU_ERR_NO_DEVICE;
  std::vector<int> devs(n_dev);
  for (int i = 0; i < n_dev; ++i) {
    devs[i] = devices ? devices[i] : (shared ? 0 : i);
    if (devs[i] < 0 || devs[i] >= n_have) return MCGPU_ERR_ARG;
    for (int j = 0; j < i; ++j) if ((devs[j] == devs[i]) != shared) return MCGPU_ERR_ARG;
  }
  mcgpu_multi* mm = new mcgpu_multi();
  mm->n_dev = n_dev;
  mm->devs = devs;
  mm->shared = shared;
  mm->force_rccl = (flags & MCGPU_MULTI_FORCE_RCCL) != 0;
  mm->ctx.assign(n_dev, nullptr);
  for (int i = 0; i < n_dev; ++i) {
    const int rc = mcgpu_create(devs[i], &mm->ctx[i]);
    if (rc) { for (int j = 0; j < i; ++j) mcgpu_destroy(mm->ctx[j]); delete mm; return rc; }
  }
  if (shared) {
    mm->ev.assign(n_dev, nullptr);
    hipSetDevice(devs[0]);
    for (int i = 0; i < n_dev; ++i)
      if (hipEventCreateWithFlags(&mm->ev[i], hipEventDisableTiming) != hipSuccess) {
        for (int j = 0; j < i; ++j) hipEventDestroy(mm->ev[j]);
        for (int j = 0; j < n_dev; ++j) mcgpu_destroy(mm->ctx[j]);
        delete mm;
        return MCGPU_ERR_HIP;
      }
  }
  *out = mm;
  return MCGPU_OK;
}

extern "C" int mcgpu_multi_create(int n_dev, const int* devices, mcgpu_multi** out) {
  return mcgpu_multi_create_ex(n_dev, devices, 0u, out);
}

// the RCCL communicators, created by the first call that has something to reduce (n_dev > 1, or forced)
static int multi_comms(mcgpu_multi* mm) {
  if ((mm->n_dev < 2 && !mm->force_rccl) || mm->shared || !mm->comm.empty()) return MCGPU_OK;
  mm->comm.assign(mm->n_dev, nullptr);
  if (ncclCommInitAll(mm->comm.data(), mm->n_dev, mm->devs.data()) != ncclSuccess) {
    mm->comm.clear();
    mm->err = "ncclCommInitAll failed";
    return MCGPU_ERR_HIP;
  }
  return MCGPU_OK;
}

extern "C" int mcgpu_multi_destroy(mcgpu_multi* mm) {
  if (!mm) return MCGPU_OK;
  for (int i = 0; i < mm->n_dev; ++i) {
    if (mm->ctx[i]) { hipSetDevice(mm->ctx[i]->device); hipDeviceSynchronize(); }
    if (i < (int)mm->comm.size() && mm->comm[i]) ncclCommDestroy(mm->comm[i]);
    if (i < (int)mm->ev.size() && mm->ev[i]) hipEventDestroy(mm->ev[i]);
  }
  for (int i = 0; i < mm->n_dev; ++i) mcgpu_destroy(mm->ctx[i]);
  delete mm;
  return MCGPU_OK;
}

extern "C" int mcgpu_multi_size(const mcgpu_multi* mm) { return mm ? mm->n_dev : 0; }
extern "C" mcgpu_ctx* mcgpu_multi_ctx(mcgpu_multi* mm, int i) { return (mm && i >= 0 && i < mm->n_dev) ? mm->ctx[i] : nullptr; }
extern "C" const char* mcgpu_multi_last_error(const mcgpu_multi* mm) { return mm ? mm->err.c_str() : "null handle"; }
// ranks of the handle's RCCL communicator as RCCL reports them (0: no communicator yet -- one device, or nothing reduced so far)
extern "C" int mcgpu_multi_rccl_ranks(mcgpu_multi* mm) {
  if (!mm || mm->comm.empty() || !mm->comm[0]) return 0;
  int n = 0;
  return ncclCommCount(mm->comm[0], &n) == ncclSuccess ? n : -1;
}
// collectives this handle has executed so far (RCCL all-reduces, or the shared-device sums that stand in for them)
extern "C" uint64_t mcgpu_multi_reductions(const mcgpu_multi* mm) { return mm ? mm->n_reduce : 0; }

// the shard of device i: contiguous, disjoint, exhaustive (same rule as mcfost_amd/distributed.py::shard_packets)
extern "C" void mcgpu_shard_packets(uint64_t n_packets, int rank, int world, uint64_t* first, uint64_t* count) {
  const uint64_t base = n_packets / (uint64_t)world, rem = n_packets % (uint64_t)world;
  if (count) *count = base + ((uint64_t)rank < rem ? 1u : 0u);
  if (first) *first = (uint64_t)rank * base + ((uint64_t)rank < rem ? (uint64_t)rank : rem);
}

static void multi_drain(mcgpu_multi* mm) {  // after an error: nothing of this call is left running on any device
  for (int i = 0; i < mm->n_dev; ++i) { hipSetDevice(mm->ctx[i]->device); hipStreamSynchronize(mm->ctx[i]->stream); }
}

// What a collective sums besides the fused accumulator: the SED step's ray-tracing deposits
enum { MULTI_XI = 1, MULTI_ISPEC = 2 };
struct RedBuf { void* p; size_t n; bool f32; };
static std::vector<RedBuf> multi_bufs(mcgpu_ctx* c, int what, bool accum) {
  std::vector<RedBuf> v;
  if (accum && c->d_accum) v.push_back({c->d_accum, c->n_accum, false});
  if ((what & MULTI_XI) && c->d_xI) v.push_back({c->d_xI, xi_dev_values(c), c->xI_bytes == 4});
  if ((what & MULTI_ISPEC) && c->d_I_spec) {   // ray tracing method 2 (radiation_field.f90:91-129)
    v.push_back({c->d_I_spec, (size_t)c->M.n_cells * c->n_phi_I * c->n_theta_I * XI_LINE, false});
    v.push_back({c->d_I_spec_star, (size_t)c->M.n_cells, false});
  }
  return v;
}
static inline int multi_what(const mcgpu_mono_opts* o) { return o->rt1 == 1 ? MULTI_XI : (o->rt1 == 2 ? MULTI_ISPEC : 0); }

// An accumulating call on several devices: after the last call's in-place all-reduce EVERY device holds the global
// sums G, and adding this call's local parts L_i to n copies of G would reduce to n G + sum L_i.  So each device first
// scales what it holds by 1/n -- n (G / n + L_i) = G + n L_i is also exactly what the in-flight temperature's
// `local * n_replicas` should see -- and devices > 0 clear their event counters (integers; device 0 keeps the totals).
// Exact for 2, 4, 8 devices (a power of two); otherwise G / n carries one rounding.  Only buffers that HOLD reduced
// totals are scaled (reduced_what: which of xI_scatt / I_spec the last collective summed).
static int multi_prepare_accumulate(mcgpu_multi* mm, int what) {
  const int n = mm->n_dev;
  if (n < 2) return MCGPU_OK;
  for (int i = 0; i < n; ++i) {
    mcgpu_ctx* c = mm->ctx[i];
    if (hipSetDevice(c->device) != hipSuccess) return MCGPU_ERR_HIP;
    if (mm->reduced && c->d_accum && i > 0 &&
        hipMemsetAsync(c->d_counters, 0, MCGPU_N_COUNTERS * sizeof(unsigned long long), c->stream) != hipSuccess) return MCGPU_ERR_HIP;
    for (const RedBuf& b : multi_bufs(c, what & mm->reduced_what, mm->reduced)) {
      const dim3 g((unsigned)((b.n + 255) / 256));
      if (b.f32) hipLaunchKernelGGL(k_scale_f32, g, dim3(256), 0, c->stream, reinterpret_cast<float*>(b.p), b.n, 1.0f / n);
      else hipLaunchKernelGGL(k_scale_f64, g, dim3(256), 0, c->stream, reinterpret_cast<double*>(b.p), b.n, 1.0 / n);
    }
    if (hipGetLastError() != hipSuccess) return MCGPU_ERR_HIP;
  }
  return MCGPU_OK;
}

// ONE all-reduce of the fused accumulator (counters in its tail), in place, on every device's own stream; `what`:
// and one of xI_scatt (ray tracing method 1) or of I_spec + I_spec_star (method 2)
static int multi_allreduce(mcgpu_multi* mm, int what) {
  const int n = mm->n_dev;
  auto failed = [&](int i, int rc) { mm->err = "device " + std::to_string(i) + ": " + mcgpu_last_error(mm->ctx[i]); return rc; };
  if (n < 2 && !mm->force_rccl) return MCGPU_OK;
  int rc = multi_comms(mm);
  if (rc) return rc;
  for (int i = 0; i < n; ++i) if ((rc = mcgpu_counters_to_accum(mm->ctx[i]))) return failed(i, rc);
  std::vector<std::vector<RedBuf>> bufs(n);
  for (int i = 0; i < n; ++i) {
    bufs[i] = multi_bufs(mm->ctx[i], what, true);
    bool same = bufs[i].size() == bufs[0].size();
    for (size_t q = 0; same && q < bufs[i].size(); ++q) same = bufs[i][q].n == bufs[0][q].n && bufs[i][q].f32 == bufs[0][q].f32;
    if (!same) { mm->err = "the contexts of the handle do not hold the same model"; return MCGPU_ERR_STATE; }
  }
  if (mm->shared) {
    // every context on one device: context 0's stream waits for the others, sums their buffers into its own, and the
    // others copy the totals back -- what the in-place all-reduce leaves behind, by plain kernels
    mcgpu_ctx* c0 = mm->ctx[0];
    if (hipSetDevice(c0->device) != hipSuccess) return MCGPU_ERR_HIP;
    bool bad = false;
    for (int i = 1; i < n && !bad; ++i) {
      bad = hipEventRecord(mm->ev[i], mm->ctx[i]->stream) != hipSuccess || hipStreamWaitEvent(c0->stream, mm->ev[i], 0) != hipSuccess;
      for (size_t q = 0; q < bufs[0].size() && !bad; ++q) {
        const RedBuf &a = bufs[0][q], &b = bufs[i][q];
        const dim3 g((unsigned)((a.n + 255) / 256));
        if (a.f32) hipLaunchKernelGGL(k_sum_into_f32, g, dim3(256), 0, c0->stream, reinterpret_cast<float*>(a.p), reinterpret_cast<const float*>(b.p), a.n);
        else hipLaunchKernelGGL(k_sum_into, g, dim3(256), 0, c0->stream, reinterpret_cast<double*>(a.p), reinterpret_cast<const double*>(b.p), a.n);
        bad = hipGetLastError() != hipSuccess;
      }
    }
    bad = bad || hipEventRecord(mm->ev[0], c0->stream) != hipSuccess;
    for (int i = 1; i < n && !bad; ++i) {
      mcgpu_ctx* c = mm->ctx[i];
      bad = hipStreamWaitEvent(c->stream, mm->ev[0], 0) != hipSuccess;
      for (size_t q = 0; q < bufs[0].size() && !bad; ++q)
        bad = hipMemcpyAsync(bufs[i][q].p, bufs[0][q].p, bufs[0][q].n * (bufs[0][q].f32 ? 4 : 8), hipMemcpyDeviceToDevice, c->stream) != hipSuccess;
      // (context 0 must not start its next launch before the others have read its totals)
      bad = bad || hipEventRecord(mm->ev[i], c->stream) != hipSuccess || hipStreamWaitEvent(c0->stream, mm->ev[i], 0) != hipSuccess;
    }
    if (bad) { mm->err = "shared-device reduction failed"; return MCGPU_ERR_HIP; }
  } else {
    if (ncclGroupStart() != ncclSuccess) { mm->err = "ncclGroupStart failed"; return MCGPU_ERR_HIP; }
    bool bad = false;
    for (int i = 0; i < n && !bad; ++i) {
      mcgpu_ctx* c = mm->ctx[i];
      hipSetDevice(c->device);
      for (size_t q = 0; q < bufs[i].size() && !bad; ++q)
        bad = ncclAllReduce(bufs[i][q].p, bufs[i][q].p, bufs[i][q].n, bufs[i][q].f32 ? ncclFloat : ncclDouble, ncclSum, mm->comm[i], c->stream) != ncclSuccess;
    }
    if (ncclGroupEnd() != ncclSuccess || bad) { mm->err = "ncclAllReduce failed"; return MCGPU_ERR_HIP; }
  }
  for (int i = 0; i < n; ++i) if ((rc = mcgpu_counters_from_accum(mm->ctx[i]))) return failed(i, rc);
  mm->reduced = true;
  mm->reduced_what |= what;
  mm->n_reduce++;
  return MCGPU_OK;
}

extern "C" int mcgpu_multi_run_thermal(mcgpu_multi* mm, const mcgpu_run_opts* opts, double* E_abs, double* sed,
                                       double* n_sent, uint64_t* counters, double* kernel_ms) {
  if (!mm || !opts) return MCGPU_ERR_ARG;
  const int n = mm->n_dev;
  auto failed = [&](int i, int rc) { mm->err = "device " + std::to_string(i) + ": " + mcgpu_last_error(mm->ctx[i]); multi_drain(mm); return rc; };
  int rc;
  if (opts->accumulate) { if ((rc = multi_prepare_accumulate(mm, 0))) { multi_drain(mm); return rc; } }
  else mm->reduced = false;
  // 1) every device runs its shard of the packet ids; the in-flight temperature scales the local partial sum by
  //    the number of replicas (thermal_emission.f90:670)
  for (int i = 0; i < n; ++i) {
    mcgpu_run_opts o = *opts;
    uint64_t first = 0, count = 0;
    mcgpu_shard_packets(opts->n_packets, i, n, &first, &count);
    o.first_packet = opts->first_packet + first;
    o.n_packets = count;
    o.n_replicas = (opts->n_replicas >= 1.0 ? opts->n_replicas : 1.0) * (double)n;
    if ((rc = mcgpu_launch_thermal(mm->ctx[i], &o))) return failed(i, rc);
  }
  // 2) the all-reduce (nothing to do on one device: no communicator is ever created there)
  if ((rc = multi_allreduce(mm, 0))) { multi_drain(mm); return rc; }
  // 3) wait; the packet loop's time is the slowest device's
  double ms_max = 0.0;
  for (int i = 0; i < n; ++i) {
    double ms = 0.0;
    if ((rc = mcgpu_sync(mm->ctx[i], &ms))) return failed(i, rc);
    if (ms > ms_max) ms_max = ms;
  }
  if (kernel_ms) *kernel_ms = ms_max;
  if ((rc = mcgpu_fetch(mm->ctx[0], E_abs, sed, n_sent, counters))) return failed(0, rc);  // every device now holds the global sums
  return MCGPU_OK;
}

// One wavelength of the SED Monte Carlo on every device of the handle: replaces the call at dust_transfer.f90:939 for a
// host with several GPUs.  The opts->n_chunks streams are split into contiguous ranges (they are independent and carry
// their own stopping rule, dust_transfer.f90:525-553), each device runs mcgpu_run_mono on its range -- from a host
// thread of its own, the call's scout / commit passes are synchronous --, then ONE all-reduce sums
// [sed | n_sent | counters] and one xI_scatt.  Needs n_dev <= n_chunks.  Afterwards every device holds the sums
// (mcgpu_rt1_dust_map may run on any of them); sed, n_sent, counters and xI_scatt are read from device 0's context.
extern "C" int mcgpu_multi_run_mono(mcgpu_multi* mm, const mcgpu_mono_opts* opts, double frac_E_stars, double frac_E_disk,
                                    const double* prob_E_cell, uint64_t* n_sent_chunk, double* kernel_ms) {
  if (!mm || !opts) return MCGPU_ERR_ARG;
  const int n = mm->n_dev;
  if (opts->n_chunks < n) { mm->err = "mcgpu_multi_run_mono: more devices than streams"; return MCGPU_ERR_ARG; }
  int rc;
  const int what = multi_what(opts);
  if (opts->accumulate) { if ((rc = multi_prepare_accumulate(mm, what))) { multi_drain(mm); return rc; } }
  else { mm->reduced = false; mm->reduced_what &= ~what; }   // (this call clears what it deposits into; the other kind keeps its state)
  std::vector<int> rcs(n, 0);
  std::vector<double> ms(n, 0.0);
  std::vector<std::thread> th;
  for (int i = 0; i < n; ++i) {
    th.emplace_back([&, i]() {
      mcgpu_mono_opts o = *opts;
      uint64_t first = 0, count = 0;
      mcgpu_shard_packets((uint64_t)opts->n_chunks, i, n, &first, &count);
      o.first_chunk = opts->first_chunk + (int)first;
      o.n_chunks = (int)count;
      rcs[i] = mcgpu_run_mono(mm->ctx[i], &o, frac_E_stars, frac_E_disk, prob_E_cell, n_sent_chunk ? n_sent_chunk + first : nullptr, &ms[i]);
    });
  }
  for (auto& t : th) t.join();
  for (int i = 0; i < n; ++i)
    if (rcs[i]) { mm->err = "device " + std::to_string(i) + ": " + mcgpu_last_error(mm->ctx[i]); multi_drain(mm); return rcs[i]; }
  if ((rc = multi_allreduce(mm, what))) { multi_drain(mm); return rc; }
  double ms_max = 0.0;
  for (int i = 0; i < n; ++i) {
    hipSetDevice(mm->ctx[i]->device);
    if (hipStreamSynchronize(mm->ctx[i]->stream) != hipSuccess) { mm->err = "stream synchronisation failed"; return MCGPU_ERR_HIP; }
    if (ms[i] > ms_max) ms_max = ms[i];
  }
  if (kernel_ms) *kernel_ms = ms_max;
  return MCGPU_OK;
}

// The SED step sharded BY WAVELENGTH (round 6).  run_sed_mc's loop over the wavelengths
// (dust_transfer.f90:899-1027) is itself the natural partition: a wavelength's xI_scatt is produced by that wavelength's
// packets and consumed by that wavelength's ray tracing, then dead.  So device d takes WHOLE wavelengths -- longest first
// by the caller's cost hint, each to the device that is free first -- and runs repartition_energie -> scout / commit ->
// dust_map locally; only the wavelength's SED bins, packet count and ray-traced Stokes values travel (a few KB), and no
// xI_scatt is ever reduced (mcgpu_multi_run_mono all-reduces 200 MB of it per wavelength).  Nothing is summed across
// devices: a wavelength's numbers are those of the single-device call, whichever device ran it.
extern "C" int mcgpu_multi_run_sed(mcgpu_multi* mm, const mcgpu_mono_opts* opts, int n_wl, const mcgpu_sed_wavelength* wl,
                                   const float* Tdust, const mcgpu_rt_opts* rt, const float* tab_RT_az, double* sed,
                                   double* n_sent, double* E_disk, double* stokes_rt, uint64_t* counters, int* device_of,
                                   double* seconds) {
  if (!mm || !opts || n_wl < 1 || !wl || !Tdust) return MCGPU_ERR_ARG;
  const int n = mm->n_dev;
  if (rt && !tab_RT_az) { mm->err = "mcgpu_multi_run_sed: the ray tracing needs tab_RT_az"; return MCGPU_ERR_ARG; }
  if (opts->accumulate) { mm->err = "mcgpu_multi_run_sed: a wavelength is run once, by one device (accumulate = 0)"; return MCGPU_ERR_ARG; }
  // longest first, each to the device with the least work so far (a wavelength of ref4.1 takes 13 to 370 ms)
  std::vector<int> order(n_wl);
  for (int i = 0; i < n_wl; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return wl[a].cost > wl[b].cost; });
  std::vector<std::vector<int>> mine(n);
  std::vector<double> load(n, 0.0);
  for (int i : order) {
    int d = 0;
    for (int k = 1; k < n; ++k) if (load[k] < load[d]) d = k;
    mine[d].push_back(i);
    load[d] += wl[i].cost > 0.0 ? wl[i].cost : 1.0;
    if (device_of) device_of[i] = d;
  }
  mm->reduced = false; mm->reduced_what = 0;   // (every context's accumulators now hold its own last wavelength)
  std::vector<int> rcs(n, 0);
  std::vector<std::thread> th;
  for (int d = 0; d < n; ++d) {
    th.emplace_back([&, d]() {
      mcgpu_ctx* ctx = mm->ctx[d];
      const DevModel& M = ctx->M;
      const size_t n_bins = (size_t)M.N_thet * M.N_phi, nsed = n_sed(M);
      const int nRT = ctx->have_rt1 ? ctx->RT_n_incl * ctx->RT_n_az : 0;
      std::vector<double> sed_all(nsed), ns_all(M.n_lambda);
      for (int i : mine[d]) {
        const auto t0 = std::chrono::steady_clock::now();
        const mcgpu_sed_wavelength& W = wl[i];
        double fs = 0.0, fd = 0.0, Ed = 0.0;
        int rc = mcgpu_repartition_energie(ctx, W.lambda, W.wl_um, W.E_star, W.E_ISM, Tdust, nullptr, &fs, &fd, &Ed, nullptr);
        mcgpu_mono_opts o = *opts;
        o.lambda = W.lambda; o.p_lambda = W.p_lambda > 0 ? W.p_lambda : W.lambda; o.seed = W.seed;
        if (!rc) rc = mcgpu_run_mono(ctx, &o, fs, fd, nullptr, nullptr, nullptr);
        uint64_t cnt[MCGPU_N_COUNTERS];
        if (!rc) rc = mcgpu_fetch(ctx, nullptr, sed_all.data(), ns_all.data(), cnt);
        if (!rc) {
          // the wavelength's own bins of the nine SED arrays: sed(lambda, N_thet, N_phi, type), lambda fastest
          if (sed)
            for (int t = 0; t < MCGPU_N_SED_TYPES; ++t)
              for (size_t b = 0; b < n_bins; ++b)
                sed[((size_t)i * MCGPU_N_SED_TYPES + t) * n_bins + b] = sed_all[(size_t)(W.lambda - 1) + (size_t)M.n_lambda * (b + n_bins * t)];
          if (n_sent) n_sent[i] = ns_all[W.lambda - 1];
          if (E_disk) E_disk[i] = Ed;
          if (counters) for (int q = 0; q < MCGPU_N_COUNTERS; ++q) counters[(size_t)i * MCGPU_N_COUNTERS + q] = cnt[q];
        }
        if (!rc && rt && stokes_rt && o.rt1 == 1) {
          mcgpu_rt_opts r = *rt;
          r.lambda = W.lambda; r.wl_um = W.wl_um; r.E_src = W.E_star + Ed + W.E_ISM; r.n_sent_photons = ns_all[W.lambda - 1];
          rc = mcgpu_rt1_dust_map(ctx, &r, tab_RT_az, Tdust, stokes_rt + (size_t)i * nRT * ctx->N_type_flux, nullptr);
        }
        if (seconds) seconds[i] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc) { rcs[d] = rc; return; }
      }
    });
  }
  for (auto& t : th) t.join();
  for (int d = 0; d < n; ++d)
    if (rcs[d]) { mm->err = "device " + std::to_string(d) + ": " + mcgpu_last_error(mm->ctx[d]); multi_drain(mm); return rcs[d]; }
  return MCGPU_OK;
}
