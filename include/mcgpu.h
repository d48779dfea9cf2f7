/*
 * mcgpu.h -- C-ABI of the MI355X Monte Carlo packet engine (libmcfost_hip.so).
 *
 * Drop-in boundary for ONE path of MCFOST (cpinte/mcfost 4.1.13): the body of
 *     subroutine mc_photon_loop            src/dust_transfer.f90:439-572
 * as called by run_thermal_mc              src/dust_transfer.f90:576-650
 * plus the reduction that follows it,
 *     subroutine Temp_finale               src/thermal_emission.f90:870-906.
 *
 * In the reference every input/output of that loop is a module-level array
 * owned by the Fortran host (shared list, dust_transfer.f90:486-488).  The
 * setters below receive exactly those arrays -- column-major, 1-based cell
 * ids, caller-owned, copied to HBM once -- and mcgpu_run_thermal() replaces
 * the OpenMP region.  The Fortran side of the shim is
 * mcfost_amd/fortran/mcgpu_f.f90 (ISO_C_BINDING); INTEGRATION.md shows the
 * patch to dust_transfer.f90.
 *
 * Conventions (following the reference's own FFI precedents,
 * Voronoi.f90:70-96 `voro_C` and mcfost2phantom.f90:159-162):
 *   - scalars by value, arrays as plain pointers, `int` return = ierr
 *     (0 = ok); the library never calls exit();
 *   - no hidden state besides the context handle;
 *   - packet counts are 64-bit (config 3 is 1e9 packets; the reference counts
 *     in real(dp), dust_transfer.f90:462-464);
 *   - thread-safe for a single caller per context.
 *
 * There is NO CPU fallback: every entry point fails with MCGPU_ERR_NO_DEVICE
 * when no HIP device is usable.
 */
#ifndef MCGPU_H
#define MCGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcgpu_ctx mcgpu_ctx;

enum {
  MCGPU_OK = 0,
  MCGPU_ERR_NO_DEVICE = 1,   /* no usable HIP device                         */
  MCGPU_ERR_HIP = 2,         /* a HIP runtime call failed (see last_error)   */
  MCGPU_ERR_ARG = 3,         /* bad argument                                 */
  MCGPU_ERR_UNSUPPORTED = 4, /* valid MCFOST input this engine does not take */
  MCGPU_ERR_STATE = 5,       /* a required mcgpu_set_* call is missing       */
  MCGPU_ERR_KERNEL = 6       /* the kernel flagged an internal error         */
};
/* MCGPU_ERR_KERNEL: mcgpu_last_error names the kernel's code -- 12: an emission source outside the engine's scope;
 * 13: a packet of more than 2e8 crossings was dropped (a packet that never leaves the grid); 15: a scheduling watchdog
 * ended the launch instead of letting it hang (every wait of the role / pool schedules is bounded: a logic error, not an
 * input error); 16, 17: a hand-over buffer (role kernel -> k_tail, k_tail -> host) overflowed; 18: a launch of the SED
 * commit pass logged more deposits than its log holds (option "xi_log" = 0 runs the model with atomics). */

#define MCGPU_N_SED_TYPES 9 /* sed, sed_q, sed_u, sed_v, n_phot_sed, sed_star,
                               sed_star_scat, sed_disk, sed_disk_scat
                               (output.f90:572-592, allocate_sed :103)       */
#define MCGPU_N_COUNTERS 10
enum {
  MCGPU_CNT_PACKETS = 0,   /* packets launched (sum of n_phot_envoyes)       */
  MCGPU_CNT_CROSSINGS = 1, /* cross_cell evaluations                         */
  MCGPU_CNT_FLIGHTS = 2,   /* physical_length calls                          */
  MCGPU_CNT_SCATT = 3,
  MCGPU_CNT_ABS = 4,
  MCGPU_CNT_ESCAPED = 5,
  MCGPU_CNT_KILLED_STAR = 6,
  MCGPU_CNT_DARK = 7,
  MCGPU_CNT_MRW_WALKS = 8, /* modified random walks (mcgpu_set_mrw)            */
  MCGPU_CNT_MRW_STEPS = 9  /* sphere steps of those walks                      */
};

/* Device context on HIP device `device` (one per process/rank). */
int mcgpu_create(int device, mcgpu_ctx **ctx);
int mcgpu_destroy(mcgpu_ctx *ctx);
const char *mcgpu_last_error(const mcgpu_ctx *ctx);

/*
 * Cylindrical grid: module cylindrical_grid (cylindrical_grid.f90:20-35).
 *   r_lim_2[0:n_rad], zmax[n_rad], z_lim(n_rad,nz+2), tan_phi_lim[n_az],
 *   volume[n_cells], cell_map(0:n_rad+1, jstart2:nz+1, n_az),
 *   cell_map_i/j/k[ntot2], lexit_cell[ntot2]  -- as built by
 *   build_cylindrical_cell_mapping (:45) and define_cylindrical_grid (:183).
 * zmaxmax is private in the reference module (:20): pass maxval(zmax).
 * The engine keeps (ri,zj,phik) in registers and evaluates the mapping in
 * closed form; the arrays are used to VERIFY that closed form (returns
 * MCGPU_ERR_UNSUPPORTED for a mapping or a non-uniform z grid it cannot
 * reproduce, e.g. lidefix grids :467-477).
 */
int mcgpu_set_grid_cyl(mcgpu_ctx *ctx, int n_rad, int nz, int n_az, int l3D,
                       const double *r_lim_2, const double *zmax,
                       const double *z_lim, const double *tan_phi_lim,
                       double zmaxmax, double Rmax2, const double *volume,
                       const int *cell_map, const int *cell_map_i,
                       const int *cell_map_j, const int *cell_map_k,
                       const int *lexit_cell);

/*
 * Interstellar radiation field: packets whose emission draw exceeds frac_E_disk(lambda) start
 * on the sphere (centre_ISM, R_ISM) with a cosine law towards the interior (emit_packet_ISM,
 * stars.f90:27-28, 655-666, 728-785) and are binned only once the dust has absorbed and
 * re-emitted them (flag_ISM, dust_transfer.f90:549).  R_ISM = 0 (default): no such source, a
 * draw beyond frac_E_disk is an error.  The weight of the field enters through the host's
 * spectre_emission_cumul / frac_E_disk (E_ISM, thermal_emission.f90:329-341, 1916-1917).
 */
int mcgpu_set_ism(mcgpu_ctx *ctx, double R_ISM, const double *centre_ISM);

/*
 * Voronoi grid: module Voronoi_grid (Voronoi.f90:23-67), the arrays
 * Voronoi_tesselation (:183-640) leaves behind.  All ids are 1-based.
 *   voronoi_xyz(3,n_cells)   Voronoi_xyz, default real (:62)
 *   xyz_dp(3,n_cells)        Voronoi(:)%xyz (:24)
 *   h[n_cells]               Voronoi(:)%h (:25)
 *   first/last_neighbour     Voronoi(:)%first_neighbour, %last_neighbour (:26)
 *   neighbours_list[n_neighbours]  cell id > 0, or -iwall (:65, :560-600)
 *   was_cut, is_star_neighbour     Voronoi(:)%was_cut (:27), %is_star_neighbour (:28);
 *                                  NULL = all false
 *   walls[6*4]               wall(i)%x1..x4 in the order of init_Voronoi_walls
 *                            (:1262-1283): -x,+x,-y,+y,-z,+z
 *   cutting_distance_o_h     PS%cutting_distance_o_h (:50)
 *   wall_first[7], wall_cells wall(i)%neighbour_list(1:n_neighbours) (:36-37)
 *                            concatenated, wall_first[i] = 0-based start of wall i+1
 *   volume[n_cells]          volume (cylindrical_grid.f90:26; filled at Voronoi.f90:529)
 * Star sites are ordinary cells here (:361-376); pass their ids as `icell` to
 * mcgpu_set_stars.  The grid operators replaced: cross_Voronoi_cell (:839),
 * test_exit_grid_Voronoi (:1446), move_to_grid_Voronoi (:1379),
 * index_cell_voronoi (:1548), pos_em_cell_voronoi (:1510).
 */
int mcgpu_set_grid_voronoi(mcgpu_ctx *ctx, int n_cells, const float *voronoi_xyz,
                           const double *xyz_dp, const double *h,
                           const int *first_neighbour, const int *last_neighbour,
                           const int *neighbours_list, long long n_neighbours,
                           const unsigned char *was_cut,
                           const unsigned char *is_star_neighbour,
                           const float *walls, double cutting_distance_o_h,
                           const int *wall_first, const int *wall_cells,
                           const double *volume);

/*
 * Spherical grid (grid_type = 2: the operators of spherical_grid.f90 bound at grid.f90:345-357).  Arrays of module
 * cylindrical_grid (cylindrical_grid.f90:28-31, filled by the spherical branch of define_cylindrical_grid,
 * :496-580): r_lim_2(0:n_rad), r_lim_3(0:n_rad), tan_theta_lim(0:nz), theta_lim(0:nz), tan_phi_lim(n_az); the cell
 * mapping arrays are the ones build_cylindrical_cell_mapping fills for both structured grids (verified like
 * mcgpu_set_grid_cyl's).  Replaces cross_spherical_cell (:182), index_cell_sph (:48), move_to_grid_sph (:562),
 * pos_em_cell_sph (:619), test_exit_grid_sph (:24).  On this grid: the temperature step (with the random walk; or, round 5,
 * with a dark zone -- l_dark_zone flags of mcgpu_set_opacity -- or with dust classes, mcgpu_set_variable_dust /
 * mcgpu_opacity; not both at once, not with the walk), the SED / image Monte Carlo with the deposits of ray tracing method
 * 1 or (2D) method 2, both ray tracers, the optical-depth maps -- with dust classes too.  MCGPU_ERR_UNSUPPORTED, because the
 * reference has no such thing on this grid (`if (lspherical.or.l3D) call no_dark_zone()`, dust_transfer.f90:290-293,
 * 734-735, 916-917): mcgpu_define_dark_zone, the diffusion fill, a dark zone in SED mode or in the ray tracer.
 */
int mcgpu_set_grid_sph(mcgpu_ctx *ctx, int n_rad, int nz, int n_az, int l3D,
                       const double *r_lim_2, const double *r_lim_3,
                       const double *tan_theta_lim, const double *theta_lim,
                       const double *tan_phi_lim, double Rmax2, const double *volume,
                       const int *cell_map, const int *cell_map_i, const int *cell_map_j,
                       const int *cell_map_k, const int *lexit_cell);

/*
 * 3D grids only.  on == 0 (default): the reference's literal arithmetic (a build without FMA
 * contraction) -- a packet that crosses the midplane lands at z0 + t*w, whose SIGN (the last ulp of
 * the product, compiler and libm dependent) decides on which side of the midplane the packet is for
 * the next cell.  on != 0: the packet lands at z = sign(grid_prec, w), i.e. the reference's own
 * z1 == 0 correction (cylindrical_grid.f90:1158-1165) applied to every rounding residue; an
 * axisymmetric 3D run then reproduces the 2D run packet for packet, and two builds of the algorithm
 * (this engine and its CPU oracle) can be compared packet for packet -- the parity tests switch it on.
 */
int mcgpu_set_midplane_snap(mcgpu_ctx *ctx, int on);

/*
 * Per-context run options (the library reads no environment variable; the switches of the
 * reference that select code paths are its command-line flags, init_mcfost.f90):
 *   "deposit"      0 = automatic (default): the absorbed-energy grid of a 2D model is kept
 *                      per workgroup in LDS; 3D cylindrical grids write binned deposit records
 *                      to a log in HBM that is folded into the grid between chunks of the run
 *                      (mc_binned.hip.h; the launch stays asynchronous); Voronoi grids deposit
 *                      through a hashed LDS cache in front of HBM atomics; 1 = HBM atomics only;
 *                      2 = LDS (refused when the grid does not fit); 3 = binned (refused where
 *                      it is not built)
 *   "tail"         the role kernels hand their last packets to the tail kernel (one packet per wave with the other
 *                      lanes working ahead for it, mc_tail.hip.h) once a workgroup has this many left; 0 = never;
 *                      -1 (default) = automatic: 48 where packets get trapped (the context's last launch had at
 *                      least one interaction per packet, or -- first launch -- the midplane is optically thick)
 *   "tail_where"   where a launch's LAST packets end: 0 (default) = automatic, 2 = k_tail thins the tail out (thousands of
 *                      packets at once) and, once no more than "tail_host_packets" are unfinished, hands them to the
 *                      library's host threads (host_tail.cpp: the device source compiled for the CPU, one packet per
 *                      thread -- a packet is one dependent chain of events, which a wave runs at 1.0-1.6 us per event
 *                      and a host core at 30-100 ns; never the CPU oracle of the tests); 1 = k_tail finishes every
 *                      packet.  The hand-over is stream-ordered (copies + a host callback on the context's stream):
 *                      a launch stays asynchronous.  Automatic = 2 wherever k_tail runs.  Which packets the host
 *                      finishes depends on the schedule, and its libm rounds log / sin / cos in other last digits
 *                      than the device's: with 2, two runs of the same seed are two realisations of those few hundred
 *                      packets (equal to Monte Carlo noise, like two runs of the reference on different thread
 *                      counts); 1 gives the same sums every time (to the order of the atomic additions).
 *   "host_threads" host threads of such a tail: 0 (default) = the machine's hardware threads divided by its GPUs,
 *                      at most 32; 1..256
 *   "tail_host_packets"  packets k_tail leaves to the host: 0 (default) = 8 per host thread; 1..65536
 *   "xi_log"       SED mode, default-real xI_scatt (mcgpu_set_xI_precision(4)), one dust class, cylindrical grid: 1 (default) =
 *                      the commit pass LOGS its deposits (12 bytes per crossing + the flight's weights once) and a sort
 *                      by sub-bin + segmented sums replace the atomics, at the wavelengths where flights are long
 *                      enough for that to pay (>= 20 crossings per flight, measured by the pass's first launch) and a crossing's
 *                      atomics would touch at least four 64-byte lines (many observers);
 *                      0 = atomics always; 2 = the log always.  Same sums to the order of default-real additions.
 *   "deposit_log_mb"  size of the binned-deposit log in MiB; 0 (default) = up to 64 GiB (what the packets asked for need), at most a quarter of
 *                      the free device memory.  A smaller log means more, shorter chunks; a
 *                      block that finds its part of the log full is added with atomics.
 *   "schedule"     0 = automatic (default): waves with roles and LDS packet queues where
 *                      the queues fit (cylindrical grids; Voronoi grids run the single-role
 *                      kernel, which is faster on them); 1 = the single-role kernel;
 *                      2 = the role schedule wherever it is built (also Voronoi grids);
 *                      3 = Voronoi grids: the pool schedule (packet records in HBM, queues by phase and
 *                      by neighbour-list length in LDS, one phase per wave pass; mc_voronoi_pool.hip.h)
 *   "crossing"     0 (default) = the reference's crossing arithmetic everywhere (cross_cylindrical_cell,
 *                      cylindrical_grid.f90:918-1175: golden walks bit for bit, packets equal to the CPU
 *                      restatement's one for one); 1 = 2D grids without dark zone / random walk / dust
 *                      classes: the flying waves take the wall distances as functions of the path parameter
 *                      along the flight (no position update per crossing; Pascucci +22 %).  NOT the
 *                      reference's arithmetic: the same cells but for ties at the rounding level, parity with
 *                      the reference statistical only (tests/test_param_crossing.py).  Never chosen by itself.
 *   "speculation"  SED mode: 1 (default) = most of every stream is committed before the
 *                      scout pass (exact; see mcgpu_run_mono), 0 = scout every packet first
 *   "voronoi_cache_log_slots"  6..13 (default 13): log2 of the slots of the Voronoi deposit cache
 *   "voronoi_pool_log_records" 6..12 (default 12): log2 of the packet records per workgroup of schedule 3
 *   "radiation_field"  bit 0: keep xN_abs, bit 1: keep xJ_abs in the thermal step (mcgpu_fetch_radiation_field);
 *                      cylindrical grids then run the single-role kernel
 * Results do not depend on any of them (same packets, same random numbers) -- except "crossing" = 1, which changes
 * the crossing's arithmetic (statistical parity only), and to the rounding of a sum's order ("tail_where": a host
 * thread evaluates log / sin / cos with the host's libm, the device with its own: a packet's history is the same
 * function of the same random numbers, and differs where a last digit decides a branch).
 */
int mcgpu_set_option(mcgpu_ctx *ctx, const char *name, int value);
/* Diagnostics of the last launches: "bin_buckets", "bin_log_blocks", "bin_chunks",
 * "bin_deposits_per_packet", "bin_overflow_blocks", "bin_drained_records", "tail_threshold", "tau_midplane",
 * "tail_ms", "longest_packet_events" / _crossings / _scatterings / _absorptions / _walks / _steps, and of the last
 * launch's tail: "tail_where" (0: it had none, 1: k_tail finished it, 2: the host threads did), "tail_host_ms",
 * "tail_host_packets", "tail_host_threads", "tail_host_events"; of the last mcgpu_run_mono's commit passes:
 * "xi_log_chunks" (launches that logged their deposits; 0: atomics), "xi_log_records", "xi_log_flights"; of the packed
 * default-real xI_scatt layout this context's observers get (mcgpu_set_xI_precision): "xi_bin_floats" (default reals per
 * sub-bin), "xi_lines_per_crossing" (64-byte lines one crossing's deposits touch), "xi_split" (1: the split arrangement). */
int mcgpu_get_info(mcgpu_ctx *ctx, const char *name, double *value);

/* Stars: type star_type (parameters.f90:230-242); icell/out_model from
 * stars_cell_indices (stars.f90:789-808). Lengths in AU. */
int mcgpu_set_stars(mcgpu_ctx *ctx, int n_stars, const double *x,
                    const double *y, const double *z, const double *r,
                    const int *icell, const int *out_model);

/* Opacities for one cell class (lvariable_dust = .false., p_n_cells = 1):
 * kappa(1,:), kappa_abs_LTE(1,:) (dust_prop.f90:17-19), tab_albedo_pos(1,:)
 * (grains.f90:62), kappa_factor(n_cells) (dust_prop.f90:17),
 * l_dark_zone(n_cells) as bytes or NULL (cylindrical_grid.f90:38). */
int mcgpu_set_opacity(mcgpu_ctx *ctx, int n_lambda, const double *kappa,
                      const double *kappa_abs_LTE, const float *tab_albedo_pos,
                      const double *kappa_factor,
                      const unsigned char *l_dark_zone);

/* Scattering tables of scattering method 2 (per cell class), grains.f90:62-64:
 * prob_s11_pos(0:nang,1,n_lambda), tab_sXX_o_s11_pos(0:nang,1,n_lambda),
 * tab_g_pos(1,n_lambda).  aniso_method 1 = tabulated, 2 = HG.
 * p_lambda_fixed != 0 reproduces the reference: during the thermal step the
 * angle CDF is sampled at wavelength index 1 (dust_transfer.f90:491-502). */
int mcgpu_set_scattering(mcgpu_ctx *ctx, int nang_scatt, int aniso_method,
                         int lisotropic, int lsepar_pola, int p_lambda_fixed,
                         const float *prob_s11_pos, const float *s12_o_s11,
                         const float *s22_o_s11, const float *s33_o_s11,
                         const float *s34_o_s11, const float *s44_o_s11,
                         const float *tab_g_pos);

/* Thermal tables (thermal_emission.f90:34-60; several are `private` there, so
 * the Fortran shim lives next to that module):
 *   tab_Temp[n_T] (Temperature.f90:11), log_Qcool_minus_extra_heating(n_T,1),
 *   kdB_dT_CDF(n_lambda,n_T,1), spectre_emission_cumul(0:n_lambda),
 *   frac_E_stars[n_lambda], frac_E_disk[n_lambda],
 *   CDF_E_star(n_lambda,0:n_stars) (stars.f90:575-604),
 *   prob_E_cell(0:n_cells,n_lambda) or NULL when frac_E_stars == 1,
 *   L_packet_th (:355-356), T_min (parameters.f90:269).
 * log_Qcool and kdB_dT_CDF may both be NULL: mcgpu_init_reemission then builds them on the device (a launch before
 * that call fails with MCGPU_ERR_STATE). */
int mcgpu_set_thermal(mcgpu_ctx *ctx, int n_T, const float *tab_Temp,
                      const double *log_Qcool, const double *kdB_dT_CDF,
                      const double *spectre_emission_cumul,
                      const double *frac_E_stars, const double *frac_E_disk,
                      const double *CDF_E_star, const double *prob_E_cell,
                      double L_packet_th, float T_min);

/* SED bins used by capteur (output.f90:294): N_thet, N_phi and the symmetry
 * flags (parameters.f90:103-107). */
int mcgpu_set_sed_bins(mcgpu_ctx *ctx, int N_thet, int N_phi,
                       int l_sym_centrale, int l_sym_axiale);

typedef struct {
  uint64_t seed;         /* Philox key (reference: SPRNG seed 269753)        */
  uint64_t first_packet; /* global id of this call's first packet: ranks use
                            disjoint ranges                                  */
  uint64_t n_packets;
  double n_replicas;     /* ranks sharing the job: the in-flight temperature
                            uses local E_abs * n_replicas, the analogue of
                            `* nb_proc` in thermal_emission.f90:670          */
  int frozen;            /* 0 = live Bjorkman & Wood feedback (reference);
                            1 = Temp_LTE reads the prior set by
                                mcgpu_set_E_prior (reproducible mode)        */
  int accumulate;        /* 0 = zero the accumulators first (what
                            reset_radiation_field does, dust_transfer.f90:594)
                            1 = keep accumulating                            */
  int grid_blocks;       /* 0 = auto                                         */
  int block_threads;     /* 0 = auto                                         */
} mcgpu_run_opts;

/* Prior absorbed-energy grid for frozen mode (host array, n_cells). */
int mcgpu_set_E_prior(mcgpu_ctx *ctx, const double *E_prior);

/*
 * mc_photon_loop, thermal step (letape_th = .true., lmono = .false.).
 * Synchronous convenience form for the Fortran host.  Outputs (host, caller
 * owned, overwritten with this context's totals):
 *   E_abs[n_cells]                          -> xKJ_abs(:,1)
 *   sed[9][N_phi][N_thet][n_lambda]         -> sed, sed_q, ... (:,:,:,1)
 *   n_sent[n_lambda]                        -> n_phot_envoyes(:,1)
 *   counters[MCGPU_N_COUNTERS], kernel_ms   (any may be NULL)
 * One deliberate difference in arithmetic: a flight's optical depth tau = -log(1 - rand) is formed from the same
 * default-real `rand` as the reference's but with an FP64 logarithm, where the reference's `tau` is default real
 * (dust_transfer.f90:1182,1208-1215): the same distribution to 6e-8 (the CPU oracle has both forms;
 * tests/test_oracle_kats.py::test_fp32_and_fp64_tau_are_statistically_equivalent).
 */
int mcgpu_run_thermal(mcgpu_ctx *ctx, const mcgpu_run_opts *opts,
                      double *E_abs, double *sed, double *n_sent,
                      uint64_t *counters, double *kernel_ms);

/* Asynchronous pieces of the same call, for hosts that keep the accumulators
 * on the device between the launch and the RCCL all-reduce. */
int mcgpu_launch_thermal(mcgpu_ctx *ctx, const mcgpu_run_opts *opts);
int mcgpu_sync(mcgpu_ctx *ctx, double *kernel_ms);
/* Device pointers of the fused accumulator [E_abs | sed | n_sent | counters] (doubles; the last
 * MCGPU_N_COUNTERS entries are the tail filled by mcgpu_counters_to_accum) and of the counters
 * themselves (uint64[MCGPU_N_COUNTERS]). */
int mcgpu_device_accumulators(mcgpu_ctx *ctx, void **accum_dev,
                              uint64_t *n_doubles, void **counters_dev);
int mcgpu_fetch(mcgpu_ctx *ctx, double *E_abs, double *sed, double *n_sent,
                uint64_t *counters);

/* Multi-GPU hosts reduce ONE buffer per temperature iteration: mcgpu_counters_to_accum copies the
 * event counters into the tail of the fused accumulator as doubles (exact below 2^53) before the
 * all-reduce, mcgpu_counters_from_accum takes the summed values back afterwards.  Both are
 * asynchronous on the context's stream. */
int mcgpu_counters_to_accum(mcgpu_ctx *ctx);
int mcgpu_counters_from_accum(mcgpu_ctx *ctx);
/* Use an external stream (e.g. torch's current stream); NULL = own stream. */
int mcgpu_set_stream(mcgpu_ctx *ctx, void *hip_stream);

/* ------------------------------------------------------------------------
 * SED mode: one wavelength of run_sed_mc (dust_transfer.f90:828-1042), i.e.
 * mc_photon_loop (:439-572) with lmono = .true., lmono0 = .false.:
 * monochromatic packets, forced scattering (:1263-1278), ray-tracing method 1
 * deposits xI_scatt (radiation_field.f90:63-89, dust_ray_tracing.f90:409-632)
 * and capteur into the SED arrays.  Cylindrical grids.
 * ------------------------------------------------------------------------ */

/*
 * Ray-tracing method 1 tables (module dust_ray_tracing):
 *   tab_u_rt, tab_v_rt (RT_n_incl, RT_n_az), tab_w_rt[RT_n_incl]   (:18-19, :234-300)
 *   n_az_rt, n_theta_rt   layout of xI_scatt: 45, 2 in 2D; 1, 1 in 3D (:91-98)
 *   N_type_flux           8 / 4 / 5 / 1 (init_mcfost.f90:1603-1616)
 *   tab_s11_pos(0:nang_scatt, 1, n_lambda_pos)                      (grains.f90:63)
 * Call after mcgpu_set_scattering.
 */
int mcgpu_set_rt1(mcgpu_ctx *ctx, int RT_n_incl, int RT_n_az,
                  const double *tab_u_rt, const double *tab_v_rt,
                  const double *tab_w_rt, int n_az_rt, int n_theta_rt,
                  int N_type_flux, int lsepar_contrib, const float *tab_s11_pos,
                  int n_lambda_pos);

typedef struct {
  uint64_t seed;
  int lambda;           /* 1-based wavelength index                                */
  int p_lambda;         /* index into the phase-function tables (run_sed_mc :899-909) */
  int n_chunks;         /* n_photons_loop: independent sequential streams (:525)   */
  int first_chunk;      /* global id of this call's first stream (0 on one GPU; rank r
                           of a multi-GPU run takes its own contiguous range)       */
  uint64_t n_photons2;  /* n_photons_lambda: packets each stream must see binned in
                           inclination bin capt_sup before it stops (:526, :551)   */
  double n_phot_lim;    /* n_photons_lim: cap on the packets a stream sends (:526) */
  int capt_sup;         /* read_param.f90:184                                      */
  int rt1;              /* 1: lscatt_ray_tracing1, accumulate xI_scatt; 2: lscatt_ray_tracing2,
                           accumulate I_spec / I_spec_star (mcgpu_set_rt2); 0: neither */
  int accumulate;       /* 0: zero sed / n_sent / counters / xI_scatt / I_spec first */
  int grid_blocks, block_threads; /* 0 = automatic                                  */
} mcgpu_mono_opts;

/*
 * Ray tracing method 2 (lscatt_ray_tracing2; the reference's default for 2D images, init_mcfost.f90:1854-1863): the
 * packet loop stores the specific intensity per cell and direction bin instead of the field scattered towards each
 * observer (save_radiation_field, radiation_field.f90:91-129; 2D grids only):
 *   I_spec(1:N_type_flux, theta_I, phi_I, icell) += l * Stokes   phi_I: azimuth of the flight direction relative to the
 *                                                                azimuth of the path's midpoint, theta_I: cos(theta),
 *                                                                mirrored below the midplane (n_phi_I x n_theta_I bins,
 *                                                                15 x 15 in dust_ray_tracing.f90:102-105)
 *   I_spec_star(icell)                           += l * Stokes(1) for starlight that has not interacted yet
 * One 64-byte record per crossing, whatever the number of observers.  mcgpu_set_rt2 allocates (and zeroes) the two
 * arrays in HBM; mcgpu_run_mono with opts->rt1 = 2 deposits; mcgpu_fetch_I_spec returns them in the reference's layout
 * (N_type_flux, n_theta_I, n_phi_I, n_cells) as default real (the type of the reference's arrays) and / or as the FP64
 * sums the device holds; any pointer may be NULL.  N_type_flux = n_Stokes (+ 4 with lsepar_contrib).
 * mcgpu_set_I_spec hands the arrays back (e.g. after a sum over processes; reference layout, double).
 *
 * mcgpu_rt2_source is init_dust_source_fct2(lambda, p_lambda, ibin) (dust_ray_tracing.f90:717-806) on the device -- with
 * calc_Isca_rt2_star (:1245-1440; angles_scatt_rt2 :304-405), calc_Isca_rt2 (:907-1240) and calc_Jth, the slow steps of
 * the reference's method-2 ray tracer -- from the I_spec / I_spec_star the device holds: the source function of the
 * inclination ibin (1-based) in the reference's arrays and type,
 *   eps_dust2(N_type_flux, nang_ray_tracing, 0:1, n_cells)        ( (I_sca2 + J_th) / kappa_ext; (Q, U) as (P, angle) )
 *   eps_dust2_star(n_Stokes, nang_ray_tracing_star, 0:1, n_cells) ( once-scattered starlight )
 * (default real; nang_ray_tracing = 15, nang_ray_tracing_star = 1000 in dust_ray_tracing.f90:109-110).  opts: lambda,
 * wl_um, E_src, n_sent_photons; Tdust[n_cells]; r_grid / z_grid[n_cells] (cylindrical_grid.f90:26).  Needs
 * mcgpu_set_rt1 (the observers' inclinations, tab_s11_pos) and mcgpu_set_rt2.  The ray integration with this source
 * function: mcgpu_rt2_dust_map / mcgpu_rt2_image below.
 */
int mcgpu_set_rt2(mcgpu_ctx *ctx, int n_theta_I, int n_phi_I, int N_type_flux, int lsepar_contrib);
int mcgpu_fetch_I_spec(mcgpu_ctx *ctx, float *I_spec, double *I_spec_f64, float *I_spec_star, double *I_spec_star_f64);
int mcgpu_set_I_spec(mcgpu_ctx *ctx, const double *I_spec, const double *I_spec_star);
/* (mcgpu_rt2_source is declared below, behind mcgpu_rt_opts) */

/*
 * repartition_energie(lambda) (thermal_emission.f90:1771-1949; the call at dust_transfer.f90:924), LTE grains
 * (lRE_LTE, :1814-1831, with kappa_abs_LTE per class under mcgpu_set_variable_dust and l_dark_zone as set with the
 * opacity): how the energy emitted at one wavelength splits between the stars, the cells of the disk and the
 * interstellar field, from the dust temperature of the thermal step -- on the device (one thread per cell; the
 * cumulative distribution is summed in the reference's own cell order, so it is monotone and ends in exactly 1).
 *   lambda, wl_um           the wavelength index (1-based) and tab_lambda(lambda) in micron
 *   E_star, E_ISM           E_stars(lambda), E_ISM(lambda)
 *   Tdust[n_cells]          default real (host); weight_proba_emission[n_cells] or NULL (lweight_emission).  The weights
 *                           must all be 1 (as the reference's own are: its generator of weight_proba_emission and
 *                           correct_E_emission is commented out, thermal_emission.f90:2078-2135, 2147-2148); any other
 *                           value is refused with MCGPU_ERR_UNSUPPORTED -- the engine does not apply the compensating
 *                           packet weight Stokes(1) *= correct_E_emission(icell) (dust_transfer.f90:1140-1142), and a
 *                           biased table without it would bias the SED silently.
 * Out (any may be NULL): frac_E_stars(lambda), frac_E_disk(lambda), E_disk(lambda), prob_E_cell(0:n_cells, lambda).
 * The cumulative distribution also STAYS on the device: a following mcgpu_run_mono of the same wavelength may pass
 * prob_E_cell = NULL.  Fails (MCGPU_ERR_ARG) where the reference stops: no energy at all at this wavelength (:1899).
 * The per-grain branches (lRE_nLTE, lnRE: :1833-1884) are not built -- the engine holds no per-grain temperatures.
 */
int mcgpu_repartition_energie(mcgpu_ctx *ctx, int lambda, double wl_um, double E_star, double E_ISM,
                              const float *Tdust, const float *weight_proba_emission,
                              double *frac_E_stars, double *frac_E_disk, double *E_disk,
                              double *prob_E_cell);

/*
 * Replaces `call mc_photon_loop(lambda, p_lambda, n_photons2, n_phot_lim, 1, .false.)`
 * at dust_transfer.f90:939.  frac_E_stars / frac_E_disk / prob_E_cell(0:n_cells) are the
 * wavelength's entries as left by repartition_energie(lambda) (:924,
 * thermal_emission.f90:1771-1949; mcgpu_repartition_energie above builds them on the device); prob_E_cell may be
 * NULL when frac_E_stars = 1 or when mcgpu_repartition_energie of this wavelength was the last to fill the table.
 * A stream's packets are id ((first_chunk + stream) << 40 | sequence) of the random generator; each stream
 * stops EXACTLY where the reference's sequential loop would: a first pass without deposits
 * finds the stopping index, a second pass replays the packets before it with deposits.
 * n_sent_chunk[n_chunks] (may be NULL) returns the packets each stream sent; their sum is
 * what the call added to n_sent(lambda) = n_phot_envoyes(lambda,:).  sed, n_sent and the
 * counters are read back with mcgpu_fetch, xI_scatt with mcgpu_fetch_xI.
 * Grids: cylindrical, spherical (no dark zone: the reference defines none there) and Voronoi (rt1 = 0 or 1).  Scattering method 1 is
 * refused (the reference forces method 2 with ray tracing, init_mcfost.f90:1659).
 */
int mcgpu_run_mono(mcgpu_ctx *ctx, const mcgpu_mono_opts *opts,
                   double frac_E_stars, double frac_E_disk,
                   const double *prob_E_cell, uint64_t *n_sent_chunk,
                   double *kernel_ms);

/* xI_scatt(n_az_rt, n_theta_rt, N_type_flux, RT_n_incl*RT_n_az, n_cells) summed over what the
 * reference keeps per thread (dust_ray_tracing.f90:33,152): default real like the reference's
 * array and/or the FP64 sums the engine accumulates.  Either pointer may be NULL. */
int mcgpu_fetch_xI(mcgpu_ctx *ctx, float *xI_scatt_f32, double *xI_scatt_f64);

/* Accumulator type of xI_scatt on the device: 8 = FP64 sums (default: device = oracle to 1e-6), 4 = default real,
 * the type of the reference's own array (dust_ray_tracing.f90:33).  With 4 a sub-bin's observers lie side by side with
 * only the values a deposit can reach (the Stokes values; with lsepar_contrib Q, U, V and the two origins of scattered
 * light -- I is not stored there: calc_xI_scatt[_pola] adds the same flux to I and to exactly one origin,
 * dust_ray_tracing.f90:515-524, 616-627, so I is read back as their sum): 3 lines of 64 bytes per crossing at 10
 * observers instead of 10 -- the memory-side line operations a run with many observers is bound by; a sum of N
 * deposits then carries a rounding error ~ sqrt(N) * 6e-8, far below its Monte Carlo noise 1/sqrt(N).  (mcgpu_set_xI on
 * such a context keeps the flux types a deposit can reach: with lsepar_contrib the I it is handed is ignored -- it must
 * be the sum of the origins n_Stokes + 2 and + 4, as the Monte Carlo makes it -- and n_Stokes + 1 and + 3, direct
 * light, which no Monte Carlo deposit ever touches, have no place there and read back as 0.)  Call before the first mcgpu_run_mono / mcgpu_set_xI; changing it drops what was
 * accumulated.  mcgpu_fetch_xI, mcgpu_set_xI and the ray tracer work with either; mcgpu_device_xI exposes
 * accumulators of this type. */
int mcgpu_set_xI_precision(mcgpu_ctx *ctx, int bytes_per_value);
int mcgpu_get_xI_precision(mcgpu_ctx *ctx);

/* Replace the device accumulator by the host's xI_scatt (same layout as mcgpu_fetch_xI's FP64 output):
 * what a host that reduced xI_scatt itself (the reference's thread sum :152, an MPI reduction) hands
 * back before mcgpu_rt1_dust_map.  Needs mcgpu_set_rt1. */
int mcgpu_set_xI(mcgpu_ctx *ctx, const double *xI_scatt_f64);

/* The device-resident xI_scatt accumulator (engine layout, n_values of the type set above) for an in-place RCCL
 * all-reduce across the ranks of a multi-GPU SED step; fetch afterwards with mcgpu_fetch_xI. */
int mcgpu_device_xI(mcgpu_ctx *ctx, void **xI_dev, uint64_t *n_values);

/* ------------------------------------------------------------------------
 * Ray-traced SED of the dust, ray-tracing method 1: what
 * dust_map(lambda, ibin, iaz) (dust_transfer.f90:1413-1600) adds to
 * Stokes_ray_tracing(lambda,1,1,ibin,iaz,:) when RT_sed_method == 1 --
 * init_dust_source_fct1 (dust_ray_tracing.f90:636-716) + calc_Jth (:810-846)
 * + the 128 x 30 log-r / phi sampling of the image plane (:1481-1535) +
 * intensite_pixel_dust with one sub-pixel (:1899-2004) + integ_ray_dust
 * (optical_depth.f90:1327-1421) -- for every observer direction of
 * mcgpu_set_rt1, from the xI_scatt the last mcgpu_run_mono(rt1=1) of this
 * wavelength left on the device (after the all-reduce on several GPUs).
 * The stellar term (compute_stars_map, :1603-1895): mcgpu_rt1_stars_map_sed below.
 * Cylindrical and spherical grids (the ray integration picks the operators of the grid at run time) and Voronoi grids
 * (move_to_grid_Voronoi + cross_Voronoi_cell; the stars' maps start from index_cell_voronoi of the point of the disc).
 * ------------------------------------------------------------------------ */
typedef struct {
  int lambda;               /* 1-based wavelength index                                   */
  double wl_um;             /* tab_lambda(lambda)                                         */
  double E_src;             /* E_totale(lambda) = E_stars + E_disk + E_ISM (:666)          */
  double n_sent_photons;    /* sum(n_phot_envoyes(lambda,:)) (:662)                       */
  double distance;          /* pc                                                         */
  double ang_disque;        /* degrees                                                    */
  int l_sym_ima;            /* half-plane sampling doubled by symmetry (:1517-1521)       */
  double tau_dark_zone_obs; /* integ_ray_dust stops beyond it (optical_depth.f90:1414)    */
  double Rmin, Rmax;        /* grid extent, AU (:1492-1493)                               */
} mcgpu_rt_opts;

/* tab_RT_az[RT_n_az] in degrees; Tdust[n_cells] (host).  stokes[RT_n_incl*RT_n_az][N_type_flux]
 * receives the dust contribution to Stokes_ray_tracing in the same units (W.m-2 once the caller
 * applies the SED normalisation of ecriture_sed_ray_tracing).  With l_sym_ima the 30 azimuths
 * sample half of the image plane with pixels of the full annulus / 30 (:1506-1511), so nothing
 * is doubled afterwards. */
int mcgpu_rt1_dust_map(mcgpu_ctx *ctx, const mcgpu_rt_opts *opts,
                       const float *tab_RT_az, const float *Tdust,
                       double *stokes, double *kernel_ms);

/* The stars in an image: compute_stars_map (dust_transfer.f90:1604-1854) with lresolved = .true. -- per observer and
 * star the 21 x 21 screen of optical depths in front of the star, then random points of the stellar sphere (1024 /
 * n_stars, or 100 per pixel of the disc when the star is wider than a pixel, :1655-1667), each placed in its pixel
 * (find_pixel, :1858-1893) with weight exp(-tau) cos_thet LimbDarkening(cos_thet) and normalised so that a star's map
 * sums to star_flux[istar] (= factor * prob_E_star(lambda, istar), :1651-1652, 1829) when its disc lies inside the map.
 * Limb darkening (llimb_darkening; n_mu > 0): the tables of read_limb_darkening_file (input.f90:628), interpolated like
 * utils.f90's interp; with pola_limb_darkening also the polarised maps Q = P cos 2 phi, U = P sin 2 phi (:1817-1823,
 * "only works for a star centered").  stars_map (npix_x, npix_y, n_maps, RT_n_incl, RT_n_az) column-major, n_maps = 3 with
 * pola_limb_darkening else 1, in double (the reference sums default reals per thread); star_position (n_stars,
 * RT_n_incl * RT_n_az, 2) in arcsec (:1847-1848), may be NULL.  The rays' positions come from the Philox stream of
 * `seed` (the reference draws them from SPRNG).  Cylindrical grids. */
int mcgpu_rt1_stars_map_image(mcgpu_ctx *ctx, const mcgpu_rt_opts *opts, const float *tab_RT_az, uint64_t seed,
                              const double *star_flux, int npix_x, int npix_y, double map_size, double zoom, int n_mu,
                              const float *mu_limb_darkening, const float *limb_darkening,
                              const float *pola_limb_darkening, double *stars_map, double *star_position);

/* init_dust_source_fct2 of one inclination on the device: described with mcgpu_set_rt2 above.  eps_dust2 /
 * eps_dust2_star may be NULL: the source function stays in HBM for the two calls below. */
int mcgpu_rt2_source(mcgpu_ctx *ctx, const mcgpu_rt_opts *opts, int p_lambda, int ibin, const float *Tdust,
                     const double *r_grid, const double *z_grid, int nang_ray_tracing, int nang_ray_tracing_star,
                     float *eps_dust2, float *eps_dust2_star, double *kernel_ms);

/* Ray tracing with method 2 (lscatt_ray_tracing2): dust_map's SED sampling and image pixels as in mcgpu_rt1_dust_map /
 * mcgpu_rt1_image, for the inclination of the last mcgpu_rt2_source (same wavelength; iaz = 1: method 2 is 2D and knows
 * one observer azimuth), with dust_source_fct's method-2 branch (dust_ray_tracing.f90:1478-1700: linear in z between the
 * cell and its vertical neighbour, linear in azimuth between the tabulated directions, interpolate_Stokes_QU :1705; the
 * radial interpolation is switched off in the reference) in integ_ray_dust (optical_depth.f90:1327-1421).
 * stokes[N_type_flux]; image(npix_x, npix_y, N_type_flux) column-major. */
int mcgpu_rt2_dust_map(mcgpu_ctx *ctx, const mcgpu_rt_opts *opts, const float *tab_RT_az, const float *Tdust,
                       double *stokes, double *kernel_ms);
int mcgpu_rt2_image(mcgpu_ctx *ctx, const mcgpu_rt_opts *opts, const float *tab_RT_az, const float *Tdust, int npix_x,
                    int npix_y, double map_size, double zoom, double *image, uint64_t *n_rays, double *kernel_ms);

/* The same for images: dust_map method 2 (dust_transfer.f90:1537-1577) -- npix_x x npix_y square pixels of
 * (map_size/zoom)/max(npix_x,npix_y) AU, each refined by intensite_pixel_dust (:1899-2004): 1, 2x2, ... 32x32
 * sub-pixel rays, at least 2 and at most 6 iterations, until Stokes I changes by less than 1 %.
 * image(npix_x, npix_y, RT_n_incl, RT_n_az, N_type_flux) column-major = Stokes_ray_tracing(lambda,:,:,:,:,:) of the
 * dust; with l_sym_ima only the columns i <= npix_x/2 + mod(npix_x,2) are computed (the rest stays 0: the
 * reference mirrors them when it writes the image, output.f90:1007-1025).  n_rays (may be NULL) returns the
 * number of rays traced.  The xI_scatt comes from an image-mode Monte Carlo: mcgpu_run_mono with
 * n_photons2 = huge and n_phot_lim = n_photons_image (run_image_mc, dust_transfer.f90:711-713: every stream
 * sends exactly n_photons_image packets). */
int mcgpu_rt1_image(mcgpu_ctx *ctx, const mcgpu_rt_opts *opts,
                    const float *tab_RT_az, const float *Tdust, int npix_x,
                    int npix_y, double map_size, double zoom, double *image,
                    uint64_t *n_rays, double *kernel_ms);

/* The stars' term of the ray-traced SED: compute_stars_map (dust_transfer.f90:1604-1854) with lresolved = .false.
 * and no limb darkening, i.e. stars_map(1,1,1) for every observer -- per star a 21 x 21 screen of optical depths
 * towards the observer (optical_length_tot, optical_depth.f90:248) and n_ray_star_SED / n_stars = 1024 / n_stars
 * random points of the stellar sphere with the screen's interpolated optical depth.
 *   star_flux[n_stars]   factor * prob_E_star(lambda, istar) of :1657-1659 and :1819 (the host's numbers)
 *   stars_flux[RT_n_incl * RT_n_az]   = sum over stars of star_flux * sum(exp(-tau) cos_thet) / sum(cos_thet)
 * Of opts only lambda and ang_disque are read.  The reference draws the points from SPRNG; here ray k of (observer q,
 * star s) is Philox block (k, 2, q * n_stars + s) of `seed`.  Cylindrical grids. */
int mcgpu_rt1_stars_map_sed(mcgpu_ctx *ctx, const mcgpu_rt_opts *opts, const float *tab_RT_az,
                            uint64_t seed, const double *star_flux, double *stars_flux);

/* The ray tracer's optical-depth maps (options -tau_map and -tau_surface, init_mcfost.f90:679-681, 1283-1291): replaces
 * `call compute_tau_map(lambda, ibin, iaz)` and `call compute_tau_surface_map(lambda, tau_surface, ibin, iaz)` at
 * dust_transfer.f90:785-786 / 799-800 for every observer direction of mcgpu_set_rt1 at once.  One ray per pixel centre,
 * sent backwards from 10 Rmax through move_to_grid:
 *   tau_map(npix_x, npix_y, RT_n_incl, RT_n_az)            optical_length_tot (optical_depth.f90:248-324) across the grid
 *   tau_surface_map(npix_x, npix_y, RT_n_incl, RT_n_az, 3) the point (AU) where physical_length (:21-182) has used up the
 *                                                          optical depth tau_surface; zeros when the ray leaves the grid or
 *                                                          ends on a star first (dust_transfer.f90:2092-2100); a cell of
 *                                                          the dark zone hands back the entry point of the cell before it
 * Default reals, column-major, the reference's arrays summed over its threads (dust_ray_tracing.f90:59-60); either may be
 * NULL.  Pixels are (map_size/zoom)/max(npix_x,npix_y) AU.  Of opts: lambda, ang_disque, Rmax.  Every grid (cylindrical,
 * spherical, Voronoi; dust classes).  The reference's call of physical_length also deposits into the radiation field (a
 * side effect of reusing the packets' routine after the Monte Carlo); the device deposits nothing. */
int mcgpu_tau_maps(mcgpu_ctx *ctx, const mcgpu_rt_opts *opts, const float *tab_RT_az, int npix_x, int npix_y,
                   double map_size, double zoom, double tau_surface, float *tau_map, float *tau_surface_map,
                   double *kernel_ms);

/* Temp_finale (thermal_emission.f90:870-906): Tdust(icell) from the summed
 * absorbed-energy grid.  E_abs == NULL uses the device accumulator. */
int mcgpu_temp_finale(mcgpu_ctx *ctx, const double *E_abs, float *Tdust);

/* Unit probes used by the parity tests (device evaluation of the operators
 * the kernel uses; n items each). */
int mcgpu_probe_cross_cell(mcgpu_ctx *ctx, int n, const double *x0,
                           const double *y0, const double *z0, const double *u,
                           const double *v, const double *w, const int *cell,
                           double *x1, double *y1, double *z1, int *next_cell,
                           double *l);
int mcgpu_probe_index_cell(mcgpu_ctx *ctx, int n, const double *x,
                           const double *y, const double *z, int *icell);
/* cross_Voronoi_cell (Voronoi.f90:839-992), all of its outputs */
int mcgpu_probe_cross_voronoi(mcgpu_ctx *ctx, int n, const double *x0,
                              const double *y0, const double *z0,
                              const double *u, const double *v, const double *w,
                              const int *cell, const int *previous_cell,
                              double *x1, double *y1, double *z1, int *next_cell,
                              double *l, double *l_contrib, double *l_void_before);
int mcgpu_probe_philox(mcgpu_ctx *ctx, const uint32_t ctr[4],
                       const uint32_t key[2], uint32_t out[4]);
int mcgpu_probe_packet_rand(mcgpu_ctx *ctx, uint64_t seed, uint64_t packet,
                            int n, float *out);

/*
 * lvariable_dust (dust settling etc.; mem.f90:213-244): the opacity and re-emission tables gain the cell axis
 * p_n_cells and the loop reads them with p_icell (optical_depth.f90:100-102, radiation_field.f90:47-53,
 * dust_transfer.f90:1284, thermal_emission.f90:659-771).  Call after mcgpu_set_opacity and mcgpu_set_thermal, with
 * the reference's own arrays and layouts:
 *   p_icell[n_cells]                        the class of every cell, 1..p_n_cells (the reference: p_icell = icell)
 *   kappa, kappa_abs_LTE (p_n_cells, n_lambda)    dust_prop.f90:17-21, class fastest
 *   tab_albedo_pos (p_n_cells, n_lambda)          grains.f90:62
 *   log_Qcool (n_T, p_n_cells)                    log_Qcool_minus_extra_heating, thermal_emission.f90:34
 *   kdB_dT_CDF (n_lambda, n_T, p_n_cells)         thermal_emission.f90:45
 *   prob_s11_pos, tab_s12_o_s11_pos ... tab_s44_o_s11_pos (0:nang_scatt, p_n_cells, n_lambda), tab_g_pos
 *   (p_n_cells, n_lambda)                         mem.f90:205-243; all seven or all NULL (NULL: every class scatters
 *                                                 with the tables of mcgpu_set_scattering)
 * p_n_cells may be smaller than n_cells (classes of cells with the same dust).  The thermal step then runs the
 * HBM-gather variant of the single-role kernel and mcgpu_temp_finale reads log_Qcool per class.  The SED mode
 * (mcgpu_run_mono: albedo, opacity and scattering tables of the crossed cell's class; with the tabulated phase function it
 * needs the cumulative tables per wavelength, i.e. mcgpu_set_scattering with p_lambda_fixed = 0, and for rt1 deposits
 * tab_s11_pos per class, mcgpu_set_variable_dust_s11), mcgpu_repartition_energie and the ray tracer (mcgpu_rt1_dust_map,
 * mcgpu_rt1_image, mcgpu_rt1_stars_map_sed), mcgpu_define_dark_zone, the diffusion fill and the random walk (mcgpu_set_mrw
 * with one row of tables per class) read per class too.
 * Grids: cylindrical (every path above) and Voronoi (the thermal step, with or without the random walk:
 * k_thermal_voro_var; mcgpu_run_mono, mcgpu_repartition_energie, mcgpu_rt1_dust_map / _image); spherical: the
 * temperature step, the SED step and the ray tracers (round 5).
 * p_n_cells = 0: off.  log_Qcool and kdB_dT_CDF may both be NULL (see mcgpu_init_reemission).
 */
int mcgpu_set_variable_dust(mcgpu_ctx *ctx, int p_n_cells, const int *p_icell, const double *kappa,
                            const double *kappa_abs_LTE, const float *tab_albedo_pos,
                            const double *log_Qcool, const double *kdB_dT_CDF, const float *prob_s11_pos,
                            const float *tab_s12_o_s11_pos, const float *tab_s22_o_s11_pos,
                            const float *tab_s33_o_s11_pos, const float *tab_s34_o_s11_pos,
                            const float *tab_s44_o_s11_pos, const float *tab_g_pos);

/* tab_s11_pos(0:nang, p_n_cells, n_lambda) of the classes (default real; mem.f90:227): the phase function, normalised
 * for the ray tracer (dust_prop.f90:1172), that the rt1 deposits read per crossed cell (tab_s11_pos(it, p_icell, p_lambda),
 * dust_ray_tracing.f90:503-512).  Needed for mcgpu_run_mono with rt1 on a variable-dust context that was set with
 * mcgpu_set_variable_dust (mcgpu_opacity builds it itself); call after mcgpu_set_variable_dust. */
int mcgpu_set_variable_dust_s11(mcgpu_ctx *ctx, const float *tab_s11_pos);

/*
 * init_reemission on the device (thermal_emission.f90:404-550, the LTE part: lines 431-452 the Planck function and its
 * temperature derivative per wavelength bin, 464-513 the cooling rate, 533-549 the re-emission CDF).  Rebuilds every
 * re-emission table of the context from the kappa_abs_LTE it already holds -- the single class of mcgpu_set_opacity
 * and, with mcgpu_set_variable_dust, every class -- on the temperature grid of mcgpu_set_thermal:
 *   log_Qcool_minus_extra_heating(T, p_icell) = log( cst_E sum_lambda kappa_abs_LTE B_lambda(T) dlambda - the same at
 *                                               tab_Temp(1) ),  -1000 where that is not positive
 *   kdB_dT_CDF(lambda, T, p_icell)            = running sum of kappa_abs_LTE dB_lambda/dT dlambda, normalised
 * so the host need not build or upload them (with lvariable_dust: n_lambda * n_T * p_n_cells doubles, 280 MB at 7000
 * cells, mem.f90:213-244).  tab_lambda / tab_delta_lambda [n_lambda] in micron (module wavelengths).  Outputs, either
 * may be NULL: the tables in the reference's layouts (of the classes when variable dust is set, else of the single
 * class).  lextra_heating / dudt (the Phantom coupling's non-radiative heating term, :486-494): mcgpu_init_reemission_ex
 * below.  Not built: the per-grain tables of the non-LTE / non-equilibrium grains (:517-532, 552-619).
 */
int mcgpu_init_reemission(mcgpu_ctx *ctx, const double *tab_lambda, const double *tab_delta_lambda,
                          double *log_Qcool, double *kdB_dT_CDF);
/* ... with the non-radiative heating of the Phantom coupling (lextra_heating, thermal_emission.f90:486-494): per class
 * (p_n_cells of them, or one) dudt and heating_norm = AU_to_m**2 * volume * kappa_factor; the floor of the cooling rate
 * becomes max(Qcool(tab_Temp(1)), dudt / heating_norm), or, ufac_implicit > 0 (ldudt_implicit),
 * max(Qcool(tab_Temp(1)), (ufac_implicit * tab_Temp(T) - dudt) / heating_norm).  MCGPU_ERR_UNSUPPORTED when the result
 * does not increase with T (the reference's "Qrad_minus_dudt is not an increasing function of T", :622-631).
 * dudt = heating_norm = NULL: mcgpu_init_reemission. */
int mcgpu_init_reemission_ex(mcgpu_ctx *ctx, const double *tab_lambda, const double *tab_delta_lambda,
                             const double *dudt, const double *heating_norm, double ufac_implicit,
                             double *log_Qcool, double *kdB_dT_CDF);

/*
 * opacity + calc_local_scattering_matrices on the device (dust_prop.f90:791-1033 and 1037-1243):
 * the opacities and scattering tables of every cell class from the grains' cross sections and Mueller matrices and
 * the local grain densities, for every wavelength (p_lambda = lambda) -- with lvariable_dust the sums over the grain
 * sizes run once per cell, and the tables (253 MB per Mueller element at 7000 classes, mem.f90:213-244) never cross
 * the bus: they are built where the kernels read them and become the context's per-class tables exactly as if
 * mcgpu_set_variable_dust had been called (the re-emission tables of the classes are left to mcgpu_init_reemission).
 *   kappa(p, lambda)          = sum_k C_ext(k, lambda) n(k, p) * AU_to_cm mum_to_cm^2,  n(k, p) = dust_density_o_n_grains(k, p) n_grains(k)
 *   tab_albedo_pos            = sum_k C_sca n / sum_k C_ext n;   kappa_abs_LTE over the grains grain_RE_LTE_start..end
 *   tab_g_pos                 (aniso_method 2) the C_sca-weighted mean of tab_g
 *   tab_s11_pos(0:nang, p, l) (aniso_method 1) sum_k tab_s11(:, k, l) S_grain(k) n(k, p); likewise S12 .. S44 (lsepar_pola),
 *                             then prob_s11_pos = the running sum of s11 sin(theta) dtheta with the unresolved forward peak in
 *                             bin 1, normalised (:1141-1154); S1x / S11; s11 dtheta / (2 pi k_sca) for the ray tracer (:1172);
 *                             aniso_method 2: the Henyey-Greenstein phase function (:1190-1194)
 * in the reference's types: default-real tables are rounded after every term.  aniso_method, lsepar_pola, nang,
 * p_lambda_fixed are the context's (mcgpu_set_scattering).  Not built: scattering_method 1 (ksca_CDF), the non-LTE /
 * non-equilibrium grain tables, lphase_function_file, loverwrite_s12, lno_scattering, lqsca_equal_qabs.
 * The grains' tables, all default real, in the reference's layouts (grains.f90:38-54, mem.f90:71-84):
 */
typedef struct mcgpu_grain_tables {
  int n_grains;                            /* n_grains_tot */
  int grain_RE_LTE_start, grain_RE_LTE_end;/* 1-based range of the grains in radiative equilibrium and LTE */
  const float *C_ext, *C_sca, *C_abs;      /* (n_grains, n_lambda) */
  const float *tab_g;                      /* (n_grains, n_lambda); aniso_method 2 only */
  const float *tab_s11, *tab_s12, *tab_s22, *tab_s33, *tab_s34, *tab_s44; /* (0:nang, n_grains, n_lambda); aniso_method 1 (s12..: lsepar_pola) */
  const float *S_grain;                    /* (n_grains) geometric cross sections */
  const double *n_grains_k;                /* (n_grains) n_grains(k), relative number of grains per size bin */
} mcgpu_grain_tables;
/* Copies of the result in the reference's layouts, any pointer may be NULL: (p_n_cells, n_lambda) and
 * (0:nang, p_n_cells, n_lambda) (prob_s11_pos: one wavelength column when p_lambda_fixed). */
typedef struct mcgpu_opacity_tables {
  double *kappa, *kappa_abs_LTE;
  float *tab_albedo_pos, *tab_g_pos;
  float *tab_s11_pos, *prob_s11_pos;
  float *tab_s12_o_s11_pos, *tab_s22_o_s11_pos, *tab_s33_o_s11_pos, *tab_s34_o_s11_pos, *tab_s44_o_s11_pos;
} mcgpu_opacity_tables;
/* p_icell[n_cells]: 1-based class of every cell (the identity when every cell has its own dust);
 * dust_density_o_n_grains (n_grains, p_n_cells), double (density.f90:32).  `out` may be NULL. */
int mcgpu_opacity(mcgpu_ctx *ctx, const mcgpu_grain_tables *grains, int p_n_cells, const int *p_icell,
                  const double *dust_density_o_n_grains, const mcgpu_opacity_tables *out);

/*
 * Scattering method 1 (lscattering_method1; dust_transfer.f90:1288-1316): at every scattering of the temperature step
 * the grain that scatters is drawn from the local population -- select_scattering_grain (dust_prop.f90:1292-1336,
 * low_mem_scattering: the CDF of C_sca(k, lambda) n(k, p_icell) is walked on the fly from the small or from the big
 * grains) -- and the packet scatters off that grain: angle_diff_theta in prob_s11(lambda, igrain, :) (scattering.f90:
 * 1387-1429) and get_Mueller_matrix_per_grain (:1302-1324), or hg(tab_g(igrain, lambda)).  The reference selects it by
 * itself when the per-cell tables of method 2 would exceed max_mem (scattering.f90:39-66): lvariable_dust with many
 * cells; the SED / image step always uses method 2 (lmono, :52-59), so mcgpu_run_mono ignores this setting.
 * Needs a variable-dust context (the opacities and albedo per class: mcgpu_set_variable_dust without scattering
 * tables, or mcgpu_opacity).  grains: C_sca, n_grains_k, and tab_g (aniso_method 2) or -- aniso_method 1 -- prob_s11
 * (n_lambda, n_grains, 0:nang) (grains.f90:53) and, with lsepar_pola, tab_s11 .. tab_s44 in the normalisation of method 1
 * (divided by s11, tab_s11 = 1: normalise_Mueller_matrix, scattering.f90:540-555); dust_density_o_n_grains
 * (n_grains, p_n_cells) with p_n_cells the context's classes.  grains = NULL: back to method 2.
 */
int mcgpu_set_scattering_method1(mcgpu_ctx *ctx, const mcgpu_grain_tables *grains, const float *prob_s11,
                                 int p_n_cells, const double *dust_density_o_n_grains);

/*
 * ksca_CDF(0:n_grains, p_n_cells, n_lambda) (dust_prop.f90:24) on the device, after mcgpu_set_scattering_method1: per class
 * and wavelength the normalised cumulative C_sca n over the grain sizes (dust_prop.f90:976-994; all ones where the sum is
 * not positive).  With it the scattering grain is selected by the dichotomy of select_grainsize_high_mem (dust_prop.f90:
 * 1245-1288) -- the branch the reference takes when the table fits max_mem (mem.f90:245-258) -- instead of the walk of the
 * low-memory mode.  ksca_CDF_out: NULL, or the table in the reference's layout.  build = 0: back to the walk.
 */
int mcgpu_build_ksca_CDF(mcgpu_ctx *ctx, int build, double *ksca_CDF_out);

/*
 * The optional accumulators of save_radiation_field's thermal branch (radiation_field.f90:54-55):
 *   xN_abs[n_cells]             path segments per cell (xN_abs(icell,1,id) with lmcfost_lib: what run_mcfost_phantom
 *                               returns, mcfost2phantom.f90:361), summed over "threads"
 *   xJ_abs[n_cells * n_lambda]  sum of l * Stokes(1) per cell and wavelength (lxJ_abs_step1), column-major (icell, lambda)
 * Switched on before a thermal launch with mcgpu_set_option(ctx, "radiation_field", bits) -- bit 0: xN_abs, bit 1:
 * xJ_abs; zeroed by a launch unless accumulate is set.  Either pointer may be NULL.
 * xN_abs is counted in 32-bit words on the device: a cell crossed more than 2^32 times between two resets wraps
 * (the reference's default-real counter stops growing at 2^24 instead); with `accumulate` over many 1e8-packet
 * launches fetch and reset it in between.
 */
int mcgpu_fetch_radiation_field(mcgpu_ctx *ctx, double *xN_abs, double *xJ_abs);

/*
 * Modified random walk (module MRW, MRW.f90; the call site dust_transfer.f90:1222-1239 is commented out in the
 * reference and make_MRW_step, MRW.f90:74-115, is an unfinished stub: this is the working form of what they
 * describe -- Min et al. 2009, Robitaille 2010 -- see DESIGN.md; PARITY UNPINNED, validated against the brute-force
 * loop).  Cylindrical and spherical grids (distance_to_closest_wall_sph, spherical_grid.f90:451-499, with the cones of
 * the polar walls in their working form), 2D and 3D, thermal step (3D: the azimuthal walls enter the distance, cylindrical_grid.f90:
 * 1198-1218 with sin / cos_phi_lim built as :586-599 -- wall 0 taken as wall n_az where the reference indexes out of
 * bounds, and the true (cos, sin) = (0, 1) where it stores the sentinel (0, 1e300) that would put a wall at phi = pi/2
 * infinitely far; the single-role kernels run it).  After more than n_interactions (reference: 5) interactions in a row
 * whose flights never left the cell, a packet that its cell has just re-emitted walks: while the distance d to the
 * closest wall (distance_to_closest_wall_cyl, cylindrical_grid.f90:1179) times the cell's mean extinction exceeds
 * gamma (gamma_MRW = 2, MRW.f90:11) it jumps to a random point of the sphere of radius d and deposits the energy of
 * the path  -log(y) (3/pi^2) chi (d + ext)^2,  zeta(y) uniform (MRW.f90:12,93-99); then it leaves as a thermal packet.
 *   zeta[n_zeta]     initialize_cumulative_zeta's table (MRW.f90:16-53), y_i = i/(n_zeta-1)
 *   chi[n_T]         mean transport extinction at tab_Temp, reference cell (the engine scales by kappa_factor):
 *                    what compute_Planck_opacities (diffusion.f90:631) calls rec_Planck_opacity
 *   kappa_dep[n_T]   mean of kappa_abs_LTE with which the walk's path deposits ("Planck_opacity")
 *   ext[n_T]         length added to d in the path (zeros: the formula of MRW.f90:99 as written)
 *   r_lim[n_rad+1]   cylindrical_grid's r_lim(0:n_rad); NULL on a Voronoi grid
 * Voronoi grids (distance_to_closest_wall_Voronoi, Voronoi.f90:996-1061): the perpendicular distance to the closest face
 * of the cell -- the reference's routine divides by the neighbour separation once too often --, 0 in cut cells and in
 * cells that touch the box, where no walk is made; the kernel is k_thermal_voro_mrw.
 * With lvariable_dust (set before this call) chi, kappa_dep and ext hold one row of n_T values per class, class after
 * class; the walk of a cell reads its class's row (single-role kernels).
 * n_zeta = 0 switches the walk off.  Counters 8 and 9 count walks and sphere steps.
 */
int mcgpu_set_mrw(mcgpu_ctx *ctx, int n_zeta, const double *zeta, const double *chi,
                  const double *kappa_dep, const double *ext, double gamma, int n_interactions,
                  const double *r_lim);

/*
 * The wavelength a walk's last step leaves its sphere with ("Only at end of MRW, to switch to MC: select new
 * wavelength", MRW.f90:108-110 -- the reference's stub does not say from which spectrum).  exit_cdf[n_T][n_lambda]
 * (one block per class with lvariable_dust), cumulative over the wavelengths like kdB_dT_CDF, rows at tab_Temp,
 * interpolated in temperature like im_reemission_LTE does; NULL (the state after mcgpu_set_mrw): the cell's emission
 * spectrum kdB_dT_CDF, i.e. kappa_abs dB/dT.  The host harness passes the spectrum of the packets IN FLIGHT in a
 * thick cell, weights dB/dT: a packet crosses the sphere in the middle of a flight, and the packets in flight are the
 * emitted ones weighted with the path 1 / kappa_abs they fly before they are absorbed.  With the emission spectrum the
 * packet prefers the opaque wavelengths, is re-absorbed next to the sphere and the walk carries too little heat
 * outwards: +2 ... +4 % in the temperature of the illuminated inner rim of a thick disk against brute force (8 + 8
 * seeds of 4e6 packets: +2.5 % over radial cells 16-23 of BASELINE config 4, -1 % behind them); with the spectrum in
 * flight 0.1 % where the statistics resolve it and at most 1.7 % ("Modified random walk").
 */
int mcgpu_set_mrw_exit_spectrum(mcgpu_ctx *ctx, const double *exit_cdf);

/*
 * define_dark_zone (optical_depth.f90:1425-1651) for a 2D cylindrical grid: steps 1-3 (the radii and heights where the
 * optical depth at `lambda` exceeds tau_max from outside) and step 4 (11 test rays from the centre of every candidate
 * cell; one ray per device thread).  r_lim[n_rad+1], r_grid / z_grid[n_cells], z_lim(n_rad, nz+1...) are module
 * cylindrical_grid's arrays.  Outputs: l_dark_zone[n_cells] (to be passed to mcgpu_set_opacity), ri_in_dark_zone,
 * ri_out_dark_zone, zj_sup_dark_zone[n_rad] (to mcgpu_temp_approx_diffusion_vertical).  The context's own dark-zone
 * flags, if any, are ignored by the rays; the flags being built are not: the reference decides the columns one after the
 * other and physical_length mirrors a ray in a cell flagged earlier (:104-112), so such a ray "does not leave" -- the
 * device repeats its pass of rays with the flags of the columns before the ray's own until nothing changes, which is the
 * sequential loop's answer.  3D and spherical grids: refused -- the reference's callers never define a dark zone there
 * (`if (lspherical.or.l3D) call no_dark_zone()`, dust_transfer.f90:290-293, 734-735, 916-917), so neither this routine
 * nor the diffusion fill below has a caller on those grids.
 */
int mcgpu_define_dark_zone(mcgpu_ctx *ctx, int lambda, double tau_max, const double *r_lim,
                           const double *r_grid, const double *z_grid, const double *z_lim,
                           unsigned char *l_dark_zone, int *ri_in_dark_zone, int *ri_out_dark_zone,
                           int *zj_sup_dark_zone);

/*
 * Temp_approx_diffusion_vertical (diffusion.f90:292-374; called after Temp_finale at dust_transfer.f90:316 / :659 when
 * define_dark_zone found a dark zone): the 1+1D diffusion approximation that refills the temperature of the dark
 * zone, column by column (clean_temperature, temperature_to_DensE, setDiffusion_coeff0/_coeff,
 * iter_Temp_approx_diffusion_vertical, DensE_to_temperature).  2D cylindrical grids.  tab_lambda / tab_delta_lambda
 * [n_lambda] in micron (module wavelengths); ri_in_dark_zone(1), ri_out_dark_zone(1), zj_sup_dark_zone(1:n_rad,1) as
 * define_dark_zone leaves them (optical_depth.f90:1459-1500, 1579-1586; module cylindrical_grid); Tdust[n_cells] in
 * and out (host).  n_iterations (may be NULL): pseudo-time steps summed over the columns.
 */
int mcgpu_temp_approx_diffusion_vertical(mcgpu_ctx *ctx, const double *tab_lambda,
                                         const double *tab_delta_lambda, int ri_in_dark_zone,
                                         int ri_out_dark_zone, const int *zj_sup_dark_zone,
                                         float *Tdust, int *n_iterations);

/*
 * The Voronoi tessellation on the device: what the reference gets from voro_C (voro++_wrapper.cpp:43-277; interface
 * Voronoi.f90:70-96, call :487-520) -- every cell built on its own from the box and the bisector planes of the sites
 * around it, ONE THREAD PER CELL (mcfost_amd/csrc/mc_tessellate.hip.h).  The candidates of a cell come from the host: knn[n_run][k],
 * 0-based site ids by increasing distance (a kd-tree search; -1 pads a short list).  A cell is complete once the next
 * candidate is farther than twice its farthest vertex; otherwise n_neigh = -1 and the host calls again for those cells
 * (cells[n_run]: the cells to build, NULL = 0 .. n_run - 1) with a larger k.  n_neigh = -2: more faces / vertices than
 * the kernel holds (96 / 160) or than max_neighbours.
 * knn_first[n_run + 1] (else NULL): the candidates are ragged rows knn[knn_first[r] .. knn_first[r + 1]) that hold EVERY
 * site that can cut cell r -- its Delaunay neighbours, e.g. qhull's -- and the security radius is not needed: the mode for
 * point sets with voids (the cells at a disk's surface reach far into the void and would want every site as candidate).
 *   xyz[n][3], h[n]            the sites (stars last, h = huge) and their smoothing lengths
 *   limits[6]                  the box (Voronoi.f90:1275-1280); every site strictly inside
 *   threshold, n_vectors, cutting_vectors[n_vectors][3], cutting_distance_o_h
 *                              the cut of elongated cells by the Platonic solid (init_Platonic_Solid, Voronoi.f90:108-192;
 *                              voro++_wrapper.cpp:209-227): when the farthest vertex is beyond threshold * h;
 *                              n_vectors <= 12 (the reference's dodecahedron, Voronoi.f90:243; a solid whose planes meet
 *                              more than three to a vertex -- 20 faces -- is MCGPU_ERR_UNSUPPORTED)
 *   extra_plane[n_run][4]      NULL, or per cell one more cut, unit normal + distance (<= 0: none): the stellar surface
 *                              for a star's neighbours (voro++_wrapper.cpp:229-262)
 * Outputs, per cell of the call: n_neigh, neigh[max_neighbours] (0-based site ids; -1 .. -6 = walls -x +x -y +y -z +z) --
 * the faces BEFORE the cuts, like the reference's list (:195-207) --, volume AFTER the cuts, delta_edge (the farthest
 * vertex before the cuts), was_cut.  kernel_ms: the kernel alone.  volume_uncut (or NULL): the volume before the cuts --
 * what a density estimate m / V of the particle wants, while the cell's mass is rho times the cut volume.
 */
int mcgpu_voronoi_tesselation(int device, int n, const double *xyz, const double *h, const double limits[6],
                              double threshold, int n_vectors, const double *cutting_vectors,
                              double cutting_distance_o_h, int n_run, const int *cells, int k, const int *knn,
                              const int *knn_first, const double *extra_plane, int max_neighbours, int *n_neigh, int *neigh,
                              double *volume, double *delta_edge, unsigned char *was_cut, double *kernel_ms,
                              double *volume_uncut);

/*
 * Several GPUs of one node behind ONE host thread -- the reference's host is a single OpenMP
 * process (mcfost.f90; the Phantom caller mcfost2phantom.f90:159), so this is the entry it binds.
 * mcgpu_multi_create opens one context per device and one RCCL communicator over them
 * (devices = NULL: 0..n_dev-1).  The host uploads the model to every context
 * (mcgpu_multi_ctx(m, i) with the mcgpu_set_* calls above: tables are replicated).
 * mcgpu_multi_run_thermal replaces the OpenMP region of run_thermal_mc (dust_transfer.f90:617):
 * opts->n_packets is the GLOBAL packet count; device i runs the contiguous id shard
 * mcgpu_shard_packets(n, i, n_dev) with n_replicas = n_dev (the `* nb_proc` of
 * thermal_emission.f90:670), then ONE ncclAllReduce sums the fused accumulator
 * [E_abs | sed | n_sent | counters] over xGMI and the outputs are read from device 0.
 * kernel_ms = the slowest device's packet loop.  With n_dev = 1 the result equals
 * mcgpu_run_thermal's (and no communicator is created: the RCCL communicators are opened by the
 * first call that has something to reduce; mcgpu_multi_rccl_ranks reads ncclCommCount back).
 * opts->accumulate = 1: after the last call's in-place all-reduce every device holds the global
 * sums, so each device first scales what it holds by 1 / n_dev (exact for 2, 4, 8 devices) and
 * devices > 0 clear their event counters -- the next all-reduce then returns old totals + new
 * parts, and the in-flight temperature's `local * n_replicas` stays the right estimate.
 * An error on one device leaves nothing of the call running on the others.
 *
 * mcgpu_multi_create_ex(..., MCGPU_MULTI_SHARED_DEVICE, ...) opens the n_dev contexts on ONE device
 * (devices[i] all equal; NULL: device 0) and sums with the library's own kernel in place of
 * ncclAllReduce, which refuses a communicator that names a device twice.  Everything else --
 * shards, n_replicas, the 1 / n_dev rescale of an accumulating call, the counters, the error
 * path, the chunked 3D launch on several contexts, the host threads of mcgpu_multi_run_mono -- is
 * the code of distinct devices, so a box with one GPU executes it (tests/test_multi_shared_device.py,
 * `bench.py --gpus N --shared-device`).  Not for production: the contexts share the device's CUs.
 * mcgpu_multi_reductions counts the collectives a handle has executed (either kind).
 * mcgpu_multi_create_ex(..., MCGPU_MULTI_FORCE_RCCL, ...) keeps distinct devices and RCCL but does
 * not skip the collective when n_dev = 1: ncclCommInitAll over one rank and the grouped
 * ncclAllReduce of every buffer a call deposits into (the fused accumulator, xI_scatt in either
 * precision, I_spec / I_spec_star) run as they would on eight devices; the sum over one rank leaves
 * the buffers bit for bit, mcgpu_multi_rccl_ranks returns 1 (tests/test_rccl_single_rank.py).
 *
 * mcgpu_multi_run_mono replaces `call mc_photon_loop(lambda, ...)` of the SED loop
 * (dust_transfer.f90:939) the same way: the opts->n_chunks independent streams are split into
 * contiguous ranges (n_dev <= n_chunks), every device runs mcgpu_run_mono on its range (one host
 * thread per device inside the call), then one all-reduce sums [sed | n_sent | counters] and one
 * xI_scatt.  Afterwards every context holds the sums: read them with mcgpu_fetch / mcgpu_fetch_xI
 * on mcgpu_multi_ctx(m, 0) and run mcgpu_rt1_dust_map on any device.  n_sent_chunk[n_chunks] as
 * in mcgpu_run_mono.  opts->rt1 = 2 (ray tracing method 2): the second all-reduce sums I_spec and
 * I_spec_star instead, so that mcgpu_rt2_source on any device sees the global field.
 */
typedef struct mcgpu_multi mcgpu_multi;
#define MCGPU_MULTI_SHARED_DEVICE 1u
#define MCGPU_MULTI_FORCE_RCCL 2u
int mcgpu_multi_create(int n_dev, const int *devices, mcgpu_multi **out);
int mcgpu_multi_create_ex(int n_dev, const int *devices, unsigned int flags, mcgpu_multi **out);
uint64_t mcgpu_multi_reductions(const mcgpu_multi *m);
int mcgpu_multi_destroy(mcgpu_multi *m);
int mcgpu_multi_size(const mcgpu_multi *m);
mcgpu_ctx *mcgpu_multi_ctx(mcgpu_multi *m, int i);
const char *mcgpu_multi_last_error(const mcgpu_multi *m);
void mcgpu_shard_packets(uint64_t n_packets, int rank, int world, uint64_t *first, uint64_t *count);
int mcgpu_multi_run_thermal(mcgpu_multi *m, const mcgpu_run_opts *opts, double *E_abs, double *sed,
                            double *n_sent, uint64_t *counters, double *kernel_ms);
int mcgpu_multi_run_mono(mcgpu_multi *m, const mcgpu_mono_opts *opts, double frac_E_stars,
                         double frac_E_disk, const double *prob_E_cell, uint64_t *n_sent_chunk,
                         double *kernel_ms);

/*
 * The SED step on several devices, sharded BY WAVELENGTH: run_sed_mc's own loop (dust_transfer.f90:899-1027: per wavelength
 * repartition_energie :924, the packet loop :939, then dust_map of every observer) is the partition -- a wavelength's
 * xI_scatt is produced and consumed within that wavelength.  Device d takes whole wavelengths (longest first by `cost`,
 * each to the device with the least work so far), builds the wavelength's emission tables from Tdust
 * (mcgpu_repartition_energie), runs the scout / commit passes (mcgpu_run_mono with opts' stream count, stop bin and
 * limits) and, with `rt` and opts->rt1 = 1, the ray-traced SED of the dust (mcgpu_rt1_dust_map); only the results travel:
 *   sed[n_wl][MCGPU_N_SED_TYPES][N_phi][N_thet]   the wavelength's own bins of the nine Monte Carlo SED arrays
 *   n_sent[n_wl], E_disk[n_wl], counters[n_wl][MCGPU_N_COUNTERS]
 *   stokes_rt[n_wl][RT_n_incl * RT_n_az][N_type_flux]   (rt / stokes_rt NULL: no ray tracing)
 *   device_of[n_wl] (which context ran it), seconds[n_wl] (its wall time: next call's `cost`)        -- any may be NULL.
 * Nothing is summed across devices and no xI_scatt is reduced (mcgpu_multi_run_mono, which splits ONE wavelength's streams,
 * all-reduces the whole array: 200 MB per wavelength on ref4.1 with ten observers): a wavelength's results are those of
 * the single-device calls whichever device ran it, to the order of its own atomic sums.  Every context must hold the model
 * and mcgpu_set_rt1.  opts->lambda, p_lambda and seed are taken from wl[i]; rt's lambda, wl_um, E_src and n_sent_photons
 * are filled in per wavelength (E_src = E_star + E_disk + E_ISM, :666).
 */
typedef struct {
  int lambda, p_lambda;     /* 1-based; p_lambda = 0: lambda                                   */
  double wl_um;             /* tab_lambda(lambda)                                              */
  double E_star, E_ISM;     /* E_stars(lambda), E_ISM(lambda)                                  */
  uint64_t seed;            /* of this wavelength's packet streams                             */
  double cost;              /* relative cost (e.g. last time's seconds[i]); 0: unknown (equal) */
} mcgpu_sed_wavelength;
int mcgpu_multi_run_sed(mcgpu_multi *mm, const mcgpu_mono_opts *opts, int n_wl, const mcgpu_sed_wavelength *wl,
                        const float *Tdust, const mcgpu_rt_opts *rt, const float *tab_RT_az, double *sed, double *n_sent,
                        double *E_disk, double *stokes_rt, uint64_t *counters, int *device_of, double *seconds);
int mcgpu_multi_rccl_ranks(mcgpu_multi *m);

#ifdef __cplusplus
}
#endif
#endif
