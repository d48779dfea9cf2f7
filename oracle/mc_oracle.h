/*
 * mc_oracle.h -- CPU ORACLE for the MCFOST continuum Monte Carlo packet loop.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (cpinte/mcfost 4.1.13, Fortran) used as the checker for the HIP
 * engine.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.  The product (libmcfost_hip.so) never links or calls it.
 *
 * Parity status ("pinning"):
 *   - geometry (cell mapping, grid definition, cell crossing, point location,
 *     move-to-grid, emission position), the temperature grid and the
 *     wavelength grid are PINNED bit-for-bit against the reference's own
 *     routines compiled from /root/reference/src (see oracle/ref_build/).
 *   - the sampling / thermal routines that live in reference modules which
 *     cannot be built in this image (they need SPRNG, generated sha.f90 /
 *     operating_system.f90) are restated from source and pinned only by
 *     analytic known-answer tests: PARITY UNPINNED for those.
 *
 * All "file:line" citations are into /root/reference/src/.
 */
#ifndef MC_ORACLE_H
#define MC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_N_SED_TYPES 9 /* sed, sed_q, sed_u, sed_v, n_phot_sed, sed_star,
                                sed_star_scat, sed_disk, sed_disk_scat
                                (output.f90:572-592) */
#define ORACLE_N_COUNTERS 10
enum {
  ORC_CNT_PACKETS = 0,     /* packets launched                              */
  ORC_CNT_CROSSINGS = 1,   /* cross_cell calls                              */
  ORC_CNT_FLIGHTS = 2,     /* physical_length calls                         */
  ORC_CNT_SCATT = 3,       /* scattering events                             */
  ORC_CNT_ABS = 4,         /* absorb + re-emit events                       */
  ORC_CNT_ESCAPED = 5,     /* packets binned by capteur                     */
  ORC_CNT_KILLED_STAR = 6, /* packets that hit a star                       */
  ORC_CNT_DARK = 7,        /* dark-zone mirror events                       */
  ORC_CNT_MRW_WALKS = 8,   /* modified random walks (each ends in one re-emission) */
  ORC_CNT_MRW_STEPS = 9    /* sphere steps of those walks                   */
};

/* One star (parameters.f90:230-238). Lengths in AU. */
typedef struct {
  double x, y, z, r;
  int icell;     /* cell holding the star centre (stars.f90:789-808) */
  int out_model; /* star outside the grid -> move_to_grid on emission */
} oracle_star;

/*
 * The model: every array the packet loop reads.  Column-major, 1-based cell
 * ids exactly as the Fortran host holds them.
 */
typedef struct {
  /* ---- cylindrical grid (cylindrical_grid.f90:20-35) ---- */
  int n_rad, nz, n_az, l3D;
  int n_cells; /* grid.f90:276-283 */
  int ntot2;   /* real + virtual cells (cylindrical_grid.f90:73-88) */
  int jdim_lo; /* jstart2: lowest j index of cell_map */
  int jdim_n;  /* number of j slots in cell_map (jend2-jstart2+1) */
  const double *r_lim_2;     /* [0..n_rad] */
  const double *zmax;        /* [n_rad] */
  const double *z_lim;       /* (n_rad, nz+2) */
  const double *tan_phi_lim; /* [n_az] */
  double zmaxmax, Rmax2;
  const int *cell_map;   /* (0:n_rad+1, jdim_lo:nz+1, 1:n_az) */
  const int *cell_map_i; /* [ntot2] */
  const int *cell_map_j;
  const int *cell_map_k;
  const int *lexit_cell; /* [ntot2] */
  const double *volume;  /* [n_cells] AU^3 */

  /* ---- stars ---- */
  int n_stars;
  const oracle_star *stars;

  /* ---- opacities: one cell class (p_n_cells = 1), dust_prop.f90:17-21 ---- */
  int n_lambda;
  const double *kappa;         /* [n_lambda] AU^-1 at the reference cell */
  const double *kappa_abs_LTE; /* [n_lambda] */
  const float *albedo;         /* tab_albedo_pos [n_lambda] (grains.f90:62) */
  const double *kappa_factor;  /* [n_cells] */
  const unsigned char *l_dark_zone; /* [n_cells] or NULL */

  /* ---- scattering (grains.f90:62-64) ---- */
  int nang_scatt;    /* 180 */
  int aniso_method;  /* 1 = tabulated phase function, 2 = HG */
  int lisotropic;    /* dust_transfer.f90:1323-1326 */
  int lsepar_pola;   /* update Stokes on scattering */
  int p_lambda_fixed; /* !=0: sample angle CDF at wavelength index 1
                         (reference behaviour, dust_transfer.f90:491-502) */
  const float *prob_s11_pos; /* (0:nang, n_lambda) */
  const float *s12_o_s11;    /* (0:nang, n_lambda) each */
  const float *s22_o_s11;
  const float *s33_o_s11;
  const float *s34_o_s11;
  const float *s44_o_s11;
  const float *tab_g_pos; /* [n_lambda] */

  /* ---- thermal tables (thermal_emission.f90:34-60) ---- */
  int n_T;
  const float *tab_Temp;     /* [n_T] (Temperature.f90:23) */
  const double *log_Qcool;   /* log_Qcool_minus_extra_heating [n_T] */
  const double *kdB_dT_CDF;  /* (n_lambda, n_T) */
  const double *spectre_emission_cumul; /* [0..n_lambda] */
  const double *frac_E_stars; /* [n_lambda] */
  const double *frac_E_disk;  /* [n_lambda] */
  const double *CDF_E_star;   /* (n_lambda, 0:n_stars) */
  const double *prob_E_cell;  /* (0:n_cells, n_lambda) or NULL */
  double L_packet_th;
  float T_min;

  /* ---- SED binning (output.f90:294-597) ---- */
  int N_thet, N_phi;
  int l_sym_centrale, l_sym_axiale;

  /* 0 = reference-literal arithmetic (pinned bit-for-bit to oracle/_ref).
   * 1 = after a 3D crossing of the midplane (zlim = 0, no grid_prec margin)
   *     put z1 at sign(grid_prec, w): the reference's own correction for
   *     z1 == 0 (cylindrical_grid.f90:1158-1165) applied to every rounding
   *     residue of z0 + t*w, whose SIGN is otherwise decided by the last ulp
   *     (FMA or not, libm) and can leave the packet on the wrong side of the
   *     midplane for one cell.  The engine's default; see DESIGN.md. */
  int midplane_snap;

  /* ---- Voronoi grid (Voronoi.f90:23-67); grid_type 3, otherwise cylindrical ---- */
  int grid_type;
  const float *v_xyz;      /* Voronoi_xyz(3,n_cells), default real */
  const double *v_xyz_dp;  /* Voronoi(:)%xyz */
  const double *v_h;       /* Voronoi(:)%h */
  const int *v_first;      /* first_neighbour (1-based into v_neigh) */
  const int *v_last;       /* last_neighbour */
  const int *v_neigh;      /* neighbours_list: cell id > 0, or -iwall */
  const unsigned char *v_was_cut;
  const unsigned char *v_is_star_neighbour;
  const float *v_walls;    /* 6 x (x1,x2,x3,x4) (Voronoi.f90:1275-1280) */
  double v_cut_o_h;        /* PS%cutting_distance_o_h */
  const int *v_wall_first; /* [7] offsets (0-based) into v_wall_cells */
  const int *v_wall_cells; /* wall(iwall)%neighbour_list, concatenated */

  /* ---- spherical grid (grid_type 2: spherical_grid.f90; arrays of cylindrical_grid.f90:28-31).  The cell mapping,
   * r_lim_2, tan_phi_lim, volume are the fields above (build_cylindrical_cell_mapping serves both grids). ---- */
  const double *tan_theta_lim; /* [0..nz] */
  const double *theta_lim;     /* [0..nz] */
  const double *r_lim_3;       /* [0..n_rad] */

  /* ---- interstellar radiation field: emitting sphere (stars.f90:27-28, 655-666) ---- */
  double R_ISM;
  double centre_ISM[3];

  /* ---- ray-tracing method 1 (dust_ray_tracing.f90:17-40, 80-160) ---- */
  int RT_n_incl, RT_n_az;   /* observer directions */
  const double *tab_u_rt;   /* (RT_n_incl, RT_n_az) */
  const double *tab_v_rt;   /* (RT_n_incl, RT_n_az) */
  const double *tab_w_rt;   /* [RT_n_incl] */
  int n_az_rt, n_theta_rt;  /* 45, 2 in 2D; 1, 1 in 3D (:91-98) */
  int N_type_flux;          /* init_mcfost.f90:1603-1616 */
  int lsepar_contrib;
  const float *tab_s11_pos; /* (0:nang_scatt, n_lambda): tab_s11_pos(:,1,p_lambda) */

  /* ---- modified random walk (MRW.f90, dust_transfer.f90:1222-1239; Min et al. 2009, Robitaille 2010).  The
   * reference's routine is an unfinished stub that is never called (SURVEY finding 2): this is the working
   * algorithm its comments describe -- PARITY UNPINNED, validated against the brute-force loop.  mrw = 0: off. ---- */
  int mrw;
  int mrw_n_zeta;              /* n = 10000 (MRW.f90:8) */
  const double *mrw_zeta;      /* zeta(y_i), y_i = (i-1)/(n-1): initialize_cumulative_zeta (MRW.f90:16-53) */
  const double *mrw_chi;       /* [n_T] mean transport extinction at tab_Temp, reference cell (AU^-1): the "rec_Planck_opacity"
                                  of make_MRW_step, D = 1/(3 chi kappa_factor) */
  const double *mrw_kappa_dep; /* [n_T] mean absorption opacity the walk deposits with (units of kappa_abs_LTE) */
  const double *mrw_ext;       /* [n_T] extrapolation length of the sphere radius, reference cell (AU); zeros: none */
  const double *mrw_exit_cdf;  /* [n_T][n_lambda] cumulative spectrum of a packet IN FLIGHT in a thick cell at tab_Temp (weights
                                  dB/dT: the emission spectrum kappa_abs dB/dT times the path 1 / kappa_abs a packet flies at
                                  that wavelength before it is absorbed), which the walk's last step leaves the sphere with;
                                  NULL: the emission spectrum kdB_dT_CDF (round 3) */
  float mrw_gamma;             /* gamma_MRW = 2 (MRW.f90:11) */
  int mrw_n_inter;             /* a walk may start after more than this many interactions in one cell: 5 (:1223) */
  /* ---- lvariable_dust (mem.f90:213-244): tables with the cell axis p_n_cells, in the reference's layouts.  p_n_cells = 0:
   * one class (the tables above).  Thermal step only. ---- */
  int p_n_cells;
  const int *p_icell;             /* [n_cells] 1..p_n_cells */
  const double *v_kappa;          /* kappa(p_n_cells, n_lambda) */
  const double *v_kappa_abs_LTE;  /* (p_n_cells, n_lambda) */
  const float *v_albedo;          /* tab_albedo_pos(p_n_cells, n_lambda) */
  const double *v_log_Qcool;      /* (n_T, p_n_cells) */
  const double *v_kdB_dT_CDF;     /* (n_lambda, n_T, p_n_cells) */
  /* optional (all or none; v_prob_s11_pos == NULL: the single-class scattering tables above) */
  const float *v_prob_s11_pos;    /* (0:nang, p_n_cells, n_lambda) */
  const float *v_s12_o_s11, *v_s22_o_s11, *v_s33_o_s11, *v_s34_o_s11, *v_s44_o_s11; /* likewise */
  const float *v_tab_g_pos;       /* (p_n_cells, n_lambda) */
  const double *r_lim;         /* [0..n_rad] (cylindrical_grid.f90:22), read by distance_to_closest_wall_cyl */
  const float *v_tab_s11_pos;     /* (0:nang, p_n_cells, n_lambda): the classes' phase function for rt1 (SED mode; may be NULL) */
  /* scattering method 1 (lscattering_method1, dust_transfer.f90:1288-1316): the scattering grain is drawn from the cell's
   * population (select_scattering_grain, dust_prop.f90:1292-1336, low_mem_scattering), then its own tables are used */
  int scattering_method1, m1_n_grains;
  const float *m1_C_sca;          /* C_sca(n_grains, n_lambda) */
  const double *m1_nk;            /* n_grains(k) */
  const double *m1_dens;          /* dust_density_o_n_grains(n_grains, p_n_cells) (one column without lvariable_dust) */
  const double *m1_ksca_CDF;      /* NULL (low_mem_scattering), or ksca_CDF(0:n_grains, p_n_cells, n_lambda) (dust_prop.f90:24): the
                                     grain is then selected by select_grainsize_high_mem (oracle_build_ksca_CDF fills it) */
  const float *m1_prob_s11;       /* prob_s11(n_lambda, n_grains, 0:nang) (grains.f90:53) */
  const float *m1_tab_g;          /* tab_g(n_grains, n_lambda) */
  const float *m1_s11, *m1_s12, *m1_s22, *m1_s33, *m1_s34, *m1_s44; /* tab_s1x(0:nang, n_grains, n_lambda), per-grain normalisation (s11 = 1) */
  /* sin_phi_lim, cos_phi_lim(n_az) (cylindrical_grid.f90:30, 586-599): the 3D branch of distance_to_closest_wall_cyl; may be NULL */
  const double *sin_phi_lim, *cos_phi_lim;
} oracle_model;

/* Run options. */
typedef struct {
  uint64_t seed;
  uint64_t first_packet; /* global id of the first packet of this run */
  uint64_t n_packets;
  int n_threads;         /* OpenMP threads; per-thread accumulators like the
                            reference (radiation_field.f90:20-27) */
  int frozen;            /* 0: live Bjorkman&Wood feedback from the running
                               accumulator (reference);
                            1: Temp_LTE reads E_prior (deterministic mode) */
  int tau_fp32;          /* 1: tau = -log(1-rand) in default real like
                               dust_transfer.f90:1208-1215; 0: same formula in
                               FP64 (used for bit-parity with the device) */
  double n_replicas;     /* in-flight Qheat scale on top of n_threads (number
                            of GPUs/ranks sharing the job); >=1 */
} oracle_opts;

/* xN_abs[n_cells] / xJ_abs[n_cells*n_lambda] (radiation_field.f90:54-55) filled by the next oracle_run_thermal calls;
 * NULL switches them off again.  The caller zeroes them. */
void oracle_set_radiation_field_outputs(double *xN_abs, double *xJ_abs);

/* Cell mapping sizes + builder (cylindrical_grid.f90:45-179). */
void oracle_cell_mapping_sizes(int n_rad, int nz, int n_az, int l3D,
                               int *n_cells, int *ntot2, int *jdim_lo,
                               int *jdim_n);
int oracle_build_cell_mapping(int n_rad, int nz, int n_az, int l3D,
                              int *cell_map, int *cell_map_i, int *cell_map_j,
                              int *cell_map_k, int *lexit_cell);

/* Geometry operators (cylindrical_grid.f90). */
int oracle_test_exit_grid_cyl(const oracle_model *m, int icell, double x,
                              double y, double z);
void oracle_index_cell_cyl(const oracle_model *m, double x, double y, double z,
                           int *icell);
void oracle_cross_cylindrical_cell(const oracle_model *m, double x0, double y0,
                                   double z0, double u, double v, double w,
                                   int cell, int previous_cell, double *x1,
                                   double *y1, double *z1, int *next_cell,
                                   double *l, double *l_contrib,
                                   double *l_void_before);
void oracle_move_to_grid_cyl(const oracle_model *m, double *x, double *y,
                             double *z, double u, double v, double w,
                             int *icell, int *lintersect);
void oracle_pos_em_cell_cyl(const oracle_model *m, int icell, float rand1,
                            float rand2, float rand3, double *x, double *y,
                            double *z);

/* Geometry operators of the spherical grid (spherical_grid.f90): PINNED bit-for-bit to the reference's
 * module compiled in oracle/_ref (tests/test_ref_geometry.py, tests/golden/geom_sph*.npz). */
int oracle_test_exit_grid_sph(const oracle_model *m, int icell);
void oracle_index_cell_sph(const oracle_model *m, double x, double y, double z, int *icell);
void oracle_cross_spherical_cell(const oracle_model *m, double x0, double y0, double z0, double u, double v,
                                 double w, int cell, int previous_cell, double *x1, double *y1, double *z1,
                                 int *next_cell, double *l, double *l_contrib, double *l_void_before);
void oracle_move_to_grid_sph(const oracle_model *m, double *x, double *y, double *z, double u, double v, double w,
                             int *icell, int *lintersect);
void oracle_pos_em_cell_sph(const oracle_model *m, int icell, float rand1, float rand2, float rand3, double *x,
                            double *y, double *z);

/*
 * One wavelength of the SED Monte Carlo (run_sed_mc, dust_transfer.f90:828-1042 ->
 * mc_photon_loop with lmono and not lmono0): n_chunks = n_photons_loop sequential
 * streams; a stream sends packets until n_photons2 of them were binned in
 * inclination bin capt_sup or n_phot_lim were sent (:526).
 */
typedef struct {
  uint64_t seed;
  int lambda, p_lambda;
  int n_chunks;        /* n_photons_loop */
  int first_chunk;     /* global id of the first stream */
  double n_photons2;   /* n_photons_lambda */
  double n_phot_lim;   /* n_photons_lim */
  int capt_sup;
  int rt1;             /* 1: lscatt_ray_tracing1, deposit xI_scatt; 2: lscatt_ray_tracing2 (2D), deposit I_spec / I_spec_star */
  int n_threads;
  /* ray tracing method 2 (radiation_field.f90:91-129): the caller's arrays, zeroed by it, in the reference's layout
   * I_spec(N_type_flux, n_theta_I, n_phi_I, n_cells), I_spec_star(n_cells) -- in double (the reference's are default real) */
  int n_theta_I, n_phi_I;
  double *I_spec, *I_spec_star;
} oracle_mono_opts;

/* xI_scatt(n_az_rt, n_theta_rt, N_type_flux, RT_n_incl*RT_n_az, n_cells) summed over
 * threads (FP64 here; the reference keeps default real per thread); sed/n_sent as in
 * oracle_run_thermal (only the lambda slice is touched); n_sent_chunk[n_chunks] =
 * packets each stream sent.  Packet (chunk c, sequence s) uses the random stream
 * of id ((first_chunk + c) << 40) | s. */
int oracle_run_mono(const oracle_model *m, const oracle_mono_opts *o, double *xI_scatt,
                    double *sed, double *n_sent, uint64_t *n_sent_chunk,
                    uint64_t *counters);

/*
 * Ray-traced SED of the dust, ray-tracing method 1: what dust_map(lambda,ibin,iaz)
 * (dust_transfer.f90:1413-1600) adds to Stokes_ray_tracing(lambda,1,1,ibin,iaz,:) with RT_sed_method = 1,
 * for every observer direction: calc_Jth (dust_ray_tracing.f90:810-905, LTE grains),
 * init_dust_source_fct1 (:636-716), the 128 x 30 image-plane sampling of dust_map, intensite_pixel_dust
 * (:1899-2004, one ray per pixel), integ_ray_dust (optical_depth.f90:1327-1421) with dust_source_fct
 * (dust_ray_tracing.f90:1442-1475, RT1 branch).  The stellar term (compute_stars_map) is not included.
 */
typedef struct {
  int lambda;
  double wl_um;              /* tab_lambda(lambda) */
  double E_src;              /* E_stars(lambda) + E_disk(lambda) */
  double n_sent_photons;     /* sum(n_phot_envoyes(lambda,:)) */
  double distance;           /* pc */
  double ang_disque;         /* degrees, after init_mcfost.f90:1781 */
  int l_sym_ima;
  double tau_dark_zone_obs;  /* parameters.f90:25 */
  double Rmin, Rmax;         /* grid.f90:234-235 */
  const float *tab_RT_az;    /* [RT_n_az] degrees */
  int n_threads;
} oracle_rt_opts;

/* Ray tracing method 2, the source function of inclination ibin: init_dust_source_fct2 (dust_ray_tracing.f90:717-806)
 * = calc_Isca_rt2_star (:1245-1440, with angles_scatt_rt2 :304-405) + calc_Isca_rt2 (:907-1240) + calc_Jth + the division
 * by kappa_ext and the (Q, U) -> (P, angle) form.  2D grids.  In: I_spec(N_type_flux, n_theta_I, n_phi_I, n_cells) and
 * I_spec_star(n_cells) summed over the threads (double), o->E_src / n_sent_photons / wl_um / lambda, Tdust, r_grid /
 * z_grid(n_cells).  Out (default real, the reference's arrays): eps_dust2(N_type_flux, nang_rt, 0:1, n_cells),
 * eps_dust2_star(n_Stokes, nang_star, 0:1, n_cells).  PARITY UNPINNED (dust_ray_tracing.f90 is unbuildable here). */
int oracle_init_dust_source_fct2(const oracle_model *m, const oracle_rt_opts *o, int p_lambda, int ibin, int n_theta_I,
                                 int n_phi_I, int nang_rt, int nang_star, const double *I_spec, const double *I_spec_star,
                                 const float *Tdust, const double *r_grid, const double *z_grid, float *eps_dust2,
                                 float *eps_dust2_star);

/* Ray tracing method 2, the ray integration: while a source function is set (eps_dust2 / eps_dust2_star of inclination
 * ibin from oracle_init_dust_source_fct2, z_grid(n_cells)), oracle_dust_map_sed and oracle_dust_map_image integrate that
 * inclination (iaz = 1) with dust_source_fct's method-2 branch (dust_ray_tracing.f90:1478-1700: linear in z between the
 * cell and its vertical neighbour, linear in azimuth between the tabulated directions, interpolate_Stokes_QU :1705) instead
 * of method 1's eps_dust1 -- xI may then be NULL; the other observers' outputs stay 0.  eps_dust2 = NULL: off. */
void oracle_set_rt2_source(const float *eps_dust2, const float *eps_dust2_star, int nang_rt, int nang_star, int ibin,
                           const double *z_grid);

/* compute_stars_map for images (dust_transfer.f90:1604-1854, lresolved = .true.; find_pixel :1858-1893; interp
 * utils.f90:130-175): the stars' discs, limb-darkened (n_mu > 0) and polarised (pola_ld) if asked, in the pixel maps of
 * the observers; see the definition.  PARITY UNPINNED like oracle_stars_map_sed. */
int oracle_stars_map_image(const oracle_model *m, const oracle_rt_opts *o, uint64_t seed, const double *star_flux,
                           int npix_x, int npix_y, double map_size, double zoom, int n_mu, const float *mu_ld,
                           const float *ld, const float *pola_ld, double *map, double *star_position);

/* compute_tau_map (dust_transfer.f90:2114-2210) and compute_tau_surface_map (:2006-2110) for every observer direction:
 * tau_map(npix_x, npix_y, RT_n_incl, RT_n_az) and tau_surface_map(npix_x, npix_y, RT_n_incl, RT_n_az, 3), default reals,
 * column-major; either may be NULL.  Of o: lambda, ang_disque, Rmax, tab_RT_az.  See the definition.  PARITY UNPINNED
 * (dust_transfer.f90 is unbuildable here); known answers in tests/test_tau_maps.py. */
int oracle_tau_maps(const oracle_model *m, const oracle_rt_opts *o, int npix_x, int npix_y, double map_size, double zoom,
                    float tau, float *tau_map, float *tau_surface_map);

/* compute_stars_map for the SED (dust_transfer.f90:1604-1854; lresolved = .false., no limb darkening): out[nRT] =
 * sum over stars of star_flux[istar] * sum(exp(-tau) cos_thet) / sum(cos_thet); 2D / 3D cylindrical grids.  PARITY
 * UNPINNED (dust_transfer.f90 is unbuildable here; the reference's points come from SPRNG). */
int oracle_stars_map_sed(const oracle_model *m, const oracle_rt_opts *o, uint64_t seed, const double *star_flux,
                         double *out);

/* xI_scatt: reference layout, FP64 (oracle_run_mono's output); Tdust [n_cells];
 * out[(ibin-1 + RT_n_incl*(iaz-1)) * N_type_flux + type-1] */
int oracle_dust_map_sed(const oracle_model *m, const oracle_rt_opts *o, const double *xI_scatt,
                        const float *Tdust, double *out);
/* dust_map method 2 (images): see mc_oracle.c; n_rays (may be NULL) returns the rays traced */
int oracle_dust_map_image(const oracle_model *m, const oracle_rt_opts *o, int npix_x, int npix_y, double map_size,
                          double zoom, const double *xI, const float *Tdust, double *image, int *n_rays);


/*
 * define_dark_zone (optical_depth.f90:1425-1651) for a 2D cylindrical grid: cells from which a
 * ray of optical depth tau_max does not leave the grid in any of 11 directions, plus the cells
 * below them.  r_lim[0..n_rad], r_grid / z_grid [n_cells] are cylindrical_grid's arrays.  A host
 * table builder of the reference, restated here so that the harness can feed
 * the packet loop the same kind of dark zone the reference's own ref4.1 run has.
 */
int oracle_define_dark_zone(const oracle_model *m, int lambda, double tau_max, const double *r_lim,
                            const double *r_grid, const double *z_grid, unsigned char *l_dark_zone);

/* distance_to_closest_wall_cyl (cylindrical_grid.f90:1179-1226), 2D: PINNED to the reference's module
 * (tests/golden/mrw_dist_*.npz); reads m->r_lim. */
double oracle_distance_to_closest_wall_cyl(const oracle_model *m, int icell, double x, double y, double z);
/* initialize_cumulative_zeta (MRW.f90:16-53): zeta[n] on y_i = (i-1)/(n-1) */
void oracle_mrw_zeta_table(int n, double *zeta);
/* y with zeta(y) = xi: the inverse the stub's sample_zeta (MRW.f90:58-70) means (it interpolates the table the
 * wrong way round) */
double oracle_mrw_sample_y(const oracle_model *m, float xi);

/* Steps 1-3 of define_dark_zone + the extension of zj_sup (optical_depth.f90:1459-1500, 1579-1586): the extent of
 * the zone the diffusion approximation refills; zj_sup[n_rad]. */
int oracle_dark_zone_extent(const oracle_model *m, int lambda, double tau_max, const double *r_lim, int *ri_in,
                            int *ri_out, int *zj_sup);
/* Temp_approx_diffusion_vertical (diffusion.f90:292-374): the 1+1D diffusion fill of the dark zone; tab_lambda /
 * tab_delta_lambda in micron; Tdust[n_cells] in and out.  PARITY UNPINNED (known-answer tests). */
int oracle_temp_approx_diffusion_vertical(const oracle_model *m, const double *tab_lambda,
                                          const double *tab_delta_lambda, int ri_in, int ri_out, const int *zj_sup,
                                          float *Tdust, int *n_iter);

/* init_reemission (thermal_emission.f90:404-550), LTE tables without extra heating:
 *   kappa_abs_LTE(p_n_cells, n_lambda) in  ->  log_Qcool_minus_extra_heating(n_T, p_n_cells),
 *   kdB_dT_CDF(n_lambda, n_T, p_n_cells) out (reference layouts, column-major).  PARITY UNPINNED (module
 * thermal_emission is unbuildable here): pinned by known answers (Stefan-Boltzmann, grey dust) in
 * tests/test_init_reemission.py. */
int oracle_init_reemission(int p_n_cells, int n_T, int n_lambda, const float *tab_Temp, const double *tab_lambda,
                           const double *tab_delta_lambda, const double *kappa_abs_LTE, double *log_Qcool,
                           double *kdB_dT_CDF);
int oracle_init_reemission_ex(int p_n_cells, int n_T, int n_lambda, const float *tab_Temp, const double *tab_lambda,
                              const double *tab_delta_lambda, const double *kappa_abs_LTE, const double *dudt,
                              const double *heating_norm, double ufac_implicit, double *log_Qcool, double *kdB_dT_CDF);

/* select_scattering_grain (dust_prop.f90:1292-1336) of the model's method-1 tables: 1-based grain for the draw `rand` */
void oracle_build_ksca_CDF(const oracle_model *m, double *ksca_CDF);
int oracle_select_scattering_grain(const oracle_model *m, int lambda, int icell, float rand);

/* opacity(lambda, p_lambda = lambda) (dust_prop.f90:791-1033; LTE grains, no scattering suppression) followed by
 * calc_local_scattering_matrices (dust_prop.f90:1037-1243; scattering_method 2), for every wavelength: from the grains'
 * C_ext / C_sca / C_abs / tab_g (n_grains, n_lambda), Mueller matrices tab_s11 .. tab_s44 (0:nang, n_grains, n_lambda),
 * S_grain(n_grains) -- all default real --, n_grains(k) (double) and dust_density_o_n_grains(n_grains, p_n_cells)
 * (double) to kappa, kappa_abs_LTE (double), tab_albedo_pos, tab_g_pos (p_n_cells, n_lambda), tab_s11_pos,
 * prob_s11_pos, tab_s12_o_s11_pos .. tab_s44_o_s11_pos (0:nang, p_n_cells, n_lambda), in the reference's layouts and
 * types.  aniso_method 2 (Henyey-Greenstein): tab_g_pos and the ray tracer's tab_s11_pos, no Mueller sums.
 * PARITY UNPINNED (module dust_prop needs utils -> SPRNG / generated sources: unbuildable here): pinned by known
 * answers (one grain, identical classes, normalisations) and an independent numpy mirror in tests/test_opacity.py. */
int oracle_opacity(int n_grains, int n_lambda, int p_n_cells, int nang, int aniso_method, int lsepar_pola,
                   int grain_RE_LTE_start, int grain_RE_LTE_end, const float *C_ext, const float *C_sca,
                   const float *C_abs, const float *tab_g, const float *tab_s11, const float *tab_s12,
                   const float *tab_s22, const float *tab_s33, const float *tab_s34, const float *tab_s44,
                   const float *S_grain, const double *nbre_grains, const double *dens, double *kappa,
                   double *kappa_abs_LTE, float *tab_albedo_pos, float *tab_g_pos, float *tab_s11_pos,
                   float *prob_s11_pos, float *s12_o_s11, float *s22_o_s11, float *s33_o_s11, float *s34_o_s11,
                   float *s44_o_s11);

/* repartition_energie(lambda) (thermal_emission.f90:1771-1949), LTE grains (lRE_LTE; :1814-1831): how the energy emitted at
 * one wavelength splits between the stars, the disk's cells and the interstellar field.  In: tab_lambda(lambda) in
 * micron, E_stars(lambda), E_ISM(lambda), Tdust(n_cells) (default real), weight_proba_emission(n_cells) or NULL
 * (lweight_emission); the model's kappa_abs_LTE (per class with lvariable_dust), kappa_factor, volume, l_dark_zone.
 * Out: frac_E_stars(lambda), frac_E_disk(lambda), E_disk(lambda), prob_E_cell(0:n_cells, lambda).  Returns 1 when the
 * wavelength has no energy at all (the reference stops there, :1899-1903).  PARITY UNPINNED (module thermal_emission
 * is unbuildable here): pinned by known answers in tests/test_repartition_energie.py. */
int oracle_repartition_energie(const oracle_model *m, int lambda, double wl_um, double E_star, double E_ISM,
                               const float *Tdust, const float *weight_proba_emission, double *frac_E_stars,
                               double *frac_E_disk, double *E_disk, double *prob_E_cell);

/* Voronoi grid operators (Voronoi.f90). */
int oracle_find_voronoi_cell(const oracle_model *m, int iwall, double x, double y, double z);
void oracle_cross_voronoi_cell(const oracle_model *m, double x, double y,
                               double z, double u, double v, double w,
                               int icell, int previous_cell, double *x1,
                               double *y1, double *z1, int *next_cell,
                               double *s, double *s_contrib,
                               double *s_void_before);
void oracle_index_cell_voronoi(const oracle_model *m, double x, double y,
                               double z, int *icell);
void oracle_move_to_grid_voronoi(const oracle_model *m, double *x, double *y,
                                 double *z, double u, double v, double w,
                                 int *icell, int *lintersect);

/* Direction / sampling helpers. */
void oracle_cdapres(double cospsi, double phi, double u0, double v0, double w0,
                    double *u1, double *v1, double *w1);
void oracle_rotation(double xinit, double yinit, double zinit, double u1,
                     double v1, double w1, double *xfin, double *yfin,
                     double *zfin);
void oracle_hg(float g, float rand, int nang_scatt, int *itheta,
               double *cospsi);
void oracle_angle_diff_theta_pos(const oracle_model *m, int p_lambda,
                                 float rand, float rand2, int *itheta,
                                 double *cospsi);
void oracle_update_stokes(double S[4], double u0, double v0, double w0,
                          double u1, double v1, double w1, const double M[16]);
void oracle_select_wl_em(const oracle_model *m, float rand, int *lambda);
void oracle_intersect_stars(const oracle_model *m, double x, double y,
                            double z, double u, double v, double w,
                            int *lintersect, int *i_star, int *icell_star);

/* Temperature from absorbed energy (Temp_LTE id=0, thermal_emission.f90:649). */
void oracle_temp_lte(const oracle_model *m, double E_abs_cell, double volume,
                     int Ti_start, int *Ti, float *Temp, double *frac);
void oracle_temp_finale(const oracle_model *m, const double *E_abs,
                        float *Tdust);

/* Philox4x32-10 block: ctr[4], key[2] -> out[4]. */
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                          uint32_t out[4]);
/* The n-th uniform float of a packet's stream. */
float oracle_packet_rand(uint64_t seed, uint64_t packet, uint32_t n);

/*
 * The thermal Monte Carlo loop (mc_photon_loop with letape_th,
 * dust_transfer.f90:439-572).  Outputs are summed over threads:
 *   E_abs[n_cells]                     (xKJ_abs)
 *   sed[9][N_phi][N_thet][n_lambda]    (flat: lambda fastest)
 *   n_sent[n_lambda]                   (n_phot_envoyes)
 *   counters[ORACLE_N_COUNTERS]
 * Returns 0 on success.
 */
int oracle_run_thermal(const oracle_model *m, const oracle_opts *o,
                       const double *E_prior, double *E_abs, double *sed,
                       double *n_sent, uint64_t *counters);

#ifdef __cplusplus
}
#endif
#endif
