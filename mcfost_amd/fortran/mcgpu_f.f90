! mcgpu_f.f90 -- ISO_C_BINDING interface to the MI355X packet engine
! (libmcfost_hip.so, include/mcgpu.h) and the drop-in replacement of the body of
!
!     subroutine mc_photon_loop            (src/dust_transfer.f90:439-572)
!
! for the temperature step (letape_th = .true.).  Style follows the reference's
! own bind(C) interface to voro++ (src/Voronoi.f90:70-96): `value` scalars,
! assumed-size arrays, integer error code.
!
! The routine mcgpu_thermal_loop below takes, as explicit arguments, exactly the
! module arrays the reference's OpenMP region shares (dust_transfer.f90:486-488
! plus the tables those routines read); INTEGRATION.md shows the 30-line patch
! that passes them from dust_transfer.f90.  Nothing here depends on MCFOST
! modules, so the file compiles stand-alone (amdflang / gfortran / ifort).

module mcgpu_f

  use, intrinsic :: iso_c_binding
  implicit none
  private

  integer, parameter :: dp = selected_real_kind(p=13,r=200) ! mcfost_env.f90:23

  integer(c_int), parameter, public :: MCGPU_N_SED_TYPES = 9, MCGPU_N_COUNTERS = 10
  integer(c_int), parameter, public :: MCGPU_MULTI_SHARED_DEVICE = 1
  integer(c_int), parameter, public :: MCGPU_MULTI_FORCE_RCCL = 2   ! the RCCL all-reduce also with n_dev = 1 (a sum over one rank)

  type, bind(C), public :: mcgpu_run_opts
     integer(c_int64_t) :: seed
     integer(c_int64_t) :: first_packet
     integer(c_int64_t) :: n_packets
     real(c_double)     :: n_replicas
     integer(c_int)     :: frozen
     integer(c_int)     :: accumulate
     integer(c_int)     :: grid_blocks
     integer(c_int)     :: block_threads
  end type mcgpu_run_opts

  ! SED mode: one wavelength of run_sed_mc (include/mcgpu.h: mcgpu_mono_opts)
  type, bind(C), public :: mcgpu_mono_opts
     integer(c_int64_t) :: seed
     integer(c_int)     :: lambda
     integer(c_int)     :: p_lambda
     integer(c_int)     :: n_chunks       ! n_photons_loop
     integer(c_int)     :: first_chunk    ! 0 on one GPU
     integer(c_int64_t) :: n_photons2     ! n_photons_lambda
     real(c_double)     :: n_phot_lim     ! n_photons_lim
     integer(c_int)     :: capt_sup
     integer(c_int)     :: rt1            ! lscatt_ray_tracing1
     integer(c_int)     :: accumulate
     integer(c_int)     :: grid_blocks
     integer(c_int)     :: block_threads
  end type mcgpu_mono_opts

  ! RT1 ray-traced dust SED (include/mcgpu.h: mcgpu_rt_opts)
  type, bind(C), public :: mcgpu_rt_opts
     integer(c_int) :: lambda
     real(c_double) :: wl_um              ! tab_lambda(lambda)
     real(c_double) :: E_src              ! E_totale(lambda)
     real(c_double) :: n_sent_photons     ! sum(n_phot_envoyes(lambda,:))
     real(c_double) :: distance
     real(c_double) :: ang_disque
     integer(c_int) :: l_sym_ima
     real(c_double) :: tau_dark_zone_obs
     real(c_double) :: Rmin, Rmax
  end type mcgpu_rt_opts

  ! one wavelength of mcgpu_multi_run_sed (include/mcgpu.h: mcgpu_sed_wavelength)
  type, bind(C), public :: mcgpu_sed_wavelength
     integer(c_int) :: lambda, p_lambda
     real(c_double) :: wl_um, E_star, E_ISM
     integer(c_int64_t) :: seed
     real(c_double) :: cost
  end type mcgpu_sed_wavelength

  ! the grains' tables for mcgpu_opacity (include/mcgpu.h: mcgpu_grain_tables): c_loc of module grains' arrays
  type, bind(C), public :: mcgpu_grain_tables
     integer(c_int) :: n_grains                                  ! n_grains_tot
     integer(c_int) :: grain_RE_LTE_start, grain_RE_LTE_end
     type(c_ptr)    :: C_ext, C_sca, C_abs                       ! real (n_grains_tot, n_lambda)
     type(c_ptr)    :: tab_g
     type(c_ptr)    :: tab_s11, tab_s12, tab_s22, tab_s33, tab_s34, tab_s44   ! real (0:nang_scatt, n_grains_tot, n_lambda)
     type(c_ptr)    :: S_grain                                   ! real (n_grains_tot)
     type(c_ptr)    :: n_grains_k                                ! real(dp) n_grains(n_grains_tot)
  end type mcgpu_grain_tables

  ! copies of what mcgpu_opacity built (include/mcgpu.h: mcgpu_opacity_tables): c_loc(...) or c_null_ptr
  type, bind(C), public :: mcgpu_opacity_tables
     type(c_ptr) :: kappa, kappa_abs_LTE                         ! real(dp) (p_n_cells, n_lambda)
     type(c_ptr) :: tab_albedo_pos, tab_g_pos                    ! real (p_n_cells, n_lambda)
     type(c_ptr) :: tab_s11_pos, prob_s11_pos                    ! real (0:nang_scatt, p_n_cells, p_n_lambda_pos)
     type(c_ptr) :: tab_s12_o_s11_pos, tab_s22_o_s11_pos, tab_s33_o_s11_pos, tab_s34_o_s11_pos, tab_s44_o_s11_pos
  end type mcgpu_opacity_tables

  public :: mcgpu_create, mcgpu_destroy, mcgpu_set_grid_cyl, mcgpu_set_grid_sph, mcgpu_set_grid_voronoi, mcgpu_set_midplane_snap, mcgpu_set_option, &
       mcgpu_set_stars, mcgpu_set_opacity, mcgpu_set_scattering, mcgpu_set_thermal, mcgpu_set_sed_bins, &
       mcgpu_run_thermal, mcgpu_temp_finale, mcgpu_thermal_loop, mcgpu_error_message, mcgpu_set_rt1, &
       mcgpu_run_mono, mcgpu_fetch, mcgpu_fetch_xI, mcgpu_rt1_dust_map, mcgpu_set_xI, mcgpu_rt1_image, mcgpu_set_xI_precision, &
       mcgpu_set_E_prior, mcgpu_multi_create, mcgpu_multi_destroy, mcgpu_multi_size, mcgpu_multi_ctx, mcgpu_multi_run_thermal, mcgpu_multi_run_mono, mcgpu_multi_run_sed, mcgpu_multi_rccl_ranks, mcgpu_multi_create_ex, mcgpu_multi_reductions, &
       mcgpu_counters_to_accum, mcgpu_counters_from_accum, mcgpu_temp_approx_diffusion_vertical, mcgpu_set_mrw, mcgpu_set_mrw_exit_spectrum, mcgpu_fetch_radiation_field, &
       mcgpu_build_ksca_CDF, mcgpu_voronoi_tesselation, &
       mcgpu_set_variable_dust, mcgpu_rt1_stars_map_sed, mcgpu_define_dark_zone, mcgpu_init_reemission, mcgpu_init_reemission_ex, mcgpu_repartition_energie, mcgpu_opacity, mcgpu_set_variable_dust_s11, mcgpu_set_scattering_method1, mcgpu_set_rt2, mcgpu_fetch_I_spec, mcgpu_rt1_stars_map_image, mcgpu_set_I_spec, mcgpu_rt2_source, mcgpu_rt2_dust_map, mcgpu_rt2_image, mcgpu_tau_maps

  interface
     integer(c_int) function mcgpu_create(device, ctx) bind(C, name="mcgpu_create")
       import :: c_int, c_ptr
       integer(c_int), value :: device
       type(c_ptr), intent(out) :: ctx
     end function mcgpu_create

     integer(c_int) function mcgpu_destroy(ctx) bind(C, name="mcgpu_destroy")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function mcgpu_destroy

     type(c_ptr) function mcgpu_last_error(ctx) bind(C, name="mcgpu_last_error")
       import :: c_ptr
       type(c_ptr), value :: ctx
     end function mcgpu_last_error

     integer(c_int) function mcgpu_set_grid_cyl(ctx, n_rad, nz, n_az, l3D, r_lim_2, zmax, z_lim, tan_phi_lim, &
          zmaxmax, Rmax2, volume, cell_map, cell_map_i, cell_map_j, cell_map_k, lexit_cell) &
          bind(C, name="mcgpu_set_grid_cyl")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_rad, nz, n_az, l3D
       real(c_double), intent(in) :: r_lim_2(*), zmax(*), z_lim(*), tan_phi_lim(*), volume(*)
       real(c_double), value :: zmaxmax, Rmax2
       integer(c_int), intent(in) :: cell_map(*), cell_map_i(*), cell_map_j(*), cell_map_k(*), lexit_cell(*)
     end function mcgpu_set_grid_cyl

     ! Voronoi grid (Voronoi.f90:23-67): pass Voronoi_xyz, Voronoi(:)%xyz packed (3,n), Voronoi(:)%h,
     ! %first_neighbour, %last_neighbour, neighbours_list, the two logical flags as int8 copies,
     ! wall(:)%x1..x4 packed (4,6), PS%cutting_distance_o_h, the wall neighbour lists and volume
     integer(c_int) function mcgpu_set_grid_voronoi(ctx, n_cells, voronoi_xyz, xyz_dp, h, first_neighbour, &
          last_neighbour, neighbours_list, n_neighbours, was_cut, is_star_neighbour, walls, &
          cutting_distance_o_h, wall_first, wall_cells, volume) bind(C, name="mcgpu_set_grid_voronoi")
       import :: c_int, c_ptr, c_double, c_float, c_long_long, c_int8_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_cells
       real(c_float), intent(in) :: voronoi_xyz(*), walls(*)
       real(c_double), intent(in) :: xyz_dp(*), h(*), volume(*)
       integer(c_int), intent(in) :: first_neighbour(*), last_neighbour(*), neighbours_list(*)
       integer(c_long_long), value :: n_neighbours
       integer(c_int8_t), intent(in) :: was_cut(*), is_star_neighbour(*)
       real(c_double), value :: cutting_distance_o_h
       integer(c_int), intent(in) :: wall_first(*), wall_cells(*)
     end function mcgpu_set_grid_voronoi

     ! spherical grid (grid_type = 2): arrays of module cylindrical_grid filled by the spherical branch of
     ! define_cylindrical_grid; replaces the operators of spherical_grid.f90 (grid.f90:345-357)
     integer(c_int) function mcgpu_set_grid_sph(ctx, n_rad, nz, n_az, l3D, r_lim_2, r_lim_3, tan_theta_lim, theta_lim, &
          tan_phi_lim, Rmax2, volume, cell_map, cell_map_i, cell_map_j, cell_map_k, lexit_cell) &
          bind(C, name="mcgpu_set_grid_sph")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_rad, nz, n_az, l3D
       real(c_double), intent(in) :: r_lim_2(*), r_lim_3(*), tan_theta_lim(*), theta_lim(*), tan_phi_lim(*), volume(*)
       real(c_double), value :: Rmax2
       integer(c_int), intent(in) :: cell_map(*), cell_map_i(*), cell_map_j(*), cell_map_k(*), lexit_cell(*)
     end function mcgpu_set_grid_sph

     integer(c_int) function mcgpu_set_midplane_snap(ctx, on) bind(C, name="mcgpu_set_midplane_snap")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: on
     end function mcgpu_set_midplane_snap

     ! name: a C string (trim(name)//c_null_char): "deposit", "schedule", "speculation"
     integer(c_int) function mcgpu_set_option(ctx, name, value) bind(C, name="mcgpu_set_option")
       import :: c_int, c_ptr, c_char
       type(c_ptr), value :: ctx
       character(kind=c_char), intent(in) :: name(*)
       integer(c_int), value :: value
     end function mcgpu_set_option

     integer(c_int) function mcgpu_set_stars(ctx, n_stars, x, y, z, r, icell, out_model) bind(C, name="mcgpu_set_stars")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_stars
       real(c_double), intent(in) :: x(*), y(*), z(*), r(*)
       integer(c_int), intent(in) :: icell(*), out_model(*)
     end function mcgpu_set_stars

     integer(c_int) function mcgpu_set_opacity(ctx, n_lambda, kappa, kappa_abs_LTE, tab_albedo_pos, kappa_factor, &
          l_dark_zone) bind(C, name="mcgpu_set_opacity")
       import :: c_int, c_ptr, c_double, c_float
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_lambda
       real(c_double), intent(in) :: kappa(*), kappa_abs_LTE(*), kappa_factor(*)
       real(c_float), intent(in) :: tab_albedo_pos(*)
       type(c_ptr), value :: l_dark_zone     ! c_loc of an integer(c_int8_t) array, or c_null_ptr
     end function mcgpu_set_opacity

     integer(c_int) function mcgpu_set_scattering(ctx, nang_scatt, aniso_method, lisotropic, lsepar_pola, &
          p_lambda_fixed, prob_s11_pos, s12, s22, s33, s34, s44, tab_g_pos) bind(C, name="mcgpu_set_scattering")
       import :: c_int, c_ptr, c_float
       type(c_ptr), value :: ctx
       integer(c_int), value :: nang_scatt, aniso_method, lisotropic, lsepar_pola, p_lambda_fixed
       real(c_float), intent(in) :: prob_s11_pos(*), s12(*), s22(*), s33(*), s34(*), s44(*), tab_g_pos(*)
     end function mcgpu_set_scattering

     integer(c_int) function mcgpu_set_thermal(ctx, n_T, tab_Temp, log_Qcool, kdB_dT_CDF, spectre_emission_cumul, &
          frac_E_stars, frac_E_disk, CDF_E_star, prob_E_cell, L_packet_th, T_min) bind(C, name="mcgpu_set_thermal")
       import :: c_int, c_ptr, c_double, c_float
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_T
       real(c_float), intent(in) :: tab_Temp(*)
       real(c_double), intent(in) :: log_Qcool(*), kdB_dT_CDF(*), spectre_emission_cumul(*), frac_E_stars(*), &
            frac_E_disk(*), CDF_E_star(*)
       type(c_ptr), value :: prob_E_cell     ! c_loc(prob_E_cell) or c_null_ptr when frac_E_stars == 1
       real(c_double), value :: L_packet_th
       real(c_float), value :: T_min
     end function mcgpu_set_thermal

     integer(c_int) function mcgpu_set_sed_bins(ctx, N_thet, N_phi, l_sym_centrale, l_sym_axiale) &
          bind(C, name="mcgpu_set_sed_bins")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: N_thet, N_phi, l_sym_centrale, l_sym_axiale
     end function mcgpu_set_sed_bins

     integer(c_int) function mcgpu_run_thermal(ctx, opts, E_abs, sed, n_sent, counters, kernel_ms) &
          bind(C, name="mcgpu_run_thermal")
       import :: c_int, c_ptr, c_double, c_int64_t, mcgpu_run_opts
       type(c_ptr), value :: ctx
       type(mcgpu_run_opts), intent(in) :: opts
       real(c_double), intent(out) :: E_abs(*), sed(*), n_sent(*)
       integer(c_int64_t), intent(out) :: counters(*)
       real(c_double), intent(out) :: kernel_ms
     end function mcgpu_run_thermal

     ! prior absorbed-energy grid of the reproducible (frozen-temperature) mode
     integer(c_int) function mcgpu_set_E_prior(ctx, E_prior) bind(C, name="mcgpu_set_E_prior")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: E_prior(*)
     end function mcgpu_set_E_prior

     ! compute_stars_map for the SED (dust_transfer.f90:1604): star_flux(istar) = factor * prob_E_star(lambda,istar),
     ! stars_flux(RT_n_incl*RT_n_az) = stars_map(1,1,1) per observer
     integer(c_int) function mcgpu_rt1_stars_map_sed(ctx, opts, tab_RT_az, seed, star_flux, stars_flux) &
          bind(C, name="mcgpu_rt1_stars_map_sed")
       import :: c_int, c_ptr, c_float, c_double, c_int64_t, mcgpu_rt_opts
       type(c_ptr), value :: ctx
       type(mcgpu_rt_opts), intent(in) :: opts
       real(c_float), intent(in) :: tab_RT_az(*)
       integer(c_int64_t), value :: seed
       real(c_double), intent(in) :: star_flux(*)
       real(c_double), intent(out) :: stars_flux(*)
     end function mcgpu_rt1_stars_map_sed

     ! compute_tau_map / compute_tau_surface_map (dust_transfer.f90:2006, 2114) for every observer: c_loc(tau_map(:,:,:,:)),
     ! c_loc(tau_surface_map(:,:,:,:,:)) of dust_ray_tracing.f90:59-60 (one thread's slice) or c_null_ptr
     integer(c_int) function mcgpu_tau_maps(ctx, opts, tab_RT_az, npix_x, npix_y, map_size, zoom, tau_surface, tau_map, &
          tau_surface_map, kernel_ms) bind(C, name="mcgpu_tau_maps")
       import :: c_int, c_ptr, c_float, c_double, mcgpu_rt_opts
       type(c_ptr), value :: ctx
       type(mcgpu_rt_opts), intent(in) :: opts
       real(c_float), intent(in) :: tab_RT_az(*)
       integer(c_int), value :: npix_x, npix_y
       real(c_double), value :: map_size, zoom, tau_surface
       type(c_ptr), value :: tau_map, tau_surface_map, kernel_ms
     end function mcgpu_tau_maps

     ! lvariable_dust: p_icell(:) and the tables with the p_n_cells axis, as the modules hold them (mem.f90:213-244)
     ! (the seven scattering tables: c_loc of the arrays, or c_null_ptr for all of them)
     integer(c_int) function mcgpu_set_variable_dust(ctx, p_n_cells, p_icell, kappa, kappa_abs_LTE, tab_albedo_pos, &
          log_Qcool, kdB_dT_CDF, prob_s11_pos, tab_s12_o_s11_pos, tab_s22_o_s11_pos, tab_s33_o_s11_pos, &
          tab_s34_o_s11_pos, tab_s44_o_s11_pos, tab_g_pos) bind(C, name="mcgpu_set_variable_dust")
       import :: c_int, c_ptr, c_double, c_float
       type(c_ptr), value :: ctx
       integer(c_int), value :: p_n_cells
       integer(c_int), intent(in) :: p_icell(*)
       real(c_double), intent(in) :: kappa(*), kappa_abs_LTE(*), log_Qcool(*), kdB_dT_CDF(*)
       real(c_float), intent(in) :: tab_albedo_pos(*)
       type(c_ptr), value :: prob_s11_pos, tab_s12_o_s11_pos, tab_s22_o_s11_pos, tab_s33_o_s11_pos, &
            tab_s34_o_s11_pos, tab_s44_o_s11_pos, tab_g_pos
     end function mcgpu_set_variable_dust

     ! init_reemission on the device (thermal_emission.f90:404): rebuilds log_Qcool_minus_extra_heating and kdB_dT_CDF
     ! of the context from its kappa_abs_LTE (the module arrays passed to mcgpu_set_thermal / mcgpu_set_variable_dust
     ! may then be unfilled); outputs c_loc(log_Qcool_minus_extra_heating), c_loc(kdB_dT_CDF) or c_null_ptr
     integer(c_int) function mcgpu_init_reemission(ctx, tab_lambda, tab_delta_lambda, log_Qcool, kdB_dT_CDF) &
          bind(C, name="mcgpu_init_reemission")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: tab_lambda(*), tab_delta_lambda(*)
       type(c_ptr), value :: log_Qcool, kdB_dT_CDF
     end function mcgpu_init_reemission

     ! ... with lextra_heating (thermal_emission.f90:486-494): dudt(p_n_cells) and heating_norm = AU_to_m**2 * volume * kappa_factor;
     ! ufac_implicit > 0: ldudt_implicit
     integer(c_int) function mcgpu_init_reemission_ex(ctx, tab_lambda, tab_delta_lambda, dudt, heating_norm, ufac_implicit, &
          log_Qcool, kdB_dT_CDF) bind(C, name="mcgpu_init_reemission_ex")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: tab_lambda(*), tab_delta_lambda(*), dudt(*), heating_norm(*)
       real(c_double), value :: ufac_implicit
       type(c_ptr), value :: log_Qcool, kdB_dT_CDF
     end function mcgpu_init_reemission_ex

     ! xN_abs(:,1) and xJ_abs(:,:) summed over threads (radiation_field.f90:54-55); pass c_null_ptr for either
     integer(c_int) function mcgpu_fetch_radiation_field(ctx, xN_abs, xJ_abs) bind(C, name="mcgpu_fetch_radiation_field")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, xN_abs, xJ_abs
     end function mcgpu_fetch_radiation_field

     ! modified random walk (module MRW): zeta(:) of initialize_cumulative_zeta, the mean opacities per tab_Temp,
     ! gamma_MRW, the interaction count of dust_transfer.f90:1223, r_lim(0:n_rad); n_zeta = 0 switches it off
     ! (r_lim: cylindrical_grid's r_lim(0:n_rad); not read on a Voronoi grid -- pass any array there)
     integer(c_int) function mcgpu_set_mrw(ctx, n_zeta, zeta, chi, kappa_dep, ext, gamma, n_interactions, r_lim) &
          bind(C, name="mcgpu_set_mrw")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_zeta, n_interactions
       real(c_double), intent(in) :: zeta(*), chi(*), kappa_dep(*), ext(*), r_lim(*)
       real(c_double), value :: gamma
     end function mcgpu_set_mrw

     ! the spectrum a walk's last step leaves its sphere with: exit_cdf(n_lambda, n_T [, p_n_cells]) cumulative over the
     ! wavelengths, or c_null_ptr for the cell's emission spectrum (include/mcgpu.h)
     integer(c_int) function mcgpu_set_mrw_exit_spectrum(ctx, exit_cdf) bind(C, name="mcgpu_set_mrw_exit_spectrum")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       type(c_ptr), value :: exit_cdf
     end function mcgpu_set_mrw_exit_spectrum

     ! ksca_CDF(0:n_grains, p_n_cells, n_lambda) on the device after mcgpu_set_scattering_method1 (dust_prop.f90:976-994): the
     ! grain of a scattering is then selected like select_grainsize_high_mem does; ksca_CDF_out: c_null_ptr or the table
     integer(c_int) function mcgpu_build_ksca_CDF(ctx, build, ksca_CDF_out) bind(C, name="mcgpu_build_ksca_CDF")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: build
       type(c_ptr), value :: ksca_CDF_out
     end function mcgpu_build_ksca_CDF

     ! the tessellation voro_C returns (Voronoi.f90:70-96), built on the device: see include/mcgpu.h and INTEGRATION.md
     integer(c_int) function mcgpu_voronoi_tesselation(device, n, xyz, h, limits, threshold, n_vectors, cutting_vectors, &
          cutting_distance_o_h, n_run, cells, k, knn, knn_first, extra_plane, max_neighbours, n_neigh, neigh, volume, &
          delta_edge, was_cut, kernel_ms, volume_uncut) bind(C, name="mcgpu_voronoi_tesselation")
       import :: c_int, c_double, c_ptr, c_int8_t
       integer(c_int), value :: device, n, n_vectors, n_run, k, max_neighbours
       real(c_double), intent(in) :: xyz(3,*), h(*), limits(6), cutting_vectors(3,*)
       real(c_double), value :: threshold, cutting_distance_o_h
       type(c_ptr), value :: cells, knn_first, extra_plane, volume_uncut
       integer(c_int), intent(in) :: knn(*)
       integer(c_int), intent(out) :: n_neigh(*), neigh(max_neighbours,*)
       real(c_double), intent(out) :: volume(*), delta_edge(*), kernel_ms
       integer(c_int8_t), intent(out) :: was_cut(*)
     end function mcgpu_voronoi_tesselation

     ! define_dark_zone (optical_depth.f90:1425), 2D: module cylindrical_grid's r_lim, r_grid, z_grid, z_lim in;
     ! l_dark_zone (as integer(c_int8_t)), ri_in/out_dark_zone(1), zj_sup_dark_zone(:,1) out
     integer(c_int) function mcgpu_define_dark_zone(ctx, lambda, tau_max, r_lim, r_grid, z_grid, z_lim, l_dark_zone, &
          ri_in_dark_zone, ri_out_dark_zone, zj_sup_dark_zone) bind(C, name="mcgpu_define_dark_zone")
       import :: c_int, c_ptr, c_double, c_int8_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: lambda
       real(c_double), value :: tau_max
       real(c_double), intent(in) :: r_lim(*), r_grid(*), z_grid(*), z_lim(*)
       integer(c_int8_t), intent(out) :: l_dark_zone(*)
       integer(c_int), intent(out) :: ri_in_dark_zone, ri_out_dark_zone, zj_sup_dark_zone(*)
     end function mcgpu_define_dark_zone

     ! Temp_approx_diffusion_vertical (diffusion.f90:292): tab_lambda / tab_delta_lambda of module wavelengths,
     ! ri_in_dark_zone(1), ri_out_dark_zone(1), zj_sup_dark_zone(:,1) of module cylindrical_grid, Tdust in/out
     integer(c_int) function mcgpu_temp_approx_diffusion_vertical(ctx, tab_lambda, tab_delta_lambda, ri_in_dark_zone, &
          ri_out_dark_zone, zj_sup_dark_zone, Tdust, n_iterations) bind(C, name="mcgpu_temp_approx_diffusion_vertical")
       import :: c_int, c_ptr, c_double, c_float
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: tab_lambda(*), tab_delta_lambda(*)
       integer(c_int), value :: ri_in_dark_zone, ri_out_dark_zone
       integer(c_int), intent(in) :: zj_sup_dark_zone(*)
       real(c_float), intent(inout) :: Tdust(*)
       integer(c_int), intent(out) :: n_iterations
     end function mcgpu_temp_approx_diffusion_vertical

     ! ---- several GPUs behind this one host thread (include/mcgpu.h: mcgpu_multi_*) ---------------------------
     ! devices: c_null_ptr = devices 0..n_dev-1, or c_loc of an integer(c_int) array
     integer(c_int) function mcgpu_multi_create(n_dev, devices, multi) bind(C, name="mcgpu_multi_create")
       import :: c_int, c_ptr
       integer(c_int), value :: n_dev
       type(c_ptr), value :: devices
       type(c_ptr), intent(out) :: multi
     end function mcgpu_multi_create

     ! flags = MCGPU_MULTI_SHARED_DEVICE: all n_dev contexts on ONE device, the library's own sum in place of the
     ! RCCL all-reduce (tests and dry runs of the n_dev > 1 code on a box with one GPU)
     integer(c_int) function mcgpu_multi_create_ex(n_dev, devices, flags, multi) bind(C, name="mcgpu_multi_create_ex")
       import :: c_int, c_ptr
       integer(c_int), value :: n_dev
       type(c_ptr), value :: devices
       integer(c_int), value :: flags
       type(c_ptr), intent(out) :: multi
     end function mcgpu_multi_create_ex

     ! collectives the handle has executed so far
     integer(c_int64_t) function mcgpu_multi_reductions(multi) bind(C, name="mcgpu_multi_reductions")
       import :: c_int64_t, c_ptr
       type(c_ptr), value :: multi
     end function mcgpu_multi_reductions

     integer(c_int) function mcgpu_multi_destroy(multi) bind(C, name="mcgpu_multi_destroy")
       import :: c_int, c_ptr
       type(c_ptr), value :: multi
     end function mcgpu_multi_destroy

     integer(c_int) function mcgpu_multi_size(multi) bind(C, name="mcgpu_multi_size")
       import :: c_int, c_ptr
       type(c_ptr), value :: multi
     end function mcgpu_multi_size

     ! the context of device i (0-based): upload the model to every one of them with the mcgpu_set_* calls
     type(c_ptr) function mcgpu_multi_ctx(multi, i) bind(C, name="mcgpu_multi_ctx")
       import :: c_int, c_ptr
       type(c_ptr), value :: multi
       integer(c_int), value :: i
     end function mcgpu_multi_ctx

     ! opts%n_packets = the GLOBAL packet count; one RCCL all-reduce inside; outputs = the global sums
     integer(c_int) function mcgpu_multi_run_thermal(multi, opts, E_abs, sed, n_sent, counters, kernel_ms) &
          bind(C, name="mcgpu_multi_run_thermal")
       import :: c_int, c_ptr, c_double, c_int64_t, mcgpu_run_opts
       type(c_ptr), value :: multi
       type(mcgpu_run_opts), intent(in) :: opts
       real(c_double), intent(out) :: E_abs(*), sed(*), n_sent(*)
       integer(c_int64_t), intent(out) :: counters(*)
       real(c_double), intent(out) :: kernel_ms
     end function mcgpu_multi_run_thermal

     ! the SED loop's `call mc_photon_loop(lambda, ...)` (dust_transfer.f90:939) on every device of the handle: streams
     ! split among the devices, one all-reduce of [sed | n_sent | counters] and one of xI_scatt inside the library;
     ! results are read with mcgpu_fetch / mcgpu_fetch_xI on mcgpu_multi_ctx(multi, 0)
     integer(c_int) function mcgpu_multi_run_mono(multi, opts, frac_E_stars, frac_E_disk, prob_E_cell, n_sent_chunk, &
          kernel_ms) bind(C, name="mcgpu_multi_run_mono")
       import :: c_int, c_ptr, c_double, c_int64_t, mcgpu_mono_opts
       type(c_ptr), value :: multi
       type(mcgpu_mono_opts), intent(in) :: opts
       real(c_double), value :: frac_E_stars, frac_E_disk
       type(c_ptr), value :: prob_E_cell    ! c_loc(prob_E_cell(0,lambda)) or c_null_ptr
       integer(c_int64_t), intent(out) :: n_sent_chunk(*)
       real(c_double), intent(out) :: kernel_ms
     end function mcgpu_multi_run_mono

     ! run_sed_mc's loop over the wavelengths (dust_transfer.f90:899-1027) sharded BY WAVELENGTH: every device takes whole
     ! wavelengths (repartition_energie, the packet loop, dust_map) and only their results travel -- no xI_scatt all-reduce
     integer(c_int) function mcgpu_multi_run_sed(multi, opts, n_wl, wl, Tdust, rt, tab_RT_az, sed, n_sent, E_disk, stokes_rt, &
          counters, device_of, seconds) bind(C, name="mcgpu_multi_run_sed")
       import :: c_int, c_ptr, c_float, mcgpu_mono_opts, mcgpu_sed_wavelength
       type(c_ptr), value :: multi
       type(mcgpu_mono_opts), intent(in) :: opts
       integer(c_int), value :: n_wl
       type(mcgpu_sed_wavelength), intent(in) :: wl(*)
       real(c_float), intent(in) :: Tdust(*)
       type(c_ptr), value :: rt, tab_RT_az      ! c_loc of a mcgpu_rt_opts and of tab_RT_az(RT_n_az), or c_null_ptr: no ray tracing
       type(c_ptr), value :: sed, n_sent, E_disk, stokes_rt, counters, device_of, seconds   ! c_loc of the output arrays or c_null_ptr
     end function mcgpu_multi_run_sed

     ! ranks of the handle's RCCL communicator (ncclCommCount); 0 while none has been opened
     integer(c_int) function mcgpu_multi_rccl_ranks(multi) bind(C, name="mcgpu_multi_rccl_ranks")
       import :: c_int, c_ptr
       type(c_ptr), value :: multi
     end function mcgpu_multi_rccl_ranks

     integer(c_int) function mcgpu_counters_to_accum(ctx) bind(C, name="mcgpu_counters_to_accum")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function mcgpu_counters_to_accum

     integer(c_int) function mcgpu_counters_from_accum(ctx) bind(C, name="mcgpu_counters_from_accum")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function mcgpu_counters_from_accum

     ! tab_u_rt, tab_v_rt, tab_w_rt, n_az_rt, N_type_flux: module dust_ray_tracing; tab_s11_pos(:,1,:): grains
     integer(c_int) function mcgpu_set_rt1(ctx, RT_n_incl, RT_n_az, tab_u_rt, tab_v_rt, tab_w_rt, n_az_rt, &
          n_theta_rt, N_type_flux, lsepar_contrib, tab_s11_pos, n_lambda_pos) bind(C, name="mcgpu_set_rt1")
       import :: c_int, c_ptr, c_double, c_float
       type(c_ptr), value :: ctx
       integer(c_int), value :: RT_n_incl, RT_n_az, n_az_rt, n_theta_rt, N_type_flux, lsepar_contrib, n_lambda_pos
       real(c_double), intent(in) :: tab_u_rt(*), tab_v_rt(*), tab_w_rt(*)
       real(c_float), intent(in) :: tab_s11_pos(*)
     end function mcgpu_set_rt1

     ! replaces `call repartition_energie(lambda)` (dust_transfer.f90:924; thermal_emission.f90:1771-1949, LTE grains);
     ! the cumulative distribution also stays on the device for the mcgpu_run_mono of the same wavelength
     integer(c_int) function mcgpu_repartition_energie(ctx, lambda, wl_um, E_star, E_ISM, Tdust, weight_proba_emission, &
          frac_E_stars, frac_E_disk, E_disk, prob_E_cell) bind(C, name="mcgpu_repartition_energie")
       import :: c_int, c_ptr, c_double, c_float
       type(c_ptr), value :: ctx
       integer(c_int), value :: lambda
       real(c_double), value :: wl_um, E_star, E_ISM
       real(c_float), intent(in) :: Tdust(*)
       type(c_ptr), value :: weight_proba_emission   ! c_loc(weight_proba_emission(1)) or c_null_ptr
       real(c_double), intent(out) :: frac_E_stars, frac_E_disk, E_disk
       type(c_ptr), value :: prob_E_cell             ! c_loc(prob_E_cell(0,lambda)) or c_null_ptr
     end function mcgpu_repartition_energie

     ! replaces `call compute_stars_map(lambda, ibin, iaz, u,v,w, taille_pix, dx, dy, .true.)` of the image branch of
     ! dust_map (dust_transfer.f90:1561) for every (ibin, iaz) at once
     integer(c_int) function mcgpu_rt1_stars_map_image(ctx, opts, tab_RT_az, seed, star_flux, npix_x, npix_y, map_size, zoom, &
          n_mu, mu_limb_darkening, limb_darkening, pola_limb_darkening, stars_map, star_position) &
          bind(C, name="mcgpu_rt1_stars_map_image")
       import :: c_int, c_ptr, c_double, c_float, c_int64_t, mcgpu_rt_opts
       type(c_ptr), value :: ctx
       type(mcgpu_rt_opts), intent(in) :: opts
       real(c_float), intent(in) :: tab_RT_az(*)
       integer(c_int64_t), value :: seed
       real(c_double), intent(in) :: star_flux(*)
       integer(c_int), value :: npix_x, npix_y, n_mu
       real(c_double), value :: map_size, zoom
       type(c_ptr), value :: mu_limb_darkening, limb_darkening, pola_limb_darkening   ! c_loc(...) or c_null_ptr
       real(c_double), intent(out) :: stars_map(*)
       type(c_ptr), value :: star_position
     end function mcgpu_rt1_stars_map_image

     ! lscatt_ray_tracing2: I_spec(N_type_flux,n_theta_I,n_phi_I,n_cells) and I_spec_star(n_cells) live on the device;
     ! mcgpu_run_mono with opts%rt1 = 2 deposits (radiation_field.f90:91-129), the fetch fills the module arrays of
     ! dust_ray_tracing (slice (...,1) of the thread axis, the others zero) before calc_Isca_rt2
     integer(c_int) function mcgpu_set_rt2(ctx, n_theta_I, n_phi_I, N_type_flux, lsepar_contrib) bind(C, name="mcgpu_set_rt2")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_theta_I, n_phi_I, N_type_flux, lsepar_contrib
     end function mcgpu_set_rt2

     integer(c_int) function mcgpu_set_I_spec(ctx, I_spec, I_spec_star) bind(C, name="mcgpu_set_I_spec")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: I_spec(*), I_spec_star(*)
     end function mcgpu_set_I_spec

     ! replaces `call init_dust_source_fct2(lambda, p_lambda, ibin)` (dust_transfer.f90:1467; dust_ray_tracing.f90:717-806):
     ! eps_dust2 / eps_dust2_star of module dust_ray_tracing, then the host's own dust_map ray-traces with them
     integer(c_int) function mcgpu_rt2_source(ctx, opts, p_lambda, ibin, Tdust, r_grid, z_grid, nang_ray_tracing, &
          nang_ray_tracing_star, eps_dust2, eps_dust2_star, kernel_ms) bind(C, name="mcgpu_rt2_source")
       import :: c_int, c_ptr, c_double, c_float, mcgpu_rt_opts
       type(c_ptr), value :: ctx
       type(mcgpu_rt_opts), intent(in) :: opts
       integer(c_int), value :: p_lambda, ibin, nang_ray_tracing, nang_ray_tracing_star
       real(c_float), intent(in) :: Tdust(*)
       real(c_double), intent(in) :: r_grid(*), z_grid(*)
       real(c_float), intent(out) :: eps_dust2(*), eps_dust2_star(*)
       real(c_double), intent(out) :: kernel_ms
     end function mcgpu_rt2_source

     ! dust_map with lscatt_ray_tracing2 for the inclination of the last mcgpu_rt2_source (dust_transfer.f90:1467-1577)
     integer(c_int) function mcgpu_rt2_dust_map(ctx, opts, tab_RT_az, Tdust, stokes, kernel_ms) bind(C, name="mcgpu_rt2_dust_map")
       import :: c_int, c_ptr, c_double, c_float, mcgpu_rt_opts
       type(c_ptr), value :: ctx
       type(mcgpu_rt_opts), intent(in) :: opts
       real(c_float), intent(in) :: tab_RT_az(*), Tdust(*)
       real(c_double), intent(out) :: stokes(*), kernel_ms
     end function mcgpu_rt2_dust_map

     integer(c_int) function mcgpu_rt2_image(ctx, opts, tab_RT_az, Tdust, npix_x, npix_y, map_size, zoom, image, n_rays, &
          kernel_ms) bind(C, name="mcgpu_rt2_image")
       import :: c_int, c_ptr, c_double, c_float, c_int64_t, mcgpu_rt_opts
       type(c_ptr), value :: ctx
       type(mcgpu_rt_opts), intent(in) :: opts
       real(c_float), intent(in) :: tab_RT_az(*), Tdust(*)
       integer(c_int), value :: npix_x, npix_y
       real(c_double), value :: map_size, zoom
       real(c_double), intent(out) :: image(*), kernel_ms
       integer(c_int64_t), intent(out) :: n_rays
     end function mcgpu_rt2_image

     integer(c_int) function mcgpu_fetch_I_spec(ctx, I_spec, I_spec_f64, I_spec_star, I_spec_star_f64) &
          bind(C, name="mcgpu_fetch_I_spec")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, I_spec, I_spec_f64, I_spec_star, I_spec_star_f64   ! c_loc(...) or c_null_ptr
     end function mcgpu_fetch_I_spec

     ! lscattering_method1 (scattering.f90:39-66 chose it): the grains' tables of method 1 and the local densities
     integer(c_int) function mcgpu_set_scattering_method1(ctx, grains, prob_s11, p_n_cells, dust_density_o_n_grains) &
          bind(C, name="mcgpu_set_scattering_method1")
       import :: c_int, c_ptr, c_float, c_double, mcgpu_grain_tables
       type(c_ptr), value :: ctx
       type(mcgpu_grain_tables), intent(in) :: grains
       real(c_float), intent(in) :: prob_s11(*)                 ! (n_lambda, n_grains_tot, 0:nang_scatt)
       integer(c_int), value :: p_n_cells
       real(c_double), intent(in) :: dust_density_o_n_grains(*)
     end function mcgpu_set_scattering_method1

     integer(c_int) function mcgpu_set_variable_dust_s11(ctx, tab_s11_pos) bind(C, name="mcgpu_set_variable_dust_s11")
       import :: c_int, c_ptr, c_float
       type(c_ptr), value :: ctx
       real(c_float), intent(in) :: tab_s11_pos(*)   ! (0:nang_scatt, p_n_cells, n_lambda)
     end function mcgpu_set_variable_dust_s11

     ! replaces the loop `do lambda=1,n_lambda; call prop_grains(lambda); call opacity(lambda, p_lambda)` 's second call
     ! (dust_prop.f90:791-1243; init in dust_transfer.f90:160-200) once prop_grains has filled module grains for every
     ! wavelength; the per-class tables stay on the device
     integer(c_int) function mcgpu_opacity(ctx, grains, p_n_cells, p_icell, dust_density_o_n_grains, out) &
          bind(C, name="mcgpu_opacity")
       import :: c_int, c_ptr, c_double, mcgpu_grain_tables
       type(c_ptr), value :: ctx
       type(mcgpu_grain_tables), intent(in) :: grains
       integer(c_int), value :: p_n_cells
       integer(c_int), intent(in) :: p_icell(*)
       real(c_double), intent(in) :: dust_density_o_n_grains(*)
       type(c_ptr), value :: out             ! c_loc(a mcgpu_opacity_tables) or c_null_ptr
     end function mcgpu_opacity

     ! replaces `call mc_photon_loop(lambda, p_lambda, n_photons2, n_phot_lim, 1, .false.)` (dust_transfer.f90:939)
     integer(c_int) function mcgpu_run_mono(ctx, opts, frac_E_stars, frac_E_disk, prob_E_cell, n_sent_chunk, &
          kernel_ms) bind(C, name="mcgpu_run_mono")
       import :: c_int, c_ptr, c_double, c_int64_t, mcgpu_mono_opts
       type(c_ptr), value :: ctx
       type(mcgpu_mono_opts), intent(in) :: opts
       real(c_double), value :: frac_E_stars, frac_E_disk
       type(c_ptr), value :: prob_E_cell    ! c_loc(prob_E_cell(0,lambda)) or c_null_ptr
       integer(c_int64_t), intent(out) :: n_sent_chunk(*)
       real(c_double), intent(out) :: kernel_ms
     end function mcgpu_run_mono

     integer(c_int) function mcgpu_fetch(ctx, E_abs, sed, n_sent, counters) bind(C, name="mcgpu_fetch")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, E_abs, sed, n_sent, counters   ! c_loc(...) or c_null_ptr
     end function mcgpu_fetch

     integer(c_int) function mcgpu_fetch_xI(ctx, xI_scatt_f32, xI_scatt_f64) bind(C, name="mcgpu_fetch_xI")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, xI_scatt_f32, xI_scatt_f64       ! c_loc(xI_scatt(1,1,1,1,1,1)) or c_null_ptr
     end function mcgpu_fetch_xI

     ! 8 (default) = FP64 sums of xI_scatt on the device, 4 = default real like the reference's array
     integer(c_int) function mcgpu_set_xI_precision(ctx, bytes_per_value) bind(C, name="mcgpu_set_xI_precision")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: bytes_per_value
     end function mcgpu_set_xI_precision

     integer(c_int) function mcgpu_set_xI(ctx, xI_scatt_f64) bind(C, name="mcgpu_set_xI")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: xI_scatt_f64(*)                  ! sum(xI_scatt, dim=6)
     end function mcgpu_set_xI

     ! replaces the `call dust_map(lambda,ibin,iaz)` loop of the SED branch (dust_transfer.f90:990-1005) minus
     ! compute_stars_map: stokes(N_type_flux, RT_n_incl, RT_n_az) is added to Stokes_ray_tracing(lambda,1,1,:,:,:,1)
     integer(c_int) function mcgpu_rt1_dust_map(ctx, opts, tab_RT_az, Tdust, stokes, kernel_ms) &
          bind(C, name="mcgpu_rt1_dust_map")
       import :: c_int, c_ptr, c_double, c_float, mcgpu_rt_opts
       type(c_ptr), value :: ctx
       type(mcgpu_rt_opts), intent(in) :: opts
       real(c_float), intent(in) :: tab_RT_az(*), Tdust(*)
       real(c_double), intent(out) :: stokes(*)
       real(c_double), intent(out) :: kernel_ms
     end function mcgpu_rt1_dust_map

     ! the image branch of dust_map (dust_transfer.f90:1537-1577) for all observers:
     ! image(npix_x, npix_y, RT_n_incl, RT_n_az, N_type_flux) -> Stokes_ray_tracing(lambda,:,:,:,:,:,1)
     integer(c_int) function mcgpu_rt1_image(ctx, opts, tab_RT_az, Tdust, npix_x, npix_y, map_size, zoom, image, &
          n_rays, kernel_ms) bind(C, name="mcgpu_rt1_image")
       import :: c_int, c_ptr, c_double, c_float, c_int64_t, mcgpu_rt_opts
       type(c_ptr), value :: ctx
       type(mcgpu_rt_opts), intent(in) :: opts
       real(c_float), intent(in) :: tab_RT_az(*), Tdust(*)
       integer(c_int), value :: npix_x, npix_y
       real(c_double), value :: map_size, zoom
       real(c_double), intent(out) :: image(*)
       integer(c_int64_t), intent(out) :: n_rays
       real(c_double), intent(out) :: kernel_ms
     end function mcgpu_rt1_image

     integer(c_int) function mcgpu_temp_finale(ctx, E_abs, Tdust) bind(C, name="mcgpu_temp_finale")
       import :: c_int, c_ptr, c_double, c_float
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: E_abs(*)
       real(c_float), intent(out) :: Tdust(*)
     end function mcgpu_temp_finale
  end interface

contains

  ! The body of mc_photon_loop for the thermal step, on the GPU.
  !
  ! Inputs are the reference's module arrays, by the names they have there:
  !   cylindrical_grid : r_lim_2, zmax, z_lim, tan_phi_lim, volume, cell_map, cell_map_i/j/k, lexit_cell,
  !                      l_dark_zone, (zmaxmax = maxval(zmax))
  !   parameters       : n_rad, nz, n_az, l3D, Rmax2, n_stars, star(:)%x,y,z,r,icell,out_model, n_T, T_min,
  !                      N_thet, N_phi, l_sym_centrale, l_sym_axiale, nang_scatt, aniso_method, lisotropic,
  !                      lsepar_pola
  !   dust_prop/grains : kappa(1,:), kappa_abs_LTE(1,:), tab_albedo_pos(1,:), kappa_factor, prob_s11_pos(:,1,:),
  !                      tab_s12/22/33/34/44_o_s11_pos(:,1,:), tab_g_pos(1,:)
  !   thermal_emission : tab_Temp, log_Qcool_minus_extra_heating(:,1), kdB_dT_CDF(:,:,1),
  !                      spectre_emission_cumul(0:), frac_E_stars, frac_E_disk, prob_E_cell, L_packet_th
  !   stars            : CDF_E_star(:,0:)
  ! Outputs go where the reference's reductions expect them:
  !   xKJ_abs(:,1) = E_abs ; xKJ_abs(:,2:) = 0   -> Temp_finale (thermal_emission.f90:668) unchanged
  !   sed, sed_q, ... (:,:,:,1) from sed(:,:,:,1:9) ; n_phot_envoyes(:,1) = n_sent
  subroutine mcgpu_thermal_loop(n_packets, seed, &
       n_rad, nz, n_az, l3D, r_lim_2, zmax, z_lim, tan_phi_lim, Rmax2, volume, cell_map, cell_map_i, cell_map_j, &
       cell_map_k, lexit_cell, &
       n_stars, star_x, star_y, star_z, star_r, star_icell, star_out_model, &
       n_lambda, kappa, kappa_abs_LTE, tab_albedo_pos, kappa_factor, &
       nang_scatt, aniso_method, lisotropic, lsepar_pola, prob_s11_pos, s12, s22, s33, s34, s44, tab_g_pos, &
       n_T, tab_Temp, log_Qcool, kdB_dT_CDF, spectre_emission_cumul, frac_E_stars, frac_E_disk, CDF_E_star, &
       L_packet_th, T_min, N_thet, N_phi, l_sym_centrale, l_sym_axiale, &
       E_abs, sed, n_sent, kernel_ms, ierr, n_dev, E_prior)

    integer(c_int64_t), intent(in) :: n_packets, seed
    integer, intent(in) :: n_rad, nz, n_az, n_stars, n_lambda, nang_scatt, aniso_method, n_T, N_thet, N_phi
    logical, intent(in) :: l3D, lisotropic, lsepar_pola, l_sym_centrale, l_sym_axiale
    real(dp), intent(in) :: r_lim_2(*), zmax(*), z_lim(*), tan_phi_lim(*), volume(*), Rmax2
    integer, intent(in) :: cell_map(*), cell_map_i(*), cell_map_j(*), cell_map_k(*), lexit_cell(*)
    real(dp), intent(in) :: star_x(*), star_y(*), star_z(*), star_r(*)
    integer, intent(in) :: star_icell(*), star_out_model(*)
    real(dp), intent(in) :: kappa(*), kappa_abs_LTE(*), kappa_factor(*)
    real, intent(in) :: tab_albedo_pos(*), prob_s11_pos(*), s12(*), s22(*), s33(*), s34(*), s44(*), tab_g_pos(*)
    real, intent(in) :: tab_Temp(*), T_min
    real(dp), intent(in) :: log_Qcool(*), kdB_dT_CDF(*), spectre_emission_cumul(*), frac_E_stars(*), &
         frac_E_disk(*), CDF_E_star(*), L_packet_th
    real(dp), intent(out) :: E_abs(*), sed(*), n_sent(*), kernel_ms
    integer, intent(out) :: ierr
    integer, intent(in), optional :: n_dev          ! GPUs of this node to use (default 1)
    real(dp), intent(in), optional :: E_prior(*)    ! present: reproducible mode, Temp_LTE reads this prior

    type(c_ptr) :: multi, ctx
    type(mcgpu_run_opts) :: opts
    integer(c_int64_t) :: counters(MCGPU_N_COUNTERS)
    integer(c_int) :: rc, nd, i

    ierr = 0
    nd = 1
    if (present(n_dev)) nd = n_dev
    rc = mcgpu_multi_create(nd, c_null_ptr, multi)
    if (rc /= 0) then
       ierr = rc ; return
    endif
    ! the tables are replicated: the same upload on every device
    do i = 0, nd-1
       ctx = mcgpu_multi_ctx(multi, i)
       if (rc == 0) rc = mcgpu_set_grid_cyl(ctx, n_rad, nz, n_az, merge(1,0,l3D), r_lim_2, zmax, z_lim, tan_phi_lim, &
            maxval(zmax(1:n_rad)), Rmax2, volume, cell_map, cell_map_i, cell_map_j, cell_map_k, lexit_cell)
       if (rc == 0) rc = mcgpu_set_stars(ctx, n_stars, star_x, star_y, star_z, star_r, star_icell, star_out_model)
       if (rc == 0) rc = mcgpu_set_opacity(ctx, n_lambda, kappa, kappa_abs_LTE, tab_albedo_pos, kappa_factor, c_null_ptr)
       if (rc == 0) rc = mcgpu_set_scattering(ctx, nang_scatt, aniso_method, merge(1,0,lisotropic), &
            merge(1,0,lsepar_pola), 1_c_int, prob_s11_pos, s12, s22, s33, s34, s44, tab_g_pos)
       if (rc == 0) rc = mcgpu_set_thermal(ctx, n_T, tab_Temp, log_Qcool, kdB_dT_CDF, spectre_emission_cumul, &
            frac_E_stars, frac_E_disk, CDF_E_star, c_null_ptr, L_packet_th, T_min)
       if (rc == 0) rc = mcgpu_set_sed_bins(ctx, N_thet, N_phi, merge(1,0,l_sym_centrale), merge(1,0,l_sym_axiale))
       if (rc == 0 .and. present(E_prior)) rc = mcgpu_set_E_prior(ctx, E_prior)
       if (rc /= 0) then
          write(*,*) "mcgpu error ", rc, " on device ", i, ": ", trim(mcgpu_error_message(ctx))
          exit
       endif
    enddo
    if (rc == 0) then
       opts%seed = seed ; opts%first_packet = 0 ; opts%n_packets = n_packets ; opts%n_replicas = 1.0_c_double
       opts%frozen = merge(1, 0, present(E_prior)) ; opts%accumulate = 0 ; opts%grid_blocks = 0 ; opts%block_threads = 0
       rc = mcgpu_multi_run_thermal(multi, opts, E_abs, sed, n_sent, counters, kernel_ms)
       if (rc /= 0) write(*,*) "mcgpu_multi_run_thermal failed: ", rc
    endif
    ierr = rc
    rc = mcgpu_multi_destroy(multi)

  end subroutine mcgpu_thermal_loop

  function mcgpu_error_message(ctx) result(msg)
    type(c_ptr), intent(in) :: ctx
    character(len=256) :: msg
    type(c_ptr) :: p
    character(kind=c_char), pointer :: s(:)
    integer :: i
    msg = ""
    p = mcgpu_last_error(ctx)
    if (.not. c_associated(p)) return
    call c_f_pointer(p, s, (/ 256 /))
    do i = 1, 256
       if (s(i) == c_null_char) exit
       msg(i:i) = s(i)
    enddo
  end function mcgpu_error_message

end module mcgpu_f
