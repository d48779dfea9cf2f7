! ref_geom_driver.f90 -- bind(C) entry points around the REFERENCE's own
! geometry / temperature-grid / wavelength-grid routines.
!
! TEST INFRASTRUCTURE.  This file is ours; it contains no reference code.  It
! is compiled together with the reference's unmodified sources where they lie
! under /root/reference/src (mcfost_env, parameters, constants, messages,
! cylindrical_grid, spherical_grid, Temperature, wavelengths) by oracle/ref_build/Makefile
! into oracle/_ref/libmcfost_ref_geom.so.  Those modules need no library,
! generated file or header the image lacks, so no stand-in is written.  The
! rest of the reference (anything that pulls utils.f90 -> sha/os generated
! modules, or sprng_f.h) is unbuildable here and is not attempted.
!
! The driver only (a) fills the module variables the routines read, as
! read_param.f90:255-279 / grid.f90:273-357 would, and (b) forwards calls.

module ref_geom_driver
  use, intrinsic :: iso_c_binding
  use mcfost_env, only : dp
  use parameters
  use constants
  use cylindrical_grid
  use spherical_grid
  use temperature, only : tab_Temp, init_tab_Temp
  use kdtree2_module, only : kdtree2, kdtree2_result, wall_kdtree2_create, kdtree2_n_nearest, allocate_kdtree2_search, &
       deallocate_kdtree2_search, kdkind
  use wavelengths, only : lmono0, lambda_min, lambda_max, n_lambda, tab_lambda, tab_lambda_inf, &
       tab_lambda_sup, tab_delta_lambda, init_lambda
  implicit none

contains

  ! Single-zone power-law disk on a cylindrical grid (configs 1-4).
  subroutine ref_setup_grid(c_n_rad, c_nz, c_n_az, c_l3D, c_n_rad_in, rin, edge, rout, rref, sclht, &
       exp_beta, surf, ierr) bind(C, name="ref_setup_grid")
    integer(c_int), value :: c_n_rad, c_nz, c_n_az, c_l3D, c_n_rad_in
    real(c_double), value :: rin, edge, rout, rref, sclht, exp_beta, surf
    integer(c_int), intent(out) :: ierr
    call setup_grid_any(1, c_n_rad, c_nz, c_n_az, c_l3D, c_n_rad_in, rin, edge, rout, rref, sclht, exp_beta, surf, ierr)
  end subroutine ref_setup_grid

  ! The same zone on a spherical grid (grid_type = 2, grid.f90:345-357): uniform in cos(theta)
  subroutine ref_setup_grid_sph(c_n_rad, c_nz, c_n_az, c_l3D, c_n_rad_in, rin, edge, rout, rref, sclht, &
       exp_beta, surf, ierr) bind(C, name="ref_setup_grid_sph")
    integer(c_int), value :: c_n_rad, c_nz, c_n_az, c_l3D, c_n_rad_in
    real(c_double), value :: rin, edge, rout, rref, sclht, exp_beta, surf
    integer(c_int), intent(out) :: ierr
    call setup_grid_any(2, c_n_rad, c_nz, c_n_az, c_l3D, c_n_rad_in, rin, edge, rout, rref, sclht, exp_beta, surf, ierr)
  end subroutine ref_setup_grid_sph

  subroutine setup_grid_any(gtype, c_n_rad, c_nz, c_n_az, c_l3D, c_n_rad_in, rin, edge, rout, rref, sclht, &
       exp_beta, surf, ierr)
    integer, intent(in) :: gtype
    integer(c_int), intent(in) :: c_n_rad, c_nz, c_n_az, c_l3D, c_n_rad_in
    real(c_double), intent(in) :: rin, edge, rout, rref, sclht, exp_beta, surf
    integer(c_int), intent(out) :: ierr

    ierr = 0
    if (allocated(disk_zone)) then
       ierr = 1 ! one grid per process: the reference allocates once
       return
    endif

    n_rad = c_n_rad ; nz = c_nz ; n_az = c_n_az ; n_rad_in = c_n_rad_in
    l3D = (c_l3D /= 0)
    lVoronoi = .false. ; lcylindrical = (gtype == 1) ; lspherical = (gtype == 2) ; grid_type = gtype
    llinear_rgrid = .false. ; lidefix = .false. ; lsphere_model = .false. ; lmodel_1d = .false.
    lfargo3d = .false. ; lSeb_Charnoz = .false. ; lregular_theta = .false.

    ! grid.f90:276-283, 316-326
    nrz = n_rad * nz
    if (l3D) then
       n_cells = 2*nrz*n_az
       j_start = -nz
    else
       n_cells = nrz
       j_start = 1
    endif

    ! read_param.f90:255-279
    n_zones = 1 ; n_regions = 1
    allocate(disk_zone(1), regions(1))
    disk_zone(1)%geometry = 1
    disk_zone(1)%Rin = rin ; disk_zone(1)%edge = edge ; disk_zone(1)%Rout = rout ; disk_zone(1)%Rc = 0.0_dp
    disk_zone(1)%Rmax = rout
    disk_zone(1)%Rmin = rin - 5*edge
    disk_zone(1)%Rref = rref ; disk_zone(1)%sclht = sclht
    disk_zone(1)%exp_beta = exp_beta ; disk_zone(1)%surf = surf
    disk_zone(1)%moins_gamma_exp = 0.0_dp ; disk_zone(1)%vert_exponent = 2.0_dp
    disk_zone(1)%diskmass = 1.0_dp ; disk_zone(1)%gas_to_dust = 100.0_dp
    disk_zone(1)%region = 1
    regions(1)%n_zones = 1
    regions(1)%Rmin = disk_zone(1)%Rmin ; regions(1)%Rmax = disk_zone(1)%Rmax
    Rmin = disk_zone(1)%Rmin ; Rmax = disk_zone(1)%Rmax

    call build_cylindrical_cell_mapping()
    call define_cylindrical_grid()
  end subroutine setup_grid_any

  ! the arrays only the spherical grid uses (cylindrical_grid.f90:28-31)
  subroutine ref_get_grid_sph(o_tan_theta_lim, o_theta_lim, o_w_lim, o_r_lim_3) bind(C, name="ref_get_grid_sph")
    real(c_double), intent(out) :: o_tan_theta_lim(*), o_theta_lim(*), o_w_lim(*), o_r_lim_3(*)
    o_tan_theta_lim(1:nz+1) = tan_theta_lim(0:nz)
    o_theta_lim(1:nz+1) = theta_lim(0:nz)
    o_w_lim(1:nz+1) = w_lim(0:nz)
    o_r_lim_3(1:n_rad+1) = r_lim_3(0:n_rad)
  end subroutine ref_get_grid_sph

  ! cross_spherical_cell, spherical_grid.f90:182
  subroutine ref_cross_cell_sph(n, x0, y0, z0, u, v, w, cell, x1, y1, z1, next_cell, l) bind(C, name="ref_cross_cell_sph")
    integer(c_int), value :: n
    real(c_double), intent(in) :: x0(n), y0(n), z0(n), u(n), v(n), w(n)
    integer(c_int), intent(in) :: cell(n)
    real(c_double), intent(out) :: x1(n), y1(n), z1(n), l(n)
    integer(c_int), intent(out) :: next_cell(n)
    integer :: i
    real(kind=dp) :: l_contrib, l_void_before
    do i = 1, n
       call cross_spherical_cell(x0(i), y0(i), z0(i), u(i), v(i), w(i), cell(i), 0, x1(i), y1(i), z1(i), &
            next_cell(i), l(i), l_contrib, l_void_before)
    enddo
  end subroutine ref_cross_cell_sph

  ! index_cell_sph, spherical_grid.f90:48
  subroutine ref_index_cell_sph(n, x, y, z, icell) bind(C, name="ref_index_cell_sph")
    integer(c_int), value :: n
    real(c_double), intent(in) :: x(n), y(n), z(n)
    integer(c_int), intent(out) :: icell(n)
    integer :: i
    do i = 1, n
       call index_cell_sph(x(i), y(i), z(i), icell(i))
    enddo
  end subroutine ref_index_cell_sph

  ! test_exit_grid_sph, spherical_grid.f90:24
  subroutine ref_test_exit_grid_sph(n, icell, x, y, z, lexit) bind(C, name="ref_test_exit_grid_sph")
    integer(c_int), value :: n
    integer(c_int), intent(in) :: icell(n)
    real(c_double), intent(in) :: x(n), y(n), z(n)
    integer(c_int), intent(out) :: lexit(n)
    integer :: i
    do i = 1, n
       lexit(i) = merge(1, 0, test_exit_grid_sph(icell(i), x(i), y(i), z(i)))
    enddo
  end subroutine ref_test_exit_grid_sph

  ! move_to_grid_sph, spherical_grid.f90:562
  subroutine ref_move_to_grid_sph(n, x, y, z, u, v, w, icell, lintersect) bind(C, name="ref_move_to_grid_sph")
    integer(c_int), value :: n
    real(c_double), intent(inout) :: x(n), y(n), z(n)
    real(c_double), intent(in) :: u(n), v(n), w(n)
    integer(c_int), intent(out) :: icell(n), lintersect(n)
    integer :: i
    logical :: lint
    do i = 1, n
       icell(i) = 0
       call move_to_grid_sph(1, x(i), y(i), z(i), u(i), v(i), w(i), icell(i), lint)
       lintersect(i) = merge(1, 0, lint)
    enddo
  end subroutine ref_move_to_grid_sph

  ! pos_em_cell_sph, spherical_grid.f90:619
  subroutine ref_pos_em_cell_sph(n, icell, r1, r2, r3, x, y, z) bind(C, name="ref_pos_em_cell_sph")
    integer(c_int), value :: n
    integer(c_int), intent(in) :: icell(n)
    real(c_float), intent(in) :: r1(n), r2(n), r3(n)
    real(c_double), intent(out) :: x(n), y(n), z(n)
    integer :: i
    do i = 1, n
       call pos_em_cell_sph(icell(i), r1(i), r2(i), r3(i), x(i), y(i), z(i))
    enddo
  end subroutine ref_pos_em_cell_sph

  subroutine ref_grid_sizes(o_n_cells, o_ntot2, o_jlo, o_jn) bind(C, name="ref_grid_sizes")
    integer(c_int), intent(out) :: o_n_cells, o_ntot2, o_jlo, o_jn
    o_n_cells = n_cells
    o_ntot2 = size(cell_map_i)
    o_jlo = lbound(cell_map,2)
    o_jn = size(cell_map,2)
  end subroutine ref_grid_sizes

  subroutine ref_get_grid(o_r_lim, o_r_lim_2, o_zmax, o_z_lim, o_tan_phi_lim, o_volume, o_r_grid, o_z_grid, &
       o_cell_map, o_cmi, o_cmj, o_cmk, o_lexit, o_rmax2) bind(C, name="ref_get_grid")
    real(c_double), intent(out) :: o_r_lim(*), o_r_lim_2(*), o_zmax(*), o_z_lim(*), o_tan_phi_lim(*), &
         o_volume(*), o_r_grid(*), o_z_grid(*)
    integer(c_int), intent(out) :: o_cell_map(*), o_cmi(*), o_cmj(*), o_cmk(*), o_lexit(*)
    real(c_double), intent(out) :: o_rmax2
    integer :: n2

    o_r_lim(1:n_rad+1) = r_lim(0:n_rad)
    o_r_lim_2(1:n_rad+1) = r_lim_2(0:n_rad)
    o_zmax(1:n_rad) = zmax(1:n_rad)
    o_z_lim(1:n_rad*(nz+2)) = reshape(z_lim, (/ n_rad*(nz+2) /))
    o_tan_phi_lim(1:n_az) = tan_phi_lim(1:n_az)
    o_volume(1:n_cells) = volume(1:n_cells)
    o_r_grid(1:n_cells) = r_grid(1:n_cells)
    o_z_grid(1:n_cells) = z_grid(1:n_cells)
    n2 = size(cell_map_i)
    o_cell_map(1:size(cell_map)) = reshape(cell_map, (/ size(cell_map) /))
    o_cmi(1:n2) = cell_map_i ; o_cmj(1:n2) = cell_map_j ; o_cmk(1:n2) = cell_map_k
    o_lexit(1:n2) = lexit_cell
    o_rmax2 = Rmax2
  end subroutine ref_get_grid

  ! cross_cylindrical_cell, cylindrical_grid.f90:918
  subroutine ref_cross_cell(n, x0, y0, z0, u, v, w, cell, x1, y1, z1, next_cell, l) bind(C, name="ref_cross_cell")
    integer(c_int), value :: n
    real(c_double), intent(in) :: x0(n), y0(n), z0(n), u(n), v(n), w(n)
    integer(c_int), intent(in) :: cell(n)
    real(c_double), intent(out) :: x1(n), y1(n), z1(n), l(n)
    integer(c_int), intent(out) :: next_cell(n)
    integer :: i
    real(kind=dp) :: l_contrib, l_void_before
    do i = 1, n
       call cross_cylindrical_cell(x0(i), y0(i), z0(i), u(i), v(i), w(i), cell(i), 0, x1(i), y1(i), z1(i), &
            next_cell(i), l(i), l_contrib, l_void_before)
    enddo
  end subroutine ref_cross_cell

  ! index_cell_cyl, cylindrical_grid.f90:833
  subroutine ref_index_cell(n, x, y, z, icell) bind(C, name="ref_index_cell")
    integer(c_int), value :: n
    real(c_double), intent(in) :: x(n), y(n), z(n)
    integer(c_int), intent(out) :: icell(n)
    integer :: i
    do i = 1, n
       call index_cell_cyl(x(i), y(i), z(i), icell(i))
    enddo
  end subroutine ref_index_cell

  ! test_exit_grid_cyl, cylindrical_grid.f90:680
  subroutine ref_test_exit_grid(n, icell, x, y, z, lexit) bind(C, name="ref_test_exit_grid")
    integer(c_int), value :: n
    integer(c_int), intent(in) :: icell(n)
    real(c_double), intent(in) :: x(n), y(n), z(n)
    integer(c_int), intent(out) :: lexit(n)
    integer :: i
    do i = 1, n
       lexit(i) = merge(1, 0, test_exit_grid_cyl(icell(i), x(i), y(i), z(i)))
    enddo
  end subroutine ref_test_exit_grid

  ! move_to_grid_cyl, cylindrical_grid.f90:1284
  subroutine ref_move_to_grid(n, x, y, z, u, v, w, icell, lintersect) bind(C, name="ref_move_to_grid")
    integer(c_int), value :: n
    real(c_double), intent(inout) :: x(n), y(n), z(n)
    real(c_double), intent(in) :: u(n), v(n), w(n)
    integer(c_int), intent(out) :: icell(n), lintersect(n)
    integer :: i
    logical :: lint
    do i = 1, n
       icell(i) = 0
       call move_to_grid_cyl(1, x(i), y(i), z(i), u(i), v(i), w(i), icell(i), lint)
       lintersect(i) = merge(1, 0, lint)
    enddo
  end subroutine ref_move_to_grid

  ! pos_em_cell_cyl, cylindrical_grid.f90:1415
  subroutine ref_pos_em_cell(n, icell, r1, r2, r3, x, y, z) bind(C, name="ref_pos_em_cell")
    integer(c_int), value :: n
    integer(c_int), intent(in) :: icell(n)
    real(c_float), intent(in) :: r1(n), r2(n), r3(n)
    real(c_double), intent(out) :: x(n), y(n), z(n)
    integer :: i
    do i = 1, n
       call pos_em_cell_cyl(icell(i), r1(i), r2(i), r3(i), x(i), y(i), z(i))
    enddo
  end subroutine ref_pos_em_cell

  ! distance_to_closest_wall_cyl, cylindrical_grid.f90:1179
  subroutine ref_distance_to_closest_wall(n, icell, x, y, z, d) bind(C, name="ref_distance_to_closest_wall")
    integer(c_int), value :: n
    integer(c_int), intent(in) :: icell(n)
    real(c_double), intent(in) :: x(n), y(n), z(n)
    real(c_double), intent(out) :: d(n)
    integer :: i
    do i = 1, n
       d(i) = distance_to_closest_wall_cyl(icell(i), x(i), y(i), z(i))
    enddo
  end subroutine ref_distance_to_closest_wall

  ! find_Voronoi_cell (Voronoi.f90:1625-1645) on one wall's sites: the tree of build_wall_kdtrees (:1593-1621,
  ! wall_kdtree2_create with rearrange = sort = .false.) and kdtree2_n_nearest with NN = 1.  idx is 1-based into sites.
  subroutine ref_kdtree_nearest(n, sites, nq, q, idx) bind(C, name="ref_kdtree_nearest")
    integer(c_int), value :: n, nq
    real(c_double), intent(in) :: sites(3, n), q(3, nq)
    integer(c_int), intent(out) :: idx(nq)
    real(kdkind), allocatable, target :: wall_cells(:,:,:)
    real(kdkind), target :: qv(3)
    type(kdtree2), pointer :: tree
    type(kdtree2_result) :: results(1)
    integer :: i
    allocate(wall_cells(3, n, 1))
    wall_cells(:,:,1) = sites
    call deallocate_kdtree2_search()
    call allocate_kdtree2_search(1)
    tree => wall_kdtree2_create(wall_cells, 1, n_points=n, rearrange=.false., sort=.false.)
    do i = 1, nq
       qv = q(:, i)
       call kdtree2_n_nearest(1, tree, qv, 1, results)
       idx(i) = results(1)%idx
    enddo
  end subroutine ref_kdtree_nearest

  ! init_tab_Temp, Temperature.f90:23
  subroutine ref_init_tab_temp(c_n_T, c_T_min, c_T_max, o_tab_Temp) bind(C, name="ref_init_tab_temp")
    integer(c_int), value :: c_n_T
    real(c_float), value :: c_T_min, c_T_max
    real(c_float), intent(out) :: o_tab_Temp(c_n_T)
    n_T = c_n_T ; T_min = c_T_min ; T_max = c_T_max
    if (allocated(tab_Temp)) deallocate(tab_Temp)
    allocate(tab_Temp(n_T))
    call init_tab_Temp()
    o_tab_Temp(1:n_T) = tab_Temp(1:n_T)
  end subroutine ref_init_tab_temp

  ! init_lambda, wavelengths.f90:25
  subroutine ref_init_lambda(c_n_lambda, c_lambda_min, c_lambda_max, o_lambda, o_inf, o_sup, o_delta) &
       bind(C, name="ref_init_lambda")
    integer(c_int), value :: c_n_lambda
    real(c_float), value :: c_lambda_min, c_lambda_max
    real(c_double), intent(out) :: o_lambda(c_n_lambda), o_inf(c_n_lambda), o_sup(c_n_lambda), o_delta(c_n_lambda)
    if (allocated(tab_lambda)) return
    lmono0 = .false. ; n_pop = 1
    n_lambda = c_n_lambda ; lambda_min = c_lambda_min ; lambda_max = c_lambda_max
    call init_lambda()
    o_lambda = tab_lambda ; o_inf = tab_lambda_inf ; o_sup = tab_lambda_sup ; o_delta = tab_delta_lambda
  end subroutine ref_init_lambda

  ! constants the restatement must agree on (constants.f90)
  subroutine ref_constants(o) bind(C, name="ref_constants")
    real(c_double), intent(out) :: o(16)
    o(1) = pi ; o(2) = hp ; o(3) = kb ; o(4) = c_light
    o(5) = real(thermal_const, kind=dp) ; o(6) = AU_to_cm ; o(7) = Rsun_to_AU
    o(8) = real(tiny_real, kind=dp) ; o(9) = real(huge_real, kind=dp) ; o(10) = real(max_int, kind=dp)
    o(11) = grid_prec ; o(12) = AU_to_m ; o(13) = Msun_to_g ; o(14) = real(cutoff, kind=dp)
    o(15) = mum_to_cm ; o(16) = pc_to_AU
  end subroutine ref_constants

end module ref_geom_driver
