! thermal_host_example.f90 -- a minimal Fortran host for the MI355X packet engine.
!
! Stands in for MCFOST's run_thermal_mc (src/dust_transfer.f90:576-650) where the
! full Fortran host cannot be built: it reads the model tables from a stream file
! written by mcfost_amd/host/dump.py (the same arrays init_dust_transfer prepares),
! calls the drop-in replacement of mc_photon_loop through the ISO_C_BINDING shim
! (mcgpu_f.f90) and writes E_abs (xKJ_abs), the SED arrays and n_phot_envoyes back.
!
!   thermal_host_example model.bin result.bin n_packets seed [n_dev [prior.bin]]
! n_dev: GPUs of this node (one host thread, RCCL all-reduce inside the library); prior.bin: n_cells doubles,
! the absorbed-energy prior of the reproducible (frozen-temperature) mode.

program thermal_host_example
  use, intrinsic :: iso_c_binding
  use mcgpu_f
  implicit none
  integer, parameter :: dp = selected_real_kind(p=13,r=200)
  character(len=512) :: fin, fout, arg
  integer(c_int64_t) :: n_packets, seed
  integer :: n_rad, nz, n_az, il3D, n_cells, ntot2, ncm, n_stars, n_lambda, nang, aniso, iiso, ipola, n_T, N_thet, N_phi, isc, isa
  real(dp) :: Rmax2, L_packet_th, kernel_ms
  real :: T_min
  real(dp), allocatable :: r_lim_2(:), zmax(:), z_lim(:), tan_phi_lim(:), volume(:), sx(:), sy(:), sz(:), sr(:), &
       kappa(:), kabs(:), kfac(:), lq(:), cdf(:), cum(:), fst(:), fdi(:), cdfs(:), E_abs(:), sed(:), n_sent(:)
  real, allocatable :: albedo(:), prob(:), s12(:), s22(:), s33(:), s34(:), s44(:), g(:), tab_Temp(:)
  integer, allocatable :: cell_map(:), cmi(:), cmj(:), cmk(:), lexit(:), sic(:), som(:)
  integer :: u, ierr, n_dev
  real(dp), allocatable :: E_prior(:)
  character(len=512) :: fprior

  call get_command_argument(1, fin) ; call get_command_argument(2, fout)
  call get_command_argument(3, arg) ; read(arg,*) n_packets
  call get_command_argument(4, arg) ; read(arg,*) seed
  n_dev = 1 ; fprior = ""
  if (command_argument_count() >= 5) then
     call get_command_argument(5, arg) ; read(arg,*) n_dev
  endif
  if (command_argument_count() >= 6) call get_command_argument(6, fprior)

  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) n_rad, nz, n_az, il3D, n_cells, ntot2, ncm, n_stars, n_lambda, nang, aniso, iiso, ipola, n_T, N_thet, N_phi, isc, isa
  read(u) Rmax2, L_packet_th
  read(u) T_min
  allocate(r_lim_2(n_rad+1), zmax(n_rad), z_lim(n_rad*(nz+2)), tan_phi_lim(n_az), volume(n_cells))
  allocate(cell_map(ncm), cmi(ntot2), cmj(ntot2), cmk(ntot2), lexit(ntot2))
  allocate(sx(n_stars), sy(n_stars), sz(n_stars), sr(n_stars), sic(n_stars), som(n_stars))
  allocate(kappa(n_lambda), kabs(n_lambda), albedo(n_lambda), kfac(n_cells))
  allocate(prob((nang+1)*n_lambda), s12((nang+1)*n_lambda), s22((nang+1)*n_lambda), s33((nang+1)*n_lambda), &
       s34((nang+1)*n_lambda), s44((nang+1)*n_lambda), g(n_lambda))
  allocate(tab_Temp(n_T), lq(n_T), cdf(n_lambda*n_T), cum(n_lambda+1), fst(n_lambda), fdi(n_lambda), &
       cdfs(n_lambda*(n_stars+1)))
  read(u) r_lim_2, zmax, z_lim, tan_phi_lim, volume
  read(u) cell_map, cmi, cmj, cmk, lexit
  read(u) sx, sy, sz, sr, sic, som
  read(u) kappa, kabs, albedo, kfac
  read(u) prob, s12, s22, s33, s34, s44, g
  read(u) tab_Temp, lq, cdf, cum, fst, fdi, cdfs
  close(u)

  allocate(E_abs(n_cells), sed(9*n_lambda*N_thet*N_phi), n_sent(n_lambda))
  if (len_trim(fprior) > 0) then
     allocate(E_prior(n_cells))
     open(newunit=u, file=trim(fprior), access='stream', form='unformatted', status='old')
     read(u) E_prior
     close(u)
     call mcgpu_thermal_loop(n_packets, seed, n_rad, nz, n_az, il3D /= 0, r_lim_2, zmax, z_lim, tan_phi_lim, Rmax2, &
          volume, cell_map, cmi, cmj, cmk, lexit, n_stars, sx, sy, sz, sr, sic, som, n_lambda, kappa, kabs, albedo, &
          kfac, nang, aniso, iiso /= 0, ipola /= 0, prob, s12, s22, s33, s34, s44, g, n_T, tab_Temp, lq, cdf, cum, &
          fst, fdi, cdfs, L_packet_th, T_min, N_thet, N_phi, isc /= 0, isa /= 0, E_abs, sed, n_sent, kernel_ms, ierr, &
          n_dev=n_dev, E_prior=E_prior)
  else
     call mcgpu_thermal_loop(n_packets, seed, n_rad, nz, n_az, il3D /= 0, r_lim_2, zmax, z_lim, tan_phi_lim, Rmax2, &
          volume, cell_map, cmi, cmj, cmk, lexit, n_stars, sx, sy, sz, sr, sic, som, n_lambda, kappa, kabs, albedo, &
          kfac, nang, aniso, iiso /= 0, ipola /= 0, prob, s12, s22, s33, s34, s44, g, n_T, tab_Temp, lq, cdf, cum, &
          fst, fdi, cdfs, L_packet_th, T_min, N_thet, N_phi, isc /= 0, isa /= 0, E_abs, sed, n_sent, kernel_ms, ierr, &
          n_dev=n_dev)
  endif
  if (ierr /= 0) then
     write(*,*) "mc_photon_loop on the GPU failed, ierr =", ierr
     call exit(1)
  endif
  write(*,'(a,i0,a,f10.3,a,es10.3,a)') " Fortran host: ", n_packets, " packets, kernel ", kernel_ms, " ms, ", &
       real(n_packets,dp)/(kernel_ms*1.0e-3_dp), " packets/s"

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) E_abs, sed, n_sent
  close(u)
end program thermal_host_example
