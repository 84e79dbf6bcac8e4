! A Fortran host that keeps its tracers on the GPU: the Fortran twin of resident_step.c.  Device memory comes from the HIP
! runtime's C entry points, the engine is called through the bind(C) interfaces (module obgc_c_api) wrapped by module
! obgc_device; after the set-up no tracer, tendency or forcing array crosses PCIe.
!
!   synthetic columns on the device -> nsteps x ( BGC_SurfaceFluxes -> SourceSink -> explicit update )
!
! usage: resident_step_f ncol nlev [outfile|-] [fused] [nsteps] [tiled]
!   fused = 0 : BGC_SourceSink alone, uniform surface forcing -- the sequence of resident_step.c, bit for bit
!   fused = 1 : BGC + DMS + MACROS tendencies in ONE call per step (obgc_fused_step), synthetic surface forcing
!   tiled = 1 : the level-dependent arrays are rows of one allocation in the column-block layout (OBGC_LAYOUT_TILED), what
!               the column kernel streams fastest; the results are the same bits
! prints a checksum and the wall time per step; with outfile writes the final tracers as raw doubles
! (30 x nlev x ncol, then 5 x nlev x ncol for dms, dmsp, prot, poly, lip when fused).
program resident_step_f
  use iso_c_binding
  use obgc_c_api
  use obgc_device
  implicit none
  type(obgc_device_columns) :: d
  integer :: ncol, nlev, nsteps, fused, tiled, s, u
  integer(c_long_long) :: kc, n, t0, t1, rate, i
  real(c_double), parameter :: dt = 900.0_c_double
  real(c_double), allocatable :: host(:), col(:)
  real(c_double) :: checksum, weighted, ms_first, ms_rest
  character(len=512) :: arg, fout

  ncol = 4096; nlev = 60; fout = '-'; fused = 0; nsteps = 3; tiled = 0
  if (command_argument_count() >= 1) then
    call get_command_argument(1, arg); read(arg, *) ncol
  end if
  if (command_argument_count() >= 2) then
    call get_command_argument(2, arg); read(arg, *) nlev
  end if
  if (command_argument_count() >= 3) call get_command_argument(3, fout)
  if (command_argument_count() >= 4) then
    call get_command_argument(4, arg); read(arg, *) fused
  end if
  if (command_argument_count() >= 5) then
    call get_command_argument(5, arg); read(arg, *) nsteps
  end if
  if (command_argument_count() >= 6) then
    call get_command_argument(6, arg); read(arg, *) tiled
  end if
  kc = int(ncol, c_long_long) * nlev

  call obgc_device_columns_create(d, ncol, nlev, fused /= 0, tiled /= 0)
  call obgc_device_synthetic(d, 0_c_long_long, fused /= 0)
  if (fused == 0) then
    ! uniform surface forcing, no external fluxes (the set-up of resident_step.c)
    allocate(col(ncol))
    col = 1.0_c_double;    call obgc_dev_put(d%surfacePressure, col, int(ncol, c_long_long))
    col = 0.1_c_double;    call obgc_dev_put(d%iceFraction, col, int(ncol, c_long_long))
    col = 6.4e5_c_double;  call obgc_dev_put(d%windSpeedSquared10m, col, int(ncol, c_long_long))
    col = 400.0_c_double;  call obgc_dev_put(d%atmCO2, col, int(ncol, c_long_long))
    col = 280.0_c_double;  call obgc_dev_put(d%atmCO2_ALT_CO2, col, int(ncol, c_long_long))
    col = 5.0_c_double;    call obgc_dev_put(d%surfaceDepth, col, int(ncol, c_long_long))
    col = 35.0_c_double;   call obgc_dev_put(d%SSS, col, int(ncol, c_long_long))
    call obgc_device_surface_row(d, d%SST, d%temp)                   ! SST = T(:, level 1)
    deallocate(col)
  end if
  call obgc_check(obgc_sync(obgc_engine()), 'obgc_sync')

  call system_clock(count_rate=rate)
  ms_first = 0.0_c_double; ms_rest = 0.0_c_double
  do s = 1, nsteps
    call system_clock(t0)
    call obgc_device_step(d, dt)
    call obgc_check(obgc_sync(obgc_engine()), 'obgc_sync')
    call system_clock(t1)
    if (s == 1) then
      ms_first = 1.0e3_c_double * real(t1 - t0, c_double) / real(rate, c_double)
    else
      ms_rest = ms_rest + 1.0e3_c_double * real(t1 - t0, c_double) / real(rate, c_double)
    end if
  end do
  if (nsteps > 1) ms_rest = ms_rest / real(nsteps - 1, c_double)

  n = 30 * kc
  if (fused /= 0) n = 35 * kc
  allocate(host(n))
  call obgc_device_get_field(d, host, d%tracers, 30)
  if (fused /= 0) call obgc_device_get_field(d, host(30 * kc + 1:), d%own, 5)
  checksum = 0.0_c_double; weighted = 0.0_c_double
  do i = 1, min(n, 30 * kc)
    checksum = checksum + host(i)
    weighted = weighted + host(i) * real(mod(i - 1, 1009_c_long_long), c_double)
  end do
  print '(a,i0,a,i0,a,i0,a,i0,a,i0,a,es24.17,a,es24.17)', 'resident_step_f columns=', ncol, ' levels=', nlev, ' steps=', nsteps, &
        ' fused=', fused, ' tiled=', tiled, ' checksum=', checksum, ' weighted=', weighted
  print '(a,f10.3,a,f10.3,a,i0)', 'resident_step_f wall per step: first ', ms_first, ' ms, later steps ', ms_rest, &
        ' ms; schedule of the last source-sink call: ', obgc_last_schedule(obgc_engine())
  if (trim(fout) /= '-') then
    open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
    write(u) host
    close(u)
  end if
  deallocate(host)
  call obgc_device_columns_destroy(d)
end program resident_step_f
