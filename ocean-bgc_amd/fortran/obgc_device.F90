! A Fortran host whose tracers LIVE on the GPU: device memory through the HIP runtime's C entry points, the engine through
! the bind(C) interfaces of obgc_c_api.  Per step nothing crosses PCIe:
!
!   obgc_device_surface_fluxes  ->  BGC_SurfaceFluxes                    (BGC_mod.F90:2706)
!   obgc_device_source_sink     ->  BGC_SourceSink                       (BGC_mod.F90:340)
!   obgc_device_fused_source_sink -> BGC_SourceSink + DMS_SourceSink + MACROS_SourceSink in one call
!                                    (BGC_mod.F90:340, DMS_mod.F90:156, MACROS_mod.F90:137)
!   obgc_device_apply           ->  the explicit update a host applies to its tracers between two calls
!
! Arrays are (column, level[, n]) in Fortran index order, i.e. the engine's native column-fastest layout -- or, with
! tiled = .true. at creation, 64-element rows of one allocation in the column-block layout (OBGC_LAYOUT_TILED); the same bits
! either way, and obgc_device_get_field returns either in the native order.  The column-block layout pays when diagnostics are
! written every step (3-5 % at a million columns); for the tendencies-only steps of this module the two tie.
module obgc_device
  use iso_c_binding
  use obgc_c_api
  implicit none
  private
  public :: obgc_device_columns, obgc_device_columns_create, obgc_device_columns_destroy, obgc_device_synthetic
  public :: obgc_device_surface_fluxes, obgc_device_source_sink, obgc_device_fused_source_sink, obgc_device_apply, obgc_device_step
  public :: obgc_dev_alloc, obgc_dev_free, obgc_dev_put, obgc_dev_get, obgc_dev_zero, obgc_dev_copy, obgc_dev_offset
  public :: obgc_device_surface_row, obgc_device_get_field

  integer(c_int), parameter :: hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3

  interface
    function hipMalloc(ptr, nbytes) bind(C, name='hipMalloc') result(rc)
      import
      type(c_ptr) :: ptr
      integer(c_size_t), value :: nbytes
      integer(c_int) :: rc
    end function
    function hipFree(ptr) bind(C, name='hipFree') result(rc)
      import
      type(c_ptr), value :: ptr
      integer(c_int) :: rc
    end function
    function hipMemcpy(dst, src, nbytes, kind) bind(C, name='hipMemcpy') result(rc)
      import
      type(c_ptr), value :: dst, src
      integer(c_size_t), value :: nbytes
      integer(c_int), value :: kind
      integer(c_int) :: rc
    end function
    function hipMemcpy2D(dst, dpitch, src, spitch, width, height, kind) bind(C, name='hipMemcpy2D') result(rc)
      import :: c_ptr, c_size_t, c_int
      type(c_ptr), value :: dst, src
      integer(c_size_t), value :: dpitch, spitch, width, height
      integer(c_int), value :: kind
      integer(c_int) :: rc
    end function
    function hipMemset(dst, val, nbytes) bind(C, name='hipMemset') result(rc)
      import
      type(c_ptr), value :: dst
      integer(c_int), value :: val
      integer(c_size_t), value :: nbytes
      integer(c_int) :: rc
    end function
  end interface

  ! one batch of columns resident on the device; every component is a DEVICE address
  ! (native layout: one allocation per array, (ncol, nlev[, n]).  Column-block layout, tiled = .true.: every level-dependent
  ! array is a set of 64-element rows of ONE allocation arena(64, rows, nlev, nblock) -- what the column kernel streams
  ! fastest -- and the components below are the addresses of their first rows; plane n of an array is pstride words further.)
  type :: obgc_device_columns
    integer :: ncol = 0, nlev = 0
    logical :: with_dms_macros = .false.
    logical :: tiled = .false.
    integer :: rows = 0                               ! column-block layout: rows per (block, level)
    integer(c_long_long) :: pstride = 0               ! words between two planes of one array
    type(c_ptr) :: arena = c_null_ptr
    type(obgc_grid) :: grid
    type(c_ptr) :: tracers = c_null_ptr, tend = c_null_ptr                      ! (ncol, nlev, 30)
    type(c_ptr) :: temp = c_null_ptr, salt = c_null_ptr, zmid = c_null_ptr, dz = c_null_ptr, zbot = c_null_ptr
    type(c_ptr) :: fesedflux = c_null_ptr, ph_prev = c_null_ptr, ph_prev_alt = c_null_ptr   ! (ncol, nlev)
    type(c_ptr) :: lat = c_null_ptr, dust_flux_in = c_null_ptr, sw_flux = c_null_ptr, kmax = c_null_ptr   ! (ncol)
    type(c_ptr) :: surfacePressure = c_null_ptr, iceFraction = c_null_ptr, windSpeedSquared10m = c_null_ptr
    type(c_ptr) :: atmCO2 = c_null_ptr, atmCO2_ALT_CO2 = c_null_ptr, surfaceDepth = c_null_ptr, SST = c_null_ptr, SSS = c_null_ptr
    type(c_ptr) :: surface_pH = c_null_ptr, surface_pH_alt_co2 = c_null_ptr
    type(c_ptr) :: depositionFlux = c_null_ptr, riverFlux = c_null_ptr, gasFlux = c_null_ptr, seaIceFlux = c_null_ptr
    type(c_ptr) :: netFlux = c_null_ptr                                         ! (ncol, 30)
    type(c_ptr) :: own = c_null_ptr, own_tend = c_null_ptr                      ! (ncol, nlev, 5): dms dmsp prot poly lip
  end type

contains

  subroutine hip_check(rc, what)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: what
    if (rc /= 0) then
      write(0, '(a,a,a,i0)') 'obgc_device: ', what, ' failed with HIP status ', rc
      error stop 1
    end if
  end subroutine

  ! n 8-byte words of device memory
  function obgc_dev_alloc(n) result(p)
    integer(c_long_long), intent(in) :: n
    type(c_ptr) :: p
    p = c_null_ptr
    call hip_check(hipMalloc(p, int(max(n, 1_c_long_long) * 8, c_size_t)), 'hipMalloc')
  end function

  subroutine obgc_dev_free(p)
    type(c_ptr), intent(inout) :: p
    if (c_associated(p)) call hip_check(hipFree(p), 'hipFree')
    p = c_null_ptr
  end subroutine

  subroutine obgc_dev_put(p, host, n)
    type(c_ptr), intent(in) :: p
    real(c_double), intent(in), target :: host(*)
    integer(c_long_long), intent(in) :: n
    call hip_check(hipMemcpy(p, c_loc(host), int(n * 8, c_size_t), hipMemcpyHostToDevice), 'hipMemcpy (to device)')
  end subroutine

  subroutine obgc_dev_get(host, p, n)
    real(c_double), intent(out), target :: host(*)
    type(c_ptr), intent(in) :: p
    integer(c_long_long), intent(in) :: n
    call hip_check(hipMemcpy(c_loc(host), p, int(n * 8, c_size_t), hipMemcpyDeviceToHost), 'hipMemcpy (to host)')
  end subroutine

  subroutine obgc_dev_copy(dst, src, n)
    type(c_ptr), intent(in) :: dst, src
    integer(c_long_long), intent(in) :: n
    call hip_check(hipMemcpy(dst, src, int(n * 8, c_size_t), hipMemcpyDeviceToDevice), 'hipMemcpy (on device)')
  end subroutine

  subroutine obgc_dev_zero(p, n)
    type(c_ptr), intent(in) :: p
    integer(c_long_long), intent(in) :: n
    call hip_check(hipMemset(p, 0_c_int, int(n * 8, c_size_t)), 'hipMemset')
  end subroutine

  ! the address `words` 8-byte words behind p
  function obgc_dev_offset(p, words) result(q)
    type(c_ptr), intent(in) :: p
    integer(c_long_long), intent(in) :: words
    type(c_ptr) :: q
    q = transfer(transfer(p, 0_c_intptr_t) + int(words * 8, c_intptr_t), c_null_ptr)
  end function

  subroutine obgc_device_columns_create(d, ncol, nlev, with_dms_macros, tiled)
    type(obgc_device_columns), intent(inout) :: d
    integer, intent(in) :: ncol, nlev
    logical, intent(in) :: with_dms_macros
    logical, intent(in), optional :: tiled
    integer(c_long_long) :: kc, c, words
    type(c_ptr) :: h
    h = obgc_engine()                     ! binds this process to its device before the first allocation
    d%ncol = ncol; d%nlev = nlev; d%with_dms_macros = with_dms_macros
    d%tiled = .false.
    if (present(tiled)) d%tiled = tiled
    d%grid%nlev = nlev; d%grid%ncol = ncol; d%grid%layout = OBGC_LAYOUT_COLFAST; d%grid%memspace = OBGC_MEM_DEVICE
    d%grid%ld = 0; d%grid%plane = 0; d%grid%ldc = 0
    kc = int(ncol, c_long_long) * nlev; c = ncol
    d%pstride = kc
    if (d%tiled) then
      ! rows: 30 tracers | T S zmid dz zbot FESEDFLUX | 2 pH | 30 tendencies | 5 own tracers, 5 own tendencies
      d%rows = 68
      if (with_dms_macros) d%rows = 78
      d%pstride = 64
      d%grid%layout = OBGC_LAYOUT_TILED; d%grid%ld = d%rows * 64; d%grid%plane = 64
      words = int((ncol + 63) / 64, c_long_long) * nlev * d%rows * 64
      d%arena = obgc_dev_alloc(words)
      call obgc_dev_zero(d%arena, words)             ! cold pH bracket, zero tendencies, defined padding columns
      d%tracers = obgc_dev_offset(d%arena, 0_c_long_long)
      d%temp = obgc_dev_offset(d%arena, 30 * 64_c_long_long); d%salt = obgc_dev_offset(d%arena, 31 * 64_c_long_long)
      d%zmid = obgc_dev_offset(d%arena, 32 * 64_c_long_long); d%dz = obgc_dev_offset(d%arena, 33 * 64_c_long_long)
      d%zbot = obgc_dev_offset(d%arena, 34 * 64_c_long_long); d%fesedflux = obgc_dev_offset(d%arena, 35 * 64_c_long_long)
      d%ph_prev = obgc_dev_offset(d%arena, 36 * 64_c_long_long); d%ph_prev_alt = obgc_dev_offset(d%arena, 37 * 64_c_long_long)
      d%tend = obgc_dev_offset(d%arena, 38 * 64_c_long_long)
      if (with_dms_macros) then
        d%own = obgc_dev_offset(d%arena, 68 * 64_c_long_long); d%own_tend = obgc_dev_offset(d%arena, 73 * 64_c_long_long)
      end if
    else
      d%tracers = obgc_dev_alloc(30 * kc); d%tend = obgc_dev_alloc(30 * kc)
      d%temp = obgc_dev_alloc(kc); d%salt = obgc_dev_alloc(kc); d%zmid = obgc_dev_alloc(kc); d%dz = obgc_dev_alloc(kc)
      d%zbot = obgc_dev_alloc(kc); d%fesedflux = obgc_dev_alloc(kc); d%ph_prev = obgc_dev_alloc(kc); d%ph_prev_alt = obgc_dev_alloc(kc)
      call obgc_dev_zero(d%ph_prev, kc); call obgc_dev_zero(d%ph_prev_alt, kc)      ! no previous solution: cold bracket
      call obgc_dev_zero(d%tend, 30 * kc)
      if (with_dms_macros) then
        d%own = obgc_dev_alloc(5 * kc); d%own_tend = obgc_dev_alloc(5 * kc)
        call obgc_dev_zero(d%own_tend, 5 * kc)
      end if
    end if
    d%lat = obgc_dev_alloc(c); d%dust_flux_in = obgc_dev_alloc(c); d%sw_flux = obgc_dev_alloc(c); d%kmax = obgc_dev_alloc(c)
    d%surfacePressure = obgc_dev_alloc(c); d%iceFraction = obgc_dev_alloc(c); d%windSpeedSquared10m = obgc_dev_alloc(c)
    d%atmCO2 = obgc_dev_alloc(c); d%atmCO2_ALT_CO2 = obgc_dev_alloc(c); d%surfaceDepth = obgc_dev_alloc(c)
    d%SST = obgc_dev_alloc(c); d%SSS = obgc_dev_alloc(c); d%surface_pH = obgc_dev_alloc(c); d%surface_pH_alt_co2 = obgc_dev_alloc(c)
    d%depositionFlux = obgc_dev_alloc(30 * c); d%riverFlux = obgc_dev_alloc(30 * c); d%gasFlux = obgc_dev_alloc(30 * c)
    d%seaIceFlux = obgc_dev_alloc(30 * c); d%netFlux = obgc_dev_alloc(30 * c)
    call obgc_dev_zero(d%surface_pH, c); call obgc_dev_zero(d%surface_pH_alt_co2, c)
    call obgc_dev_zero(d%depositionFlux, 30 * c); call obgc_dev_zero(d%riverFlux, 30 * c)
    call obgc_dev_zero(d%gasFlux, 30 * c); call obgc_dev_zero(d%seaIceFlux, 30 * c); call obgc_dev_zero(d%netFlux, 30 * c)
  end subroutine

  subroutine obgc_device_columns_destroy(d)
    type(obgc_device_columns), intent(inout) :: d
    if (d%tiled) then   ! the level-dependent components are addresses inside the arena
      call obgc_dev_free(d%arena)
      d%tracers = c_null_ptr; d%tend = c_null_ptr; d%temp = c_null_ptr; d%salt = c_null_ptr; d%zmid = c_null_ptr; d%dz = c_null_ptr
      d%zbot = c_null_ptr; d%fesedflux = c_null_ptr; d%ph_prev = c_null_ptr; d%ph_prev_alt = c_null_ptr
      d%own = c_null_ptr; d%own_tend = c_null_ptr
    end if
    call obgc_dev_free(d%tracers); call obgc_dev_free(d%tend); call obgc_dev_free(d%temp); call obgc_dev_free(d%salt)
    call obgc_dev_free(d%zmid); call obgc_dev_free(d%dz); call obgc_dev_free(d%zbot); call obgc_dev_free(d%fesedflux)
    call obgc_dev_free(d%ph_prev); call obgc_dev_free(d%ph_prev_alt); call obgc_dev_free(d%lat); call obgc_dev_free(d%dust_flux_in)
    call obgc_dev_free(d%sw_flux); call obgc_dev_free(d%kmax); call obgc_dev_free(d%surfacePressure); call obgc_dev_free(d%iceFraction)
    call obgc_dev_free(d%windSpeedSquared10m); call obgc_dev_free(d%atmCO2); call obgc_dev_free(d%atmCO2_ALT_CO2)
    call obgc_dev_free(d%surfaceDepth); call obgc_dev_free(d%SST); call obgc_dev_free(d%SSS); call obgc_dev_free(d%surface_pH)
    call obgc_dev_free(d%surface_pH_alt_co2); call obgc_dev_free(d%depositionFlux); call obgc_dev_free(d%riverFlux)
    call obgc_dev_free(d%gasFlux); call obgc_dev_free(d%seaIceFlux); call obgc_dev_free(d%netFlux)
    call obgc_dev_free(d%own); call obgc_dev_free(d%own_tend)
    d%ncol = 0; d%nlev = 0
  end subroutine

  ! the deterministic synthetic workload for global columns [col0, col0 + ncol), generated on the device:
  ! the interior state always, the surface forcing / external fluxes / own tracers when with_forcing
  subroutine obgc_device_synthetic(d, col0, with_forcing)
    type(obgc_device_columns), intent(inout) :: d
    integer(c_long_long), intent(in) :: col0
    logical, intent(in) :: with_forcing
    type(obgc_synth_args) :: a
    type(obgc_synth_extra_args) :: e
    integer(c_long_long) :: kc
    kc = int(d%ncol, c_long_long) * d%nlev
    a%nlev = d%nlev; a%ncol = d%ncol; a%col0 = col0; a%ld = d%grid%ld; a%plane = d%grid%plane; a%ldc = 0; a%ind = c_null_ptr
    a%tiled = merge(1_c_int, 0_c_int, d%tiled)
    a%tracers = d%tracers; a%temp = d%temp; a%salt = d%salt; a%zmid = d%zmid; a%dz = d%dz; a%zbot = d%zbot
    a%fesedflux = d%fesedflux; a%lat = d%lat; a%dust_flux_in = d%dust_flux_in; a%sw_flux = d%sw_flux; a%kmax = d%kmax
    call obgc_check(obgc_synth_fill(obgc_engine(), a), 'obgc_synth_fill')
    if (.not. with_forcing .and. .not. d%with_dms_macros) return
    e%nlev = d%nlev; e%ncol = d%ncol; e%col0 = col0; e%ld = d%grid%ld; e%ldc = 0
    e%tiled = merge(1_c_int, 0_c_int, d%tiled)
    e%surfacePressure = c_null_ptr; e%iceFraction = c_null_ptr; e%windSpeedSquared10m = c_null_ptr; e%atmCO2 = c_null_ptr
    e%atmCO2_ALT_CO2 = c_null_ptr; e%surfaceDepth = c_null_ptr; e%SST = c_null_ptr; e%SSS = c_null_ptr
    e%surface_pH = c_null_ptr; e%surface_pH_alt_co2 = c_null_ptr
    e%depositionFlux = c_null_ptr; e%riverFlux = c_null_ptr; e%gasFlux = c_null_ptr; e%seaIceFlux = c_null_ptr
    e%dms = c_null_ptr; e%dmsp = c_null_ptr; e%prot = c_null_ptr; e%poly = c_null_ptr; e%lip = c_null_ptr
    if (with_forcing) then
      e%surfacePressure = d%surfacePressure; e%iceFraction = d%iceFraction; e%windSpeedSquared10m = d%windSpeedSquared10m
      e%atmCO2 = d%atmCO2; e%atmCO2_ALT_CO2 = d%atmCO2_ALT_CO2; e%surfaceDepth = d%surfaceDepth; e%SST = d%SST; e%SSS = d%SSS
      e%surface_pH = d%surface_pH; e%surface_pH_alt_co2 = d%surface_pH_alt_co2
      e%depositionFlux = d%depositionFlux; e%riverFlux = d%riverFlux; e%gasFlux = d%gasFlux; e%seaIceFlux = d%seaIceFlux
    end if
    if (d%with_dms_macros) then
      e%dms = d%own; e%dmsp = obgc_dev_offset(d%own, d%pstride); e%prot = obgc_dev_offset(d%own, 2 * d%pstride)
      e%poly = obgc_dev_offset(d%own, 3 * d%pstride); e%lip = obgc_dev_offset(d%own, 4 * d%pstride)
    end if
    call obgc_check(obgc_synth_fill_extra(obgc_engine(), e), 'obgc_synth_fill_extra')
  end subroutine

  ! level 1 of a (column, level) field as a dense (ncol) device vector (e.g. SST from the temperature)
  subroutine obgc_device_surface_row(d, dst, field)
    type(obgc_device_columns), intent(in) :: d
    type(c_ptr), intent(in) :: dst, field
    integer(c_size_t) :: nfull, rest
    if (.not. d%tiled) then
      call obgc_dev_copy(dst, field, int(d%ncol, c_long_long))      ! the first row of the native layout
      return
    end if
    nfull = int(d%ncol / 64, c_size_t); rest = int(mod(d%ncol, 64), c_size_t)
    if (nfull > 0) call hip_check(hipMemcpy2D(dst, 512_c_size_t, field, int(d%grid%ld, c_size_t) * d%nlev * 8, 512_c_size_t, nfull, &
                                              hipMemcpyDeviceToDevice), 'hipMemcpy2D')
    if (rest > 0) call obgc_dev_copy(obgc_dev_offset(dst, int(nfull, c_long_long) * 64), &
                                     obgc_dev_offset(field, int(nfull, c_long_long) * d%grid%ld * d%nlev), int(rest, c_long_long))
  end subroutine

  ! nplanes planes of a level-dependent array in the native order host(ncol, nlev, nplanes), whatever the layout on the device
  subroutine obgc_device_get_field(d, host, field, nplanes)
    type(obgc_device_columns), intent(in) :: d
    real(c_double), intent(out), target :: host(*)
    type(c_ptr), intent(in) :: field
    integer, intent(in) :: nplanes
    real(c_double), allocatable, target :: tile(:)
    integer(c_long_long) :: kc, b, k, n, c0, nc, src
    integer(c_size_t) :: runs
    kc = int(d%ncol, c_long_long) * d%nlev
    if (.not. d%tiled) then
      call obgc_dev_get(host, field, nplanes * kc)
      return
    end if
    ! one block of 64 columns at a time: its nlev runs of nplanes rows
    allocate(tile(64 * nplanes * d%nlev))
    runs = int(d%nlev, c_size_t)
    do b = 0, (d%ncol + 63) / 64 - 1
      call hip_check(hipMemcpy2D(c_loc(tile), int(nplanes, c_size_t) * 512, obgc_dev_offset(field, b * d%grid%ld * d%nlev), &
                                 int(d%grid%ld, c_size_t) * 8, int(nplanes, c_size_t) * 512, runs, hipMemcpyDeviceToHost), 'hipMemcpy2D')
      c0 = b * 64; nc = min(64_c_long_long, d%ncol - c0)
      do n = 0, nplanes - 1
        do k = 0, d%nlev - 1
          src = (k * nplanes + n) * 64
          host(n * kc + k * d%ncol + c0 + 1 : n * kc + k * d%ncol + c0 + nc) = tile(src + 1 : src + nc)
        end do
      end do
    end do
    deallocate(tile)
  end subroutine

  subroutine obgc_device_surface_fluxes(d)
    type(obgc_device_columns), intent(inout) :: d
    type(obgc_bgc_surface_fluxes_args) :: a
    a%grid = d%grid; a%tracers = d%tracers; a%surfacePressure = d%surfacePressure; a%windSpeedSquared10m = d%windSpeedSquared10m
    a%atmCO2 = d%atmCO2; a%atmCO2_ALT_CO2 = d%atmCO2_ALT_CO2; a%surfaceDepth = d%surfaceDepth; a%SST = d%SST; a%SSS = d%SSS
    a%iceFraction = d%iceFraction; a%surface_pH = d%surface_pH; a%surface_pH_alt_co2 = d%surface_pH_alt_co2
    a%depositionFlux = d%depositionFlux; a%riverFlux = d%riverFlux; a%gasFlux = d%gasFlux; a%seaIceFlux = d%seaIceFlux
    a%netFlux = d%netFlux; a%diag = c_null_ptr
    a%lcalc_O2_gas_flux = 1; a%lcalc_CO2_gas_flux = 1
    call obgc_check(obgc_bgc_surface_fluxes(obgc_engine(), a), 'obgc_bgc_surface_fluxes')
  end subroutine

  subroutine fill_bgc_args(d, a)
    type(obgc_device_columns), intent(in) :: d
    type(obgc_bgc_source_sink_args), intent(out) :: a
    a%grid = d%grid; a%tracers = d%tracers; a%temp = d%temp; a%salt = d%salt; a%zmid = d%zmid; a%dz = d%dz; a%zbot = d%zbot
    a%lat = d%lat; a%kmax = d%kmax; a%fesedflux = d%fesedflux; a%dust_flux_in = d%dust_flux_in; a%sw_flux = d%sw_flux
    a%tend = d%tend; a%ph_prev = d%ph_prev; a%ph_prev_alt = d%ph_prev_alt; a%diag = c_null_ptr
    a%alt_co2_use_eco = 1
  end subroutine

  subroutine obgc_device_source_sink(d)
    type(obgc_device_columns), intent(inout) :: d
    type(obgc_bgc_source_sink_args) :: a
    call fill_bgc_args(d, a)
    call obgc_check(obgc_bgc_source_sink(obgc_engine(), a), 'obgc_bgc_source_sink')
  end subroutine

  ! BGC + DMS + MACROS tendencies in one call: DMS and MACROS read the tracers they share with BGC from d%tracers, their
  ! own tracers from d%own (dms, dmsp, prot, poly, lip), and use the surface temperature d%SST
  subroutine obgc_device_fused_source_sink(d)
    type(obgc_device_columns), intent(inout) :: d
    type(obgc_fused_step_args) :: a
    integer(c_long_long) :: kc
    if (.not. d%with_dms_macros) then
      write(0, '(a)') 'obgc_device_fused_source_sink: the batch was created without DMS / MACROS tracers'
      error stop 1
    end if
    kc = d%pstride
    call fill_bgc_args(d, a%bgc)
    a%dms = d%own; a%dmsp = obgc_dev_offset(d%own, kc); a%SST = d%SST
    a%dms_tend = d%own_tend; a%dmsp_tend = obgc_dev_offset(d%own_tend, kc); a%dms_diag = c_null_ptr
    a%prot = obgc_dev_offset(d%own, 2 * kc); a%poly = obgc_dev_offset(d%own, 3 * kc); a%lip = obgc_dev_offset(d%own, 4 * kc)
    a%prot_tend = obgc_dev_offset(d%own_tend, 2 * kc); a%poly_tend = obgc_dev_offset(d%own_tend, 3 * kc)
    a%lip_tend = obgc_dev_offset(d%own_tend, 4 * kc); a%macros_diag = c_null_ptr
    call obgc_check(obgc_fused_step(obgc_engine(), a), 'obgc_fused_step')
  end subroutine

  ! tracers <- max(0, tracers + dt * (tend + netFlux / dz(top level))) for the BGC tracers, tracers + dt * tend for the
  ! DMS / MACROS tracers of their own (no surface flux in this loop)
  subroutine obgc_device_apply(d, dt)
    type(obgc_device_columns), intent(inout) :: d
    real(c_double), intent(in) :: dt
    type(obgc_apply_tendencies_args) :: a
    a%grid = d%grid; a%ntracer = 30; a%tracers = d%tracers; a%tend = d%tend; a%net_flux = d%netFlux; a%dz = d%dz; a%kmax = d%kmax
    a%dt = dt; a%clamp_nonneg = 1
    call obgc_check(obgc_apply_tendencies(obgc_engine(), a), 'obgc_apply_tendencies')
    if (d%with_dms_macros) then
      a%ntracer = 5; a%tracers = d%own; a%tend = d%own_tend; a%net_flux = c_null_ptr
      call obgc_check(obgc_apply_tendencies(obgc_engine(), a), 'obgc_apply_tendencies (own tracers)')
    end if
  end subroutine

  ! one model step on resident tracers
  subroutine obgc_device_step(d, dt)
    type(obgc_device_columns), intent(inout) :: d
    real(c_double), intent(in) :: dt
    call obgc_device_surface_fluxes(d)
    if (d%with_dms_macros) then
      call obgc_device_fused_source_sink(d)
    else
      call obgc_device_source_sink(d)
    end if
    call obgc_device_apply(d, dt)
  end subroutine
end module obgc_device
