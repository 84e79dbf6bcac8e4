#!/usr/bin/env python3
"""Emit oracle/_ref/ref_driver.F90: a bind(C) harness around the UNMODIFIED
reference modules (compiled from /root/reference by oracle/Makefile).

ORACLE / TEST INFRASTRUCTURE ONLY.  This is harness code written for this repo:
it fills the reference's derived types (BGC_parms.F90:127-321,
DMS_parms.F90:85-154, MACROS_parms.F90:79-113) from flat C arrays, calls the
reference's public routines (BGC_mod.F90:106-110, DMS_mod.F90:51-55,
MACROS_mod.F90:51-54, co2calc.F90:24) and copies the results back.  No
reference source text is copied; the repetitive pack/unpack statements are
generated from oracle/fields.py.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fields as F  # noqa: E402


def emit_bgc():
    o = []
    w = o.append
    w("module ref_bgc_driver")
    w("  use iso_c_binding")
    w("  use BGC_parms")
    w("  use BGC_mod")
    w("  use co2calc")
    w("  implicit none")
    w("  type(BGC_indices_type), save :: g_ind")
    w("  type(autotroph_type), dimension(autotroph_cnt), save :: g_auto")
    w("  type(BGC_input_type), save :: g_in")
    w("  type(BGC_forcing_type), save :: g_frc")
    w("  type(BGC_output_type), save :: g_out")
    w("  type(BGC_diagnostics_type), save :: g_diag")
    w("  type(BGC_flux_diagnostics_type), save :: g_fdiag")
    w("  integer, save :: g_nz = -1, g_nc = -1")
    w("contains")
    # ---------------- setup
    w("  subroutine ref_bgc_setup(ind, T0K) bind(C, name='ref_bgc_setup')")
    w("    integer(c_int), intent(in) :: ind(30)")
    w("    real(c_double), value :: T0K")
    for i, n in enumerate(F.BGC_TRACERS):
        w(f"    g_ind%{n}_ind = ind({i+1})")
    w("    if (.not. allocated(g_ind%short_name)) then")
    w("      allocate(g_ind%short_name(30), g_ind%long_name(30), g_ind%units(30))")
    w("    end if")
    w("    T0_Kelvin_BGC = T0K")
    w("    call BGC_parms_init(g_ind, g_auto)")
    w("    call BGC_init(g_ind, g_auto)")
    w("  end subroutine")
    # ---------------- names
    w("  subroutine ref_bgc_get_name(i, which, buf) bind(C, name='ref_bgc_get_name')")
    w("    integer(c_int), value :: i, which")
    w("    character(kind=c_char), intent(out) :: buf(256)")
    w("    character(len=256) :: s")
    w("    integer :: j")
    w("    if (which == 0) then")
    w("      s = g_ind%short_name(i)")
    w("    else if (which == 1) then")
    w("      s = g_ind%long_name(i)")
    w("    else")
    w("      s = g_ind%units(i)")
    w("    end if")
    w("    do j = 1, 256")
    w("      buf(j) = s(j:j)")
    w("    end do")
    w("  end subroutine")
    # ---------------- params get/set
    for mode in ("get", "set"):
        w(f"  subroutine ref_bgc_{mode}_params(areal, aint, parms, sz, sv, consts) bind(C, name='ref_bgc_{mode}_params')")
        w("    real(c_double) :: areal(%d,4), parms(%d), sz(4), sv(4), consts(16)" % (len(F.AUTOTROPH_REALS), len(F.BGC_PARMS)))
        w("    integer(c_int) :: aint(%d,4)" % len(F.AUTOTROPH_INTS))
        w("    integer :: a")
        w("    do a = 1, 4")
        for i, n in enumerate(F.AUTOTROPH_REALS):
            if mode == "get":
                w(f"      areal({i+1},a) = g_auto(a)%{n}")
            else:
                w(f"      g_auto(a)%{n} = areal({i+1},a)")
        for i, n in enumerate(F.AUTOTROPH_INTS):
            logical = n in ("Nfixer", "imp_calcifier", "exp_calcifier")
            if mode == "get":
                if logical:
                    w(f"      aint({i+1},a) = merge(1, 0, g_auto(a)%{n})")
                else:
                    w(f"      aint({i+1},a) = g_auto(a)%{n}")
            else:
                if logical:
                    w(f"      g_auto(a)%{n} = (aint({i+1},a) /= 0)")
                else:
                    w(f"      g_auto(a)%{n} = aint({i+1},a)")
        w("    end do")
        for i, n in enumerate(F.BGC_PARMS):
            if mode == "get":
                w(f"    parms({i+1}) = {n}")
            else:
                w(f"    {n} = parms({i+1})")
        if mode == "get":
            w("    sz = parm_scalelen_z")
            w("    sv = parm_scalelen_vals")
            consts = ["epsC", "epsTinv", "epsnondim", "dust_to_Fe", "denitrif_C_N",
                      "dps", "yps", "DOC_reminR", "DON_reminR", "DONr_reminR",
                      "parm_Red_D_C_O2", "parm_Remin_D_C_O2", "parm_Red_D_C_O2_diaz",
                      "dust_fescav_scale", "cks", "cksi"]
            for i, n in enumerate(consts):
                w(f"    consts({i+1}) = {n}")
        else:
            w("    parm_scalelen_z = sz")
            w("    parm_scalelen_vals = sv")
        w("  end subroutine")
    # ---------------- allocation helper
    w("  subroutine bgc_alloc(nz, nc)")
    w("    integer, intent(in) :: nz, nc")
    w("    if (nz == g_nz .and. nc == g_nc) return")
    w("    if (g_nz > 0) then")
    w("      deallocate(g_in%BGC_tracers, g_in%PotentialTemperature, g_in%Salinity)")
    w("      deallocate(g_in%cell_center_depth, g_in%cell_thickness, g_in%cell_bottom_depth)")
    w("      deallocate(g_in%cell_latitude, g_in%number_of_active_levels)")
    w("      deallocate(g_frc%FESEDFLUX, g_frc%NUTR_RESTORE_RTAU, g_frc%NO3_CLIM, g_frc%PO4_CLIM, g_frc%SiO3_CLIM)")
    w("      deallocate(g_frc%dust_FLUX_IN, g_frc%ShortWaveFlux_surface)")
    for n in F.BGC_SURF_FORCING:
        w(f"      deallocate(g_frc%{n})")
    w("      deallocate(g_frc%depositionFlux, g_frc%riverFlux, g_frc%gasFlux, g_frc%seaIceFlux, g_frc%netFlux)")
    w("      deallocate(g_out%BGC_tendencies, g_out%PH_PREV_3D, g_out%PH_PREV_ALT_CO2_3D)")
    for n in F.BGC_DIAG_KC + F.BGC_DIAG_KCA + F.BGC_DIAG_CA + F.BGC_DIAG_C:
        w(f"      deallocate(g_diag%diag_{n})")
    for n in F.BGC_FLUX_DIAG:
        w(f"      deallocate(g_fdiag%{n})")
    w("    end if")
    w("    allocate(g_in%BGC_tracers(nz,nc,30), g_in%PotentialTemperature(nz,nc), g_in%Salinity(nz,nc))")
    w("    allocate(g_in%cell_center_depth(nz,nc), g_in%cell_thickness(nz,nc), g_in%cell_bottom_depth(nz,nc))")
    w("    allocate(g_in%cell_latitude(nc), g_in%number_of_active_levels(nc))")
    w("    allocate(g_frc%FESEDFLUX(nz,nc), g_frc%NUTR_RESTORE_RTAU(nz,nc), g_frc%NO3_CLIM(nz,nc))")
    w("    allocate(g_frc%PO4_CLIM(nz,nc), g_frc%SiO3_CLIM(nz,nc))")
    w("    g_frc%NUTR_RESTORE_RTAU = 0.0d0; g_frc%NO3_CLIM = 0.0d0; g_frc%PO4_CLIM = 0.0d0; g_frc%SiO3_CLIM = 0.0d0")
    w("    allocate(g_frc%dust_FLUX_IN(nc), g_frc%ShortWaveFlux_surface(nc))")
    for n in F.BGC_SURF_FORCING:
        w(f"    allocate(g_frc%{n}(nc))")
    w("    allocate(g_frc%depositionFlux(nc,30), g_frc%riverFlux(nc,30), g_frc%gasFlux(nc,30))")
    w("    allocate(g_frc%seaIceFlux(nc,30), g_frc%netFlux(nc,30))")
    w("    allocate(g_out%BGC_tendencies(nz,nc,30), g_out%PH_PREV_3D(nz,nc), g_out%PH_PREV_ALT_CO2_3D(nz,nc))")
    for n in F.BGC_DIAG_KC:
        w(f"    allocate(g_diag%diag_{n}(nz,nc))")
    for n in F.BGC_DIAG_KCA:
        w(f"    allocate(g_diag%diag_{n}(nz,nc,4))")
    for n in F.BGC_DIAG_CA:
        w(f"    allocate(g_diag%diag_{n}(nc,4))")
    for n in F.BGC_DIAG_C:
        w(f"    allocate(g_diag%diag_{n}(nc))")
    for n in F.BGC_FLUX_DIAG:
        w(f"    allocate(g_fdiag%{n}(nc))")
    w("    g_nz = nz; g_nc = nc")
    w("  end subroutine")
    # ---------------- source/sink
    nkc, nkca, nca, ncc = len(F.BGC_DIAG_KC), len(F.BGC_DIAG_KCA), len(F.BGC_DIAG_CA), len(F.BGC_DIAG_C)
    w("  subroutine ref_bgc_source_sink(nz, nc, tracers, temp, salt, zmid, dz, zbot, lat, kmax, &")
    w("       fesed, dust, sw, tend, phprev, phprev_alt, dkc, dkca, dca, dc, alt_eco, seconds) &")
    w("       bind(C, name='ref_bgc_source_sink')")
    w("    integer(c_int), value :: nz, nc, alt_eco")
    w("    real(c_double), intent(in) :: tracers(nz,nc,30), temp(nz,nc), salt(nz,nc), zmid(nz,nc), dz(nz,nc), zbot(nz,nc)")
    w("    real(c_double), intent(in) :: lat(nc), fesed(nz,nc), dust(nc), sw(nc)")
    w("    integer(c_int), intent(in) :: kmax(nc)")
    w("    real(c_double), intent(inout) :: tend(nz,nc,30), phprev(nz,nc), phprev_alt(nz,nc)")
    w(f"    real(c_double), intent(inout) :: dkc(nz,nc,{nkc}), dkca(nz,nc,4,{nkca}), dca(nc,4,{nca}), dc(nc,{ncc})")
    w("    real(c_double), intent(out) :: seconds")
    w("    integer(c_int64_t) :: t0, t1, rate")
    w("    integer :: nzl, ncl, ncmax")
    w("    logical :: leco")
    w("    call bgc_alloc(nz, nc)")
    w("    g_in%BGC_tracers = tracers")
    w("    g_in%PotentialTemperature = temp; g_in%Salinity = salt")
    w("    g_in%cell_center_depth = zmid; g_in%cell_thickness = dz; g_in%cell_bottom_depth = zbot")
    w("    g_in%cell_latitude = lat; g_in%number_of_active_levels = kmax")
    w("    g_frc%FESEDFLUX = fesed; g_frc%dust_FLUX_IN = dust; g_frc%ShortWaveFlux_surface = sw")
    w("    g_out%BGC_tendencies = tend; g_out%PH_PREV_3D = phprev; g_out%PH_PREV_ALT_CO2_3D = phprev_alt")
    for i, n in enumerate(F.BGC_DIAG_KC):
        w(f"    g_diag%diag_{n} = dkc(:,:,{i+1})")
    for i, n in enumerate(F.BGC_DIAG_KCA):
        w(f"    g_diag%diag_{n} = dkca(:,:,:,{i+1})")
    for i, n in enumerate(F.BGC_DIAG_CA):
        w(f"    g_diag%diag_{n} = dca(:,:,{i+1})")
    for i, n in enumerate(F.BGC_DIAG_C):
        w(f"    g_diag%diag_{n} = dc(:,{i+1})")
    w("    nzl = nz; ncl = nc; ncmax = nc; leco = (alt_eco /= 0)")
    w("    call system_clock(t0, rate)")
    w("    call BGC_SourceSink(g_auto, g_ind, g_in, g_frc, g_out, g_diag, nzl, ncmax, ncl, leco)")
    w("    call system_clock(t1)")
    w("    seconds = real(t1 - t0, c_double) / real(rate, c_double)")
    w("    tend = g_out%BGC_tendencies; phprev = g_out%PH_PREV_3D; phprev_alt = g_out%PH_PREV_ALT_CO2_3D")
    for i, n in enumerate(F.BGC_DIAG_KC):
        w(f"    dkc(:,:,{i+1}) = g_diag%diag_{n}")
    for i, n in enumerate(F.BGC_DIAG_KCA):
        w(f"    dkca(:,:,:,{i+1}) = g_diag%diag_{n}")
    for i, n in enumerate(F.BGC_DIAG_CA):
        w(f"    dca(:,:,{i+1}) = g_diag%diag_{n}")
    for i, n in enumerate(F.BGC_DIAG_C):
        w(f"    dc(:,{i+1}) = g_diag%diag_{n}")
    w("  end subroutine")
    # ---------------- surface fluxes
    nf, nfd = len(F.BGC_SURF_FORCING), len(F.BGC_FLUX_DIAG)
    w("  subroutine ref_bgc_surface_fluxes(nz, nc, tracers, forcing, depo, river, gas, seaice, net, fdiag, &")
    w("       lcalc_o2, lcalc_co2, seconds) bind(C, name='ref_bgc_surface_fluxes')")
    w("    integer(c_int), value :: nz, nc, lcalc_o2, lcalc_co2")
    w("    real(c_double), intent(in) :: tracers(nz,nc,30)")
    w(f"    real(c_double), intent(inout) :: forcing(nc,{nf}), depo(nc,30), river(nc,30), gas(nc,30), seaice(nc,30), net(nc,30)")
    w(f"    real(c_double), intent(inout) :: fdiag(nc,{nfd})")
    w("    real(c_double), intent(out) :: seconds")
    w("    integer(c_int64_t) :: t0, t1, rate")
    w("    integer :: ncl, ncmax")
    w("    call bgc_alloc(nz, nc)")
    w("    g_in%BGC_tracers = tracers")
    for i, n in enumerate(F.BGC_SURF_FORCING):
        w(f"    g_frc%{n} = forcing(:,{i+1})")
    w("    g_frc%depositionFlux = depo; g_frc%riverFlux = river; g_frc%gasFlux = gas")
    w("    g_frc%seaIceFlux = seaice; g_frc%netFlux = net")
    w("    g_frc%lcalc_O2_gas_flux = (lcalc_o2 /= 0); g_frc%lcalc_CO2_gas_flux = (lcalc_co2 /= 0)")
    for i, n in enumerate(F.BGC_FLUX_DIAG):
        w(f"    g_fdiag%{n} = fdiag(:,{i+1})")
    w("    ncl = nc; ncmax = nc")
    w("    call system_clock(t0, rate)")
    w("    call BGC_SurfaceFluxes(g_ind, g_in, g_frc, g_fdiag, ncmax, ncl)")
    w("    call system_clock(t1)")
    w("    seconds = real(t1 - t0, c_double) / real(rate, c_double)")
    for i, n in enumerate(F.BGC_SURF_FORCING):
        w(f"    forcing(:,{i+1}) = g_frc%{n}")
    w("    depo = g_frc%depositionFlux; river = g_frc%riverFlux; gas = g_frc%gasFlux")
    w("    seaice = g_frc%seaIceFlux; net = g_frc%netFlux")
    for i, n in enumerate(F.BGC_FLUX_DIAG):
        w(f"    fdiag(:,{i+1}) = g_fdiag%{n}")
    w("  end subroutine")
    # ---------------- co2calc unit entry points (the three public routines)
    w("  subroutine ref_comp_co3terms(n, k, depth, temp, salt, dic, ta, pt, sit, phlo, phhi, ph, h2co3, hco3, co3) &")
    w("       bind(C, name='ref_comp_co3terms')")
    w("    integer(c_int), value :: n")
    w("    integer(c_int), intent(in) :: k(n)")
    w("    real(c_double), intent(in) :: depth(n), temp(n), salt(n), dic(n), ta(n), pt(n), sit(n), phlo(n), phhi(n)")
    w("    real(c_double), intent(out) :: ph(n), h2co3(n), hco3(n), co3(n)")
    w("    integer :: i, kk")
    w("    real(c_double) :: lo, hi")
    w("    do i = 1, n")
    w("      lo = phlo(i); hi = phhi(i); kk = k(i)")
    w("      call comp_CO3terms(kk, depth(i), .true., temp(i), salt(i), dic(i), ta(i), pt(i), sit(i), &")
    w("                         lo, hi, ph(i), h2co3(i), hco3(i), co3(i))")
    w("    end do")
    w("  end subroutine")
    w("  subroutine ref_comp_co3_sat_vals(n, k, depth, temp, salt, calc, arag) bind(C, name='ref_comp_co3_sat_vals')")
    w("    integer(c_int), value :: n")
    w("    integer(c_int), intent(in) :: k(n)")
    w("    real(c_double), intent(in) :: depth(n), temp(n), salt(n)")
    w("    real(c_double), intent(out) :: calc(n), arag(n)")
    w("    integer :: i, kk")
    w("    do i = 1, n")
    w("      kk = k(i)")
    w("      call comp_co3_sat_vals(kk, depth(i), temp(i), salt(i), calc(i), arag(i))")
    w("    end do")
    w("  end subroutine")
    w("  subroutine ref_co2calc_1point(n, depth, temp, salt, dic, ta, pt, sit, phlo, phhi, xco2, atmpres, &")
    w("       ph, co2star, dco2star, pco2surf, dpco2) bind(C, name='ref_co2calc_1point')")
    w("    integer(c_int), value :: n")
    w("    real(c_double), intent(in) :: depth(n), temp(n), salt(n), dic(n), ta(n), pt(n), sit(n), phlo(n), phhi(n)")
    w("    real(c_double), intent(in) :: xco2(n), atmpres(n)")
    w("    real(c_double), intent(out) :: ph(n), co2star(n), dco2star(n), pco2surf(n), dpco2(n)")
    w("    integer :: i")
    w("    real(c_double) :: lo, hi")
    w("    do i = 1, n")
    w("      lo = phlo(i); hi = phhi(i)")
    w("      call co2calc_1point(depth(i), .true., .true., temp(i), salt(i), dic(i), ta(i), pt(i), sit(i), &")
    w("                          lo, hi, ph(i), xco2(i), atmpres(i), co2star(i), dco2star(i), pco2surf(i), dpco2(i))")
    w("    end do")
    w("  end subroutine")
    w("end module ref_bgc_driver")
    return o


def emit_dms():
    o = []
    w = o.append
    only = ", ".join(["DMS_indices_type", "DMS_input_type", "DMS_forcing_type", "DMS_output_type",
                      "DMS_flux_diagnostics_type", "DMS_diagnostics_type", "DMS_parms_init"] +
                     [f"dmsp_{n} => {n}" for n in F.DMS_PARMS])
    w("module ref_dms_driver")
    w("  use iso_c_binding")
    w(f"  use DMS_parms, only: {only}")
    w("  use DMS_mod")
    w("  implicit none")
    w("  type(DMS_indices_type), save :: d_ind")
    w("  type(DMS_input_type), save :: d_in")
    w("  type(DMS_forcing_type), save :: d_frc")
    w("  type(DMS_output_type), save :: d_out")
    w("  type(DMS_diagnostics_type), save :: d_diag")
    w("  type(DMS_flux_diagnostics_type), save :: d_fdiag")
    w("  integer, save :: d_nz = -1, d_nc = -1")
    w("contains")
    w("  subroutine ref_dms_setup(ind) bind(C, name='ref_dms_setup')")
    w("    integer(c_int), intent(in) :: ind(14)")
    for i, n in enumerate(F.DMS_TRACERS):
        w(f"    d_ind%{n}_ind = ind({i+1})")
    w("    if (.not. allocated(d_ind%short_name)) then")
    w("      allocate(d_ind%short_name(14), d_ind%long_name(14), d_ind%units(14))")
    w("    end if")
    w("    call DMS_parms_init()")
    w("    call DMS_init(d_ind)")
    w("  end subroutine")
    w("  subroutine ref_dms_get_name(i, which, buf) bind(C, name='ref_dms_get_name')")
    w("    integer(c_int), value :: i, which")
    w("    character(kind=c_char), intent(out) :: buf(256)")
    w("    character(len=256) :: s")
    w("    integer :: j")
    w("    if (which == 0) then")
    w("      s = d_ind%short_name(i)")
    w("    else if (which == 1) then")
    w("      s = d_ind%long_name(i)")
    w("    else")
    w("      s = d_ind%units(i)")
    w("    end if")
    w("    do j = 1, 256")
    w("      buf(j) = s(j:j)")
    w("    end do")
    w("  end subroutine")
    for mode in ("get", "set"):
        w(f"  subroutine ref_dms_{mode}_parms(p) bind(C, name='ref_dms_{mode}_parms')")
        w(f"    real(c_double) :: p({len(F.DMS_PARMS)})")
        for i, n in enumerate(F.DMS_PARMS):
            w(f"    p({i+1}) = dmsp_{n}" if mode == "get" else f"    dmsp_{n} = p({i+1})")
        w("  end subroutine")
    w("  subroutine dms_alloc(nz, nc)")
    w("    integer, intent(in) :: nz, nc")
    w("    if (nz == d_nz .and. nc == d_nc) return")
    w("    if (d_nz > 0) then")
    w("      deallocate(d_in%DMS_tracers, d_in%cell_thickness, d_in%number_of_active_levels)")
    w("      deallocate(d_frc%ShortWaveFlux_surface, d_frc%surfacePressure, d_frc%iceFraction)")
    w("      deallocate(d_frc%windSpeedSquared10m, d_frc%SST, d_frc%SSS, d_frc%netFlux)")
    w("      deallocate(d_out%DMS_tendencies)")
    for n in F.DMS_DIAG_KC:
        w(f"      deallocate(d_diag%diag_{n})")
    for n in F.DMS_FLUX_DIAG:
        w(f"      deallocate(d_fdiag%diag_{n})")
    w("    end if")
    w("    allocate(d_in%DMS_tracers(nz,nc,14), d_in%cell_thickness(nz,nc), d_in%number_of_active_levels(nc))")
    w("    allocate(d_frc%ShortWaveFlux_surface(nc), d_frc%surfacePressure(nc), d_frc%iceFraction(nc))")
    w("    allocate(d_frc%windSpeedSquared10m(nc), d_frc%SST(nc), d_frc%SSS(nc), d_frc%netFlux(nc,14))")
    w("    allocate(d_out%DMS_tendencies(nz,nc,14))")
    for n in F.DMS_DIAG_KC:
        w(f"    allocate(d_diag%diag_{n}(nz,nc))")
    for n in F.DMS_FLUX_DIAG:
        w(f"    allocate(d_fdiag%diag_{n}(nc))")
    w("    d_nz = nz; d_nc = nc")
    w("  end subroutine")
    nd = len(F.DMS_DIAG_KC)
    w("  subroutine ref_dms_source_sink(nz, nc, tracers, dz, kmax, sst, sw, tend, dkc, seconds) &")
    w("       bind(C, name='ref_dms_source_sink')")
    w("    integer(c_int), value :: nz, nc")
    w("    real(c_double), intent(in) :: tracers(nz,nc,14), dz(nz,nc), sst(nc), sw(nc)")
    w("    integer(c_int), intent(in) :: kmax(nc)")
    w(f"    real(c_double), intent(inout) :: tend(nz,nc,14), dkc(nz,nc,{nd})")
    w("    real(c_double), intent(out) :: seconds")
    w("    integer(c_int64_t) :: t0, t1, rate")
    w("    integer :: nzl, ncl, ncmax")
    w("    call dms_alloc(nz, nc)")
    w("    d_in%DMS_tracers = tracers; d_in%cell_thickness = dz; d_in%number_of_active_levels = kmax")
    w("    d_frc%SST = sst; d_frc%ShortWaveFlux_surface = sw")
    w("    d_out%DMS_tendencies = tend")
    for i, n in enumerate(F.DMS_DIAG_KC):
        w(f"    d_diag%diag_{n} = dkc(:,:,{i+1})")
    w("    nzl = nz; ncl = nc; ncmax = nc")
    w("    call system_clock(t0, rate)")
    w("    call DMS_SourceSink(d_ind, d_in, d_frc, d_out, d_diag, nzl, ncmax, ncl)")
    w("    call system_clock(t1)")
    w("    seconds = real(t1 - t0, c_double) / real(rate, c_double)")
    w("    tend = d_out%DMS_tendencies")
    for i, n in enumerate(F.DMS_DIAG_KC):
        w(f"    dkc(:,:,{i+1}) = d_diag%diag_{n}")
    w("  end subroutine")
    nf, nfd = len(F.DMS_SURF_FORCING), len(F.DMS_FLUX_DIAG)
    w("  subroutine ref_dms_surface_fluxes(nz, nc, tracers, forcing, net, fdiag, lcalc) &")
    w("       bind(C, name='ref_dms_surface_fluxes')")
    w("    integer(c_int), value :: nz, nc, lcalc")
    w("    real(c_double), intent(in) :: tracers(nz,nc,14)")
    w(f"    real(c_double), intent(inout) :: forcing(nc,{nf}), net(nc,14), fdiag(nc,{nfd})")
    w("    integer :: ncl, ncmax")
    w("    call dms_alloc(nz, nc)")
    w("    d_in%DMS_tracers = tracers")
    for i, n in enumerate(F.DMS_SURF_FORCING):
        w(f"    d_frc%{n} = forcing(:,{i+1})")
    w("    d_frc%netFlux = net")
    w("    d_frc%lcalc_DMS_gas_flux = (lcalc /= 0)")
    for i, n in enumerate(F.DMS_FLUX_DIAG):
        w(f"    d_fdiag%diag_{n} = fdiag(:,{i+1})")
    w("    ncl = nc; ncmax = nc")
    w("    call DMS_SurfaceFluxes(d_ind, d_in, d_frc, d_fdiag, ncmax, ncl)")
    for i, n in enumerate(F.DMS_SURF_FORCING):
        w(f"    forcing(:,{i+1}) = d_frc%{n}")
    w("    net = d_frc%netFlux")
    for i, n in enumerate(F.DMS_FLUX_DIAG):
        w(f"    fdiag(:,{i+1}) = d_fdiag%diag_{n}")
    w("  end subroutine")
    w("end module ref_dms_driver")
    return o


def emit_macros():
    o = []
    w = o.append
    only = ", ".join(["MACROS_indices_type", "MACROS_input_type", "MACROS_output_type",
                      "MACROS_diagnostics_type", "MACROS_parms_init"] +
                     [f"mp_{n} => {n}" for n in F.MACROS_PARMS])
    w("module ref_macros_driver")
    w("  use iso_c_binding")
    w(f"  use MACROS_parms, only: {only}")
    w("  use MACROS_mod")
    w("  implicit none")
    w("  type(MACROS_indices_type), save :: m_ind")
    w("  type(MACROS_input_type), save :: m_in")
    w("  type(MACROS_output_type), save :: m_out")
    w("  type(MACROS_diagnostics_type), save :: m_diag")
    w("  integer, save :: m_nz = -1, m_nc = -1")
    w("contains")
    w("  subroutine ref_macros_setup(ind) bind(C, name='ref_macros_setup')")
    w("    integer(c_int), intent(in) :: ind(8)")
    for i, n in enumerate(F.MACROS_TRACERS):
        w(f"    m_ind%{n}_ind = ind({i+1})")
    w("    if (.not. allocated(m_ind%short_name)) then")
    w("      allocate(m_ind%short_name(8), m_ind%long_name(8), m_ind%units(8))")
    w("    end if")
    w("    call MACROS_parms_init()")
    w("    call MACROS_init(m_ind)")
    w("  end subroutine")
    w("  subroutine ref_macros_get_name(i, which, buf) bind(C, name='ref_macros_get_name')")
    w("    integer(c_int), value :: i, which")
    w("    character(kind=c_char), intent(out) :: buf(256)")
    w("    character(len=256) :: s")
    w("    integer :: j")
    w("    if (which == 0) then")
    w("      s = m_ind%short_name(i)")
    w("    else if (which == 1) then")
    w("      s = m_ind%long_name(i)")
    w("    else")
    w("      s = m_ind%units(i)")
    w("    end if")
    w("    do j = 1, 256")
    w("      buf(j) = s(j:j)")
    w("    end do")
    w("  end subroutine")
    for mode in ("get", "set"):
        w(f"  subroutine ref_macros_{mode}_parms(p) bind(C, name='ref_macros_{mode}_parms')")
        w(f"    real(c_double) :: p({len(F.MACROS_PARMS)})")
        for i, n in enumerate(F.MACROS_PARMS):
            w(f"    p({i+1}) = mp_{n}" if mode == "get" else f"    mp_{n} = p({i+1})")
        w("  end subroutine")
    nd = len(F.MACROS_DIAG_KC)
    w("  subroutine ref_macros_source_sink(nz, nc, tracers, kmax, tend, dkc, seconds) &")
    w("       bind(C, name='ref_macros_source_sink')")
    w("    integer(c_int), value :: nz, nc")
    w("    real(c_double), intent(in) :: tracers(nz,nc,8)")
    w("    integer(c_int), intent(in) :: kmax(nc)")
    w(f"    real(c_double), intent(inout) :: tend(nz,nc,8), dkc(nz,nc,{nd})")
    w("    real(c_double), intent(out) :: seconds")
    w("    integer(c_int64_t) :: t0, t1, rate")
    w("    integer :: nzl, ncl, ncmax")
    w("    if (nz /= m_nz .or. nc /= m_nc) then")
    w("      if (m_nz > 0) then")
    w("        deallocate(m_in%MACROS_tracers, m_in%cell_thickness, m_in%number_of_active_levels, m_out%MACROS_tendencies)")
    for n in F.MACROS_DIAG_KC:
        w(f"        deallocate(m_diag%diag_{n})")
    w("      end if")
    w("      allocate(m_in%MACROS_tracers(nz,nc,8), m_in%cell_thickness(nz,nc), m_in%number_of_active_levels(nc))")
    w("      allocate(m_out%MACROS_tendencies(nz,nc,8))")
    for n in F.MACROS_DIAG_KC:
        w(f"      allocate(m_diag%diag_{n}(nz,nc))")
    w("      m_nz = nz; m_nc = nc")
    w("    end if")
    w("    m_in%MACROS_tracers = tracers; m_in%cell_thickness = 0.0d0; m_in%number_of_active_levels = kmax")
    w("    m_out%MACROS_tendencies = tend")
    for i, n in enumerate(F.MACROS_DIAG_KC):
        w(f"    m_diag%diag_{n} = dkc(:,:,{i+1})")
    w("    nzl = nz; ncl = nc; ncmax = nc")
    w("    call system_clock(t0, rate)")
    w("    call MACROS_SourceSink(m_ind, m_in, m_out, m_diag, nzl, ncmax, ncl)")
    w("    call system_clock(t1)")
    w("    seconds = real(t1 - t0, c_double) / real(rate, c_double)")
    w("    tend = m_out%MACROS_tendencies")
    for i, n in enumerate(F.MACROS_DIAG_KC):
        w(f"    dkc(:,:,{i+1}) = m_diag%diag_{n}")
    w("  end subroutine")
    w("end module ref_macros_driver")
    return o


def wrap(lines, width=120):
    """Fortran free-form continuation for generated long lines."""
    out = []
    for ln in lines:
        while len(ln) > width:
            cut = ln.rfind(",", 0, width - 2)
            if cut < 20:
                cut = ln.rfind(" ", 0, width - 2)
            out.append(ln[:cut + 1] + " &")
            ln = "      " + ln[cut + 1:].lstrip()
        out.append(ln)
    return out


def main():
    dst = sys.argv[1]
    text = "\n".join(wrap(emit_bgc() + [""] + emit_dms() + [""] + emit_macros())) + "\n"
    with open(dst, "w") as f:
        f.write("! GENERATED by oracle/gen_ref_driver.py -- harness, not product code\n")
        f.write(text)


if __name__ == "__main__":
    main()
