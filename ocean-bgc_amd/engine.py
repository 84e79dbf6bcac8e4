"""Thin Python host over the C ABI (include/obgc.h).  Arrays may be numpy arrays (host
memory) or torch tensors (device memory when `.is_cuda`); nothing is computed here."""
import ctypes as C

import numpy as np

from . import _lib as L


class ObgcError(RuntimeError):
    pass


def _ptr(x):
    """-> (address, is_device)"""
    if x is None:
        return None, None
    if isinstance(x, np.ndarray):
        if not x.flags["C_CONTIGUOUS"] and not x.flags["F_CONTIGUOUS"]:
            raise ObgcError("array must be contiguous")
        return x.ctypes.data, False
    if hasattr(x, "data_ptr"):      # torch tensor
        return x.data_ptr(), bool(x.is_cuda)
    raise ObgcError("unsupported array type %r" % type(x))


def make_grid(nlev, ncol, layout, memspace, ld=0, plane=0, ldc=0):
    g = L.Grid()
    g.nlev, g.ncol, g.layout, g.memspace, g.ld, g.plane, g.ldc = nlev, ncol, layout, memspace, ld, plane, ldc
    return g


class TileArena:
    """ONE device allocation for every level-dependent array of a call in the column-block layout (OBGC_LAYOUT_TILED):
    arena[nblock][nlev][nrows][64], block = 64 adjacent columns.  An array of n planes owns n consecutive rows; the C ABI
    gets the address of its first row, grid.ld = nrows * 64 and grid.plane = 64."""

    def __init__(self, torch, device, ncol, nlev, nrows):
        self.torch, self.ncol, self.nlev, self.nrows = torch, int(ncol), int(nlev), int(nrows)
        self.nblk = (self.ncol + 63) // 64
        self.buf = torch.zeros((self.nblk, self.nlev, self.nrows, 64), dtype=torch.float64, device=device)
        # the zero fill runs on torch's stream and the engine's streams do not order against it (include/obgc.h, "Stream
        # ordering"): it must have finished before a library call writes into the arena
        torch.cuda.synchronize(device)
        self.used = 0

    @property
    def ld(self):
        return self.nrows * 64

    def take(self, nplanes, shape=None):
        if self.used + nplanes > self.nrows:
            raise ObgcError("TileArena: %d rows are not enough" % self.nrows)
        f = TileField(self, self.used, nplanes, shape)
        self.used += nplanes
        return f


class TileField:
    """n consecutive rows of a TileArena: what the C ABI sees as one (k,column[,n]) array."""
    is_cuda = True

    def __init__(self, arena, row, nplanes, shape):
        self.arena, self.row, self.nplanes = arena, row, nplanes
        self.shape = shape if shape is not None else (nplanes, arena.ncol, arena.nlev)

    def data_ptr(self):
        return self.arena.buf.data_ptr() + self.row * 64 * 8

    def view(self):
        return self.arena.buf[:, :, self.row:self.row + self.nplanes, :]          # [nblk, nlev, n, 64]

    def sub(self, j, n=1):
        """planes [j, j + n) of this array as an array of their own"""
        return TileField(self.arena, self.row + j, n, (self.arena.ncol, self.arena.nlev) if n == 1 else None)

    def zero_(self):
        self.view().zero_()
        return self

    def surface_row(self, j=0):
        """level 1 of plane j as a dense (ncol) device vector"""
        return self.arena.buf[:, 0, self.row + j, :].reshape(-1)[:self.arena.ncol].contiguous()

    def put(self, x):
        """x: numpy [.., ncol, nlev] in the reference's order (leading extent = planes, may be absent)."""
        A, torch = self.arena, self.arena.torch
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.nplanes, A.ncol, A.nlev)
        pad = np.zeros((self.nplanes, A.nblk * 64, A.nlev))
        pad[:, :A.ncol, :] = x
        t = torch.from_numpy(pad.reshape(self.nplanes, A.nblk, 64, A.nlev)).to(A.buf.device)
        self.view().copy_(t.permute(1, 3, 0, 2))
        return self

    def get(self):
        A = self.arena
        t = self.view().permute(2, 0, 3, 1).contiguous().cpu().numpy()              # [n, nblk, 64, nlev]
        return t.reshape(self.nplanes, A.nblk * 64, A.nlev)[:, :A.ncol, :].reshape(self.shape).copy()


def partition_columns(kmax, nparts, balance="cells"):
    """Contiguous column ranges for `nparts` devices (obgc_partition_columns; host-only, no GPU needed).
    balance: "columns" (equal counts) or "cells" (equal sums of active levels).  -> list of nparts+1 bounds."""
    kmax = np.ascontiguousarray(kmax, dtype=np.int32)
    bounds = (C.c_longlong * (nparts + 1))()
    rc = L.load().obgc_partition_columns(kmax.ctypes.data_as(C.POINTER(C.c_int)), kmax.size, nparts,
                                         {"columns": 0, "cells": 1}[balance], bounds)
    if rc != 0:
        raise ObgcError("obgc_partition_columns: bad argument (rc=%d)" % rc)
    return list(bounds)


class Engine:
    """One obgc_handle = one GPU + its streams and scratch (one per host thread per GPU)."""

    def __init__(self, device=0, flags=0):
        self.lib = L.load()
        h = C.c_void_p()
        rc = self.lib.obgc_create(C.byref(h), int(device), int(flags))
        if rc != 0:
            raise ObgcError("obgc_create(device=%d) failed: %s -- a gfx950 GPU is required, there is no CPU fallback"
                            % (device, L.STATUS.get(rc, rc)))
        self.h = h
        self.bgc = L.BgcParams()
        self.dms = L.DmsParams()
        self.macros = L.MacrosParams()
        self.lib.obgc_bgc_params_init(C.byref(self.bgc), None, 273.15)
        self.lib.obgc_dms_params_init(C.byref(self.dms), None)
        self.lib.obgc_macros_params_init(C.byref(self.macros), None)

    def close(self):
        if getattr(self, "h", None):
            self.lib.obgc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise ObgcError("%s: %s (%s)" % (what, L.STATUS.get(rc, rc), self.lib.obgc_last_error(self.h).decode()))

    # ---- parameters ---------------------------------------------------------------------
    def set_bgc_indices(self, ind30=None, T0_Kelvin=273.15):
        arr = (C.c_int * 30)(*[int(i) for i in ind30]) if ind30 is not None else None
        self.lib.obgc_bgc_params_init(C.byref(self.bgc), arr, float(T0_Kelvin))
        self.push_params()

    def set_dms_indices(self, ind14=None):
        arr = (C.c_int * 14)(*[int(i) for i in ind14]) if ind14 is not None else None
        self.lib.obgc_dms_params_init(C.byref(self.dms), arr)
        self.push_params()

    def set_macros_indices(self, ind8=None):
        arr = (C.c_int * 8)(*[int(i) for i in ind8]) if ind8 is not None else None
        self.lib.obgc_macros_params_init(C.byref(self.macros), arr)
        self.push_params()

    def push_params(self):
        self._check(self.lib.obgc_set_bgc_params(self.h, C.byref(self.bgc)), "obgc_set_bgc_params")
        self._check(self.lib.obgc_set_dms_params(self.h, C.byref(self.dms)), "obgc_set_dms_params")
        self._check(self.lib.obgc_set_macros_params(self.h, C.byref(self.macros)), "obgc_set_macros_params")

    def bgc_get_params(self):
        """Same dictionary shape as the oracle bindings (for parameter-table tests)."""
        areal = np.array([[getattr(self.bgc.autotrophs[a], n) for n in L.AUTOTROPH_REALS] for a in range(4)])
        aint = np.array([[getattr(self.bgc.autotrophs[a], n) for n in L.AUTOTROPH_INTS] for a in range(4)], dtype=np.int32)
        return dict(areal=areal, aint=aint, parms=np.array([getattr(self.bgc, n) for n in L.BGC_PARMS]),
                    scalelen_z=np.array(self.bgc.parm_scalelen_z[:]), scalelen_vals=np.array(self.bgc.parm_scalelen_vals[:]))

    def bgc_set_params(self, p):
        for a in range(4):
            for i, n in enumerate(L.AUTOTROPH_REALS):
                setattr(self.bgc.autotrophs[a], n, float(p["areal"][a][i]))
            for i, n in enumerate(L.AUTOTROPH_INTS):
                setattr(self.bgc.autotrophs[a], n, int(p["aint"][a][i]))
        for i, n in enumerate(L.BGC_PARMS):
            setattr(self.bgc, n, float(p["parms"][i]))
        for i in range(4):
            self.bgc.parm_scalelen_z[i] = float(p["scalelen_z"][i])
            self.bgc.parm_scalelen_vals[i] = float(p["scalelen_vals"][i])
        self.push_params()

    def sync(self):
        self._check(self.lib.obgc_sync(self.h), "obgc_sync")

    def last_kernel_ms(self, which=0):
        return self.lib.obgc_last_kernel_ms(self.h, which)

    def last_schedule(self):
        return self.lib.obgc_last_schedule(self.h)

    # ---- helpers ------------------------------------------------------------------------
    @staticmethod
    def _fill(struct, names, arrays, memspace):
        for nm in names:
            a = arrays.get(nm) if isinstance(arrays, dict) else getattr(arrays, nm, None)
            if a is None:
                setattr(struct, nm, None)
                continue
            p, dev = _ptr(a)
            if (L.OBGC_MEM_DEVICE if dev else L.OBGC_MEM_HOST) != memspace:
                raise ObgcError("array %s is not in the call's memory space" % nm)
            setattr(struct, nm, p)

    @staticmethod
    def _memspace(x):
        _, dev = _ptr(x)
        return L.OBGC_MEM_DEVICE if dev else L.OBGC_MEM_HOST

    # ---- compute entry points -------------------------------------------------------------
    def bgc_source_sink(self, *args, **kw):
        self.bgc_source_sink_run(self.bgc_source_sink_prepare(*args, **kw))

    def bgc_source_sink_run(self, prepared):
        """One obgc_bgc_source_sink call on an argument block built by bgc_source_sink_prepare: what a compiled host does
        per step (the arrays, hence the block, do not change between steps)."""
        self._check(self.lib.obgc_bgc_source_sink(self.h, prepared[0]), "obgc_bgc_source_sink")

    def bgc_source_sink_prepare(self, nlev, ncol, layout, tracers, temp, salt, zmid, dz, zbot, lat, kmax, fesedflux,
                                dust_flux_in, sw_flux, tend, ph_prev, ph_prev_alt, diag=None, alt_co2_use_eco=True, ld=0,
                                plane=0, ldc=0):
        ms = self._memspace(tracers)
        a = L.BgcSourceSinkArgs()
        a.grid = make_grid(nlev, ncol, layout, ms, ld, plane, ldc)
        self._fill(a, ["tracers", "temp", "salt", "zmid", "dz", "zbot", "lat", "kmax", "fesedflux", "dust_flux_in",
                       "sw_flux", "tend", "ph_prev", "ph_prev_alt"],
                   dict(tracers=tracers, temp=temp, salt=salt, zmid=zmid, dz=dz, zbot=zbot, lat=lat, kmax=kmax,
                        fesedflux=fesedflux, dust_flux_in=dust_flux_in, sw_flux=sw_flux, tend=tend, ph_prev=ph_prev,
                        ph_prev_alt=ph_prev_alt), ms)
        d = None
        if isinstance(diag, C.POINTER(L.BgcDiag)):      # a DiagSet's struct (lives as long as the set)
            a.diag = diag
        elif diag is not None:
            d = L.BgcDiag()
            self._fill(d, L.BGC_DIAG_KC + L.BGC_DIAG_KCA + L.BGC_DIAG_CA + L.BGC_DIAG_C, diag, ms)
            a.diag = C.pointer(d)
        a.alt_co2_use_eco = 1 if alt_co2_use_eco else 0
        return (C.byref(a), a, d)          # byref object + what it points to (kept alive by the tuple)

    # ---- fused BGC + DMS + MACROS step (obgc_fused_step) ----
    FUSED_EXTRA = ("dms", "dmsp", "SST", "dms_tend", "dmsp_tend", "prot", "poly", "lip", "prot_tend", "poly_tend", "lip_tend")

    def fused_step_prepare(self, nlev, ncol, layout, tracers, temp, salt, zmid, dz, zbot, lat, kmax, fesedflux, dust_flux_in,
                           sw_flux, tend, ph_prev, ph_prev_alt, diag=None, alt_co2_use_eco=True, ld=0, plane=0, ldc=0,
                           dms_diag=None, macros_diag=None, **own):
        """own: dms, dmsp, SST, dms_tend, dmsp_tend (DMS part) and / or prot, poly, lip, prot_tend, poly_tend, lip_tend
        (MACROS part): (k,column) planes in the same index space as the BGC arrays; SST is (column)."""
        _, a_bgc, d_bgc = self.bgc_source_sink_prepare(nlev, ncol, layout, tracers, temp, salt, zmid, dz, zbot, lat, kmax, fesedflux,
                                                       dust_flux_in, sw_flux, tend, ph_prev, ph_prev_alt, diag, alt_co2_use_eco,
                                                       ld, plane, ldc)
        ms = a_bgc.grid.memspace
        a = L.FusedStepArgs()
        a.bgc = a_bgc
        bad = set(own) - set(self.FUSED_EXTRA)
        if bad:
            raise ObgcError("fused_step: unknown arrays %s" % sorted(bad))
        self._fill(a, [n for n in self.FUSED_EXTRA if own.get(n) is not None], own, ms)
        keep = [d_bgc]
        if isinstance(dms_diag, C.POINTER(L.DmsDiag)):
            a.dms_diag = dms_diag
        elif dms_diag is not None:
            d = L.DmsDiag()
            self._fill(d, L.DMS_DIAG_KC, dms_diag, ms)
            a.dms_diag = C.pointer(d)
            keep.append(d)
        if isinstance(macros_diag, C.POINTER(L.MacrosDiag)):
            a.macros_diag = macros_diag
        elif macros_diag is not None:
            d = L.MacrosDiag()
            self._fill(d, L.MACROS_DIAG_KC, macros_diag, ms)
            a.macros_diag = C.pointer(d)
            keep.append(d)
        return (C.byref(a), a, keep)

    def fused_step_run(self, prepared):
        self._check(self.lib.obgc_fused_step(self.h, prepared[0]), "obgc_fused_step")

    def fused_step(self, *args, **kw):
        self.fused_step_run(self.fused_step_prepare(*args, **kw))

    def bgc_surface_fluxes(self, nlev, ncol, layout, tracers, forcing, depositionFlux, riverFlux, gasFlux, seaIceFlux,
                           netFlux, diag=None, lcalc_O2_gas_flux=True, lcalc_CO2_gas_flux=True, ld=0, plane=0, ldc=0):
        ms = self._memspace(tracers)
        a = L.BgcSurfaceFluxesArgs()
        a.grid = make_grid(nlev, ncol, layout, ms, ld, plane, ldc)
        arrs = dict(forcing)
        arrs.update(tracers=tracers, depositionFlux=depositionFlux, riverFlux=riverFlux, gasFlux=gasFlux,
                    seaIceFlux=seaIceFlux, netFlux=netFlux)
        self._fill(a, ["tracers", "surfacePressure", "windSpeedSquared10m", "atmCO2", "atmCO2_ALT_CO2", "surfaceDepth",
                       "SST", "SSS", "iceFraction", "surface_pH", "surface_pH_alt_co2", "depositionFlux", "riverFlux",
                       "gasFlux", "seaIceFlux", "netFlux"], arrs, ms)
        if diag is not None:
            d = L.BgcFluxDiag()
            self._fill(d, L.BGC_FLUX_DIAG, diag, ms)
            a.diag = C.pointer(d)
        a.lcalc_O2_gas_flux = 1 if lcalc_O2_gas_flux else 0
        a.lcalc_CO2_gas_flux = 1 if lcalc_CO2_gas_flux else 0
        self._check(self.lib.obgc_bgc_surface_fluxes(self.h, C.byref(a)), "obgc_bgc_surface_fluxes")

    def dms_source_sink(self, nlev, ncol, layout, tracers, dz, kmax, SST, sw_flux, tend, diag=None, ld=0, plane=0, ldc=0,
                        noalloc=False):
        """noalloc=True: obgc_dms_source_sink_noalloc (fixed library scratch, no device allocation in the call)."""
        ms = self._memspace(tracers)
        a = L.DmsSourceSinkArgs()
        a.grid = make_grid(nlev, ncol, layout, ms, ld, plane, ldc)
        self._fill(a, ["tracers", "dz", "kmax", "SST", "sw_flux", "tend"],
                   dict(tracers=tracers, dz=dz, kmax=kmax, SST=SST, sw_flux=sw_flux, tend=tend), ms)
        if diag is not None:
            d = L.DmsDiag()
            self._fill(d, L.DMS_DIAG_KC, diag, ms)
            a.diag = C.pointer(d)
        fn = "obgc_dms_source_sink_noalloc" if noalloc else "obgc_dms_source_sink"
        self._check(getattr(self.lib, fn)(self.h, C.byref(a)), fn)

    def dms_surface_fluxes(self, nlev, ncol, layout, tracers, forcing, netFlux, diag=None, lcalc_DMS_gas_flux=True, ld=0,
                           plane=0, ldc=0):
        ms = self._memspace(tracers)
        a = L.DmsSurfaceFluxesArgs()
        a.grid = make_grid(nlev, ncol, layout, ms, ld, plane, ldc)
        arrs = dict(forcing)
        arrs.update(tracers=tracers, netFlux=netFlux)
        self._fill(a, ["tracers", "surfacePressure", "windSpeedSquared10m", "SST", "SSS", "iceFraction", "netFlux"], arrs, ms)
        if diag is not None:
            d = L.DmsFluxDiag()
            self._fill(d, L.DMS_FLUX_DIAG, diag, ms)
            a.diag = C.pointer(d)
        a.lcalc_DMS_gas_flux = 1 if lcalc_DMS_gas_flux else 0
        self._check(self.lib.obgc_dms_surface_fluxes(self.h, C.byref(a)), "obgc_dms_surface_fluxes")

    def macros_source_sink(self, nlev, ncol, layout, tracers, kmax, tend, diag=None, ld=0, plane=0, ldc=0):
        ms = self._memspace(tracers)
        a = L.MacrosSourceSinkArgs()
        a.grid = make_grid(nlev, ncol, layout, ms, ld, plane, ldc)
        self._fill(a, ["tracers", "kmax", "tend"], dict(tracers=tracers, kmax=kmax, tend=tend), ms)
        if diag is not None:
            d = L.MacrosDiag()
            self._fill(d, L.MACROS_DIAG_KC, diag, ms)
            a.diag = C.pointer(d)
        self._check(self.lib.obgc_macros_source_sink(self.h, C.byref(a)), "obgc_macros_source_sink")

    def apply_tendencies(self, nlev, ncol, layout, tracers, tend, kmax, dt, net_flux=None, dz=None, clamp_nonneg=False,
                         ld=0, plane=0, ldc=0):
        """x <- x + dt*(tend + [k==1]*net_flux/dz) on device-resident tracers (no reference routine: the host model owns the time integration)."""
        ms = self._memspace(tracers)
        a = L.ApplyTendenciesArgs()
        a.grid = make_grid(nlev, ncol, layout, ms, ld, plane, ldc)
        a.ntracer = int(tracers.shape[0])
        vals = dict(tracers=tracers, tend=tend, kmax=kmax)
        names = ["tracers", "tend", "kmax"]
        if net_flux is not None:
            vals.update(net_flux=net_flux, dz=dz)
            names += ["net_flux", "dz"]
        self._fill(a, names, vals, ms)
        a.dt, a.clamp_nonneg = float(dt), 1 if clamp_nonneg else 0
        self._check(self.lib.obgc_apply_tendencies(self.h, C.byref(a)), "obgc_apply_tendencies")

    def bgc_diag_accumulate(self, nlev, ncol, layout, inst, mean, weight, reset=False, ld=0, plane=0, ldc=0):
        """mean[X] = (0 if reset else mean[X]) + weight*inst[X] for every diagnostic present in both dicts (device tensors)."""
        names = L.BGC_DIAG_KC + L.BGC_DIAG_KCA + L.BGC_DIAG_CA + L.BGC_DIAG_C
        any_arr = next(v for v in inst.values() if v is not None)
        ms = self._memspace(any_arr)
        g = make_grid(nlev, ncol, layout, ms, ld, plane, ldc)
        di, dm = L.BgcDiag(), L.BgcDiag()
        self._fill(di, names, inst, ms)
        self._fill(dm, names, mean, ms)
        self._check(self.lib.obgc_bgc_diag_accumulate(self.h, C.byref(g), C.byref(di), C.byref(dm), float(weight), 1 if reset else 0),
                    "obgc_bgc_diag_accumulate")

    def _diag_accumulate(self, fn, struct, names, nlev, ncol, layout, inst, mean, weight, reset, ld, plane, ldc):
        any_arr = next(v for v in inst.values() if v is not None)
        ms = self._memspace(any_arr)
        g = make_grid(nlev, ncol, layout, ms, ld, plane, ldc)
        di, dm = struct(), struct()
        self._fill(di, names, inst, ms)
        self._fill(dm, names, mean, ms)
        self._check(getattr(self.lib, fn)(self.h, C.byref(g), C.byref(di), C.byref(dm), float(weight), 1 if reset else 0), fn)

    def dms_diag_accumulate(self, nlev, ncol, layout, inst, mean, weight, reset=False, ld=0, plane=0, ldc=0):
        self._diag_accumulate("obgc_dms_diag_accumulate", L.DmsDiag, L.DMS_DIAG_KC, nlev, ncol, layout, inst, mean, weight, reset,
                              ld, plane, ldc)

    def macros_diag_accumulate(self, nlev, ncol, layout, inst, mean, weight, reset=False, ld=0, plane=0, ldc=0):
        self._diag_accumulate("obgc_macros_diag_accumulate", L.MacrosDiag, L.MACROS_DIAG_KC, nlev, ncol, layout, inst, mean, weight,
                              reset, ld, plane, ldc)

    def accumulate_arrays(self, nlev, ncol, layout, planes=(), vecs=(), weight=1.0, reset=False, ld=0, plane=0, ldc=0):
        """planes / vecs: sequences of (inst, mean) device tensors, (k,column) in `layout` resp. (column)."""
        def table(pairs, j):
            arr = (C.c_void_p * max(1, len(pairs)))()
            for i, pr in enumerate(pairs):
                addr, is_dev = _ptr(pr[j])
                if not is_dev:
                    raise ObgcError("accumulate_arrays needs device tensors")
                arr[i] = addr
            return arr
        g = make_grid(nlev, ncol, layout, L.OBGC_MEM_DEVICE, ld, plane, ldc)
        self._check(self.lib.obgc_accumulate_arrays(self.h, C.byref(g), len(planes), table(planes, 0), table(planes, 1), len(vecs),
                                                    table(vecs, 0), table(vecs, 1), float(weight), 1 if reset else 0),
                    "obgc_accumulate_arrays")

    def diag_set(self, nlev, ncol, names, layout=L.OBGC_LAYOUT_COLFAST, ld=0, plane=0, ldc=0, inst_rows=None, mean_rows=None):
        """Diagnostics selected by the reference's member names (obgc_diag_set_*): see DiagSet."""
        return DiagSet(self, nlev, ncol, names, layout, ld, plane, ldc, inst_rows, mean_rows)

    def synth_fill(self, nlev, ncol, col0, ld=0, plane=0, ind=None, tiled=False, **dev_arrays):
        """Fill device tensors (native COLFAST layout, or column blocks with tiled=True) with the synthetic workload."""
        a = L.SynthArgs()
        a.nlev, a.ncol, a.col0, a.ld, a.plane, a.ldc = nlev, ncol, col0, ld, plane, 0
        a.tiled = 1 if tiled else 0
        keep = (C.c_int * 30)(*[int(i) for i in ind]) if ind is not None else None
        a.ind = keep
        for nm in ("tracers", "temp", "salt", "zmid", "dz", "zbot", "fesedflux", "lat", "dust_flux_in", "sw_flux", "kmax"):
            x = dev_arrays.get(nm)
            if x is None:
                continue
            p, dev = _ptr(x)
            if not dev:
                raise ObgcError("synth_fill needs device tensors")
            setattr(a, nm, p)
        self._check(self.lib.obgc_synth_fill(self.h, C.byref(a)), "obgc_synth_fill")

    def synth_fill_extra(self, nlev, ncol, col0, ld=0, ldc=0, tiled=False, **dev_arrays):
        """Surface forcing, external fluxes and the DMS / MACROS own tracers of the synthetic workload (device tensors)."""
        a = L.SynthExtraArgs()
        a.nlev, a.ncol, a.col0, a.ld, a.ldc = nlev, ncol, col0, ld, ldc
        a.tiled = 1 if tiled else 0
        for nm, x in dev_arrays.items():
            if nm not in L.SYNTH_EXTRA_FIELDS:
                raise ObgcError("synth_fill_extra: unknown field %s" % nm)
            if x is None:
                continue
            p, dev = _ptr(x)
            if not dev:
                raise ObgcError("synth_fill_extra needs device tensors")
            setattr(a, nm, p)
        self._check(self.lib.obgc_synth_fill_extra(self.h, C.byref(a)), "obgc_synth_fill_extra")


def diag_names_from_file(path):
    """The qualified names of a parameter file's [diagnostics] section (obgc_diag_names_read_text; host-only)."""
    lib = L.load()
    n = C.c_int(0)
    err = C.create_string_buffer(512)
    buf = C.create_string_buffer(1 << 16)
    rc = lib.obgc_diag_names_read_text(path.encode(), buf, len(buf), C.byref(n), err, len(err))
    if rc != 0:
        raise ObgcError("obgc_diag_names_read_text: %s" % err.value.decode())
    return [x.decode() for x in buf.raw.split(b"\0")[:n.value]]


def diag_file_read(path):
    """-> (header dict, {qualified name: numpy array in the reference's order}) of a file written by DiagSet.write."""
    lib = L.load()
    nlev, ncol, nrec, which = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    if lib.obgc_diag_file_info(path.encode(), C.byref(nlev), C.byref(ncol), C.byref(nrec), C.byref(which)) != 0:
        raise ObgcError("not a diagnostics file: %s" % path)
    out = {}
    for i in range(nrec.value):
        name = C.create_string_buffer(64)
        mod, kind, nelem = C.c_int(), C.c_int(), C.c_longlong()
        if lib.obgc_diag_file_record(path.encode(), i, name, len(name), C.byref(mod), C.byref(kind), C.byref(nelem)) != 0:
            raise ObgcError("bad record %d in %s" % (i, path))
        a = np.empty(nelem.value)
        if lib.obgc_diag_file_read(path.encode(), name.value, a.ctypes.data, a.size) != 0:
            raise ObgcError("cannot read %s from %s" % (name.value.decode(), path))
        shape = {0: (ncol.value, nlev.value), 1: (4, ncol.value, nlev.value), 2: (4, ncol.value), 3: (ncol.value,)}[kind.value]
        out[name.value.decode()] = a.reshape(shape)
    return dict(nlev=nlev.value, ncol=ncol.value, nrec=nrec.value, which=which.value), out


class DiagSet:
    """A selection of diagnostics by name with its instantaneous arrays and running means on the device.
    .bgc / .dms / .macros are what to pass as diag= / dms_diag= / macros_diag= to Engine.bgc_source_sink / fused_step."""

    def __init__(self, eng, nlev, ncol, names, layout, ld, plane, ldc, inst_rows, mean_rows):
        self.eng, self.lib = eng, eng.lib
        self.nlev, self.ncol = nlev, ncol
        self.grid = make_grid(nlev, ncol, layout, L.OBGC_MEM_DEVICE, ld, plane, ldc)
        arr = (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
        s = C.c_void_p()
        ir = _ptr(inst_rows)[0] if inst_rows is not None else None
        mr = _ptr(mean_rows)[0] if mean_rows is not None else None
        eng._check(self.lib.obgc_diag_set_create(eng.h, C.byref(self.grid), len(names), arr, ir, mr, C.byref(s)), "obgc_diag_set_create")
        self.s = s
        self.bgc = self.lib.obgc_diag_set_bgc(s) or None
        self.dms = self.lib.obgc_diag_set_dms(s) or None
        self.macros = self.lib.obgc_diag_set_macros(s) or None

    @staticmethod
    def rows(names):
        lib = L.load()
        arr = (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
        r, v = C.c_int(), C.c_int()
        err = C.create_string_buffer(256)
        if lib.obgc_diag_set_rows(len(names), arr, C.byref(r), C.byref(v), err, len(err)) != 0:
            raise ObgcError(err.value.decode())
        return r.value, v.value

    def __len__(self):
        return self.lib.obgc_diag_set_count(self.s)

    def accumulate(self, weight, reset=False):
        self.eng._check(self.lib.obgc_diag_set_accumulate(self.s, float(weight), 1 if reset else 0), "obgc_diag_set_accumulate")

    def download(self, name, which="mean"):
        n = self.lib.obgc_diag_set_elems(self.s, name.encode())
        if n < 0:
            raise ObgcError("%s is not in the selection" % name)
        a = np.empty(n)
        self.eng._check(self.lib.obgc_diag_set_download(self.s, name.encode(), 0 if which == "mean" else 1, a.ctypes.data, n),
                        "obgc_diag_set_download")
        kc = self.nlev * self.ncol
        shape = {kc: (self.ncol, self.nlev), 4 * kc: (4, self.ncol, self.nlev), 4 * self.ncol: (4, self.ncol), self.ncol: (self.ncol,)}
        return a.reshape(shape[n])

    def write(self, path, which="mean"):
        self.eng._check(self.lib.obgc_diag_set_write(self.s, path.encode(), 0 if which == "mean" else 1), "obgc_diag_set_write")

    def close(self):
        if getattr(self, "s", None):
            self.lib.obgc_diag_set_destroy(self.s)
            self.s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
