"""Python mirror of the reference's module API on top of the GPU engine.

The reference's call surface is

    BGC_parms_init / BGC_init / BGC_SourceSink / BGC_SurfaceFluxes      (BGC_mod.F90:106-110)
    DMS_parms_init / DMS_init / DMS_SourceSink / DMS_SurfaceFluxes      (DMS_mod.F90:51-55)
    MACROS_parms_init / MACROS_init / MACROS_SourceSink                 (MACROS_mod.F90:51-54)

with derived-type arguments whose array components are Fortran X(k, column[, n]).  Here the
same routines take numpy arrays in exactly that memory order -- C-order shape [n, ncol, nlev]
-- and return result dictionaries shaped like the oracle bindings', so a parity test reads
`engine.BGC_SourceSink(...)` against `oracle.bgc_source_sink(...)` line for line.

`path` selects how the arrays reach the kernels (every path ends in the same HIP kernels):
  "host"            host pointers, reference layout: staged H2D/D2H by the library
  "device_kfast"    device pointers, reference layout
  "device_colfast"  device pointers, native column-fastest layout
  "device_tiled"    device pointers, column blocks of 64 in one arena (OBGC_LAYOUT_TILED; what bench.py times)
"""
import numpy as np

from . import _lib as L
from .engine import Engine, ObgcError, TileArena, TileField

PATHS = ("host", "device_kfast", "device_colfast", "device_tiled")


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise ObgcError("device paths need a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    return torch


class ColumnEngine:
    def __init__(self, device=0, bgc_indices=None, dms_indices=None, macros_indices=None, T0_Kelvin=273.15, path="host"):
        assert path in PATHS
        self.path = path
        self.device = device
        self.eng = Engine(device)
        # *_parms_init followed by *_init, as the host must call them (BGC_parms.F90:497, BGC_mod.F90:184)
        self.eng.set_bgc_indices(bgc_indices, T0_Kelvin)
        self.eng.set_dms_indices(dms_indices)
        self.eng.set_macros_indices(macros_indices)

    # ---- parameter access in the oracle bindings' shape ----
    def bgc_get_params(self):
        return self.eng.bgc_get_params()

    def bgc_set_params(self, p):
        self.eng.bgc_set_params(p)

    def dms_get_parms(self):
        return np.array([getattr(self.eng.dms, n) for n in L.DMS_PARMS])

    def macros_get_parms(self):
        return np.array([getattr(self.eng.macros, n) for n in L.MACROS_PARMS])

    def dms_set_parms(self, p):
        """The host changing DMS_parms module variables between calls (DMS_parms.F90:203-241 sets the defaults)."""
        for n, v in zip(L.DMS_PARMS, p):
            setattr(self.eng.dms, n, float(v))
        self.eng.push_params()

    def macros_set_parms(self, p):
        for n, v in zip(L.MACROS_PARMS, p):
            setattr(self.eng.macros, n, float(v))
        self.eng.push_params()

    def names(self, module):
        n = {0: 30, 1: 14, 2: 8}[module]
        return [tuple(self.eng.lib.obgc_tracer_name(module, i, w).decode() for w in range(3)) for i in range(1, n + 1)]

    # ---- array plumbing ----
    def _begin(self, nc, nz, nrows):
        """Column-block path: one arena of `nrows` 64-column rows per (block, level) for the (k,column) arrays of this call."""
        self._arena = TileArena(_torch(), "cuda:%d" % self.device, nc, nz, nrows) if self.path == "device_tiled" else None

    @property
    def _gkw(self):
        """leading dimensions of the call's grid"""
        return dict(ld=self._arena.ld, plane=64) if self.path == "device_tiled" else {}

    def _up(self, x, kind):
        """host reference-layout array -> what the selected path hands to the C ABI.
        kind: 'kc' [.., ncol, nlev]; 'c' per-column / (column,n) arrays."""
        x = np.ascontiguousarray(x)
        if self.path == "host":
            return x.copy()
        if self.path == "device_tiled" and kind == "kc":
            f = self._arena.take(x.shape[0] if x.ndim == 3 else 1, x.shape).put(x)
            self._ready()
            return f
        torch = _torch()
        t = torch.from_numpy(x).to("cuda:%d" % self.device)
        if self.path == "device_colfast" and kind == "kc":
            t = t.transpose(-1, -2).contiguous()      # [.., nlev, ncol]
        self._ready()
        return t

    def _down(self, t, kind):
        if isinstance(t, np.ndarray):
            return t
        if isinstance(t, TileField):
            return t.get()
        if self.path == "device_colfast" and kind == "kc":
            t = t.transpose(-1, -2).contiguous()
        return t.cpu().numpy()

    def _ready(self):
        """Device arrays are produced on torch's stream; the engine's streams are non-blocking and
        do not order against it, so the producer must have finished before the C-ABI call
        (include/obgc.h, "Stream ordering")."""
        if self.path != "host":
            _torch().cuda.synchronize()

    @property
    def _layout(self):
        return {"device_colfast": L.OBGC_LAYOUT_COLFAST, "device_tiled": L.OBGC_LAYOUT_TILED}.get(self.path, L.OBGC_LAYOUT_KFAST)

    @staticmethod
    def _bgc_rows(diagnostics):
        """rows of a column-block arena for BGC_SourceSink: 30 tracers, 6 fields, 30 tendencies, 2 pH (+ diagnostics)"""
        return 68 + (len(L.BGC_DIAG_KC) + 4 * len(L.BGC_DIAG_KCA) if diagnostics else 0)

    # ---- BGC_SourceSink (BGC_mod.F90:340) ----
    def BGC_SourceSink(self, tracers, temp, salt, zmid, dz, zbot, lat, kmax, fesed, dust, sw, ph_prev, ph_prev_alt,
                       alt_co2_use_eco=True, diag_init=0.0, diagnostics=True):
        n, nc, nz = tracers.shape
        self._begin(nc, nz, self._bgc_rows(diagnostics))
        up = self._up
        tend = up(np.full((30, nc, nz), diag_init), "kc")
        php, pha = up(ph_prev, "kc"), up(ph_prev_alt, "kc")
        diag = None
        if diagnostics:
            diag = {}
            for nm in L.BGC_DIAG_KC:
                diag[nm] = up(np.full((nc, nz), diag_init), "kc")
            for nm in L.BGC_DIAG_KCA:
                diag[nm] = up(np.full((4, nc, nz), diag_init), "kc")
            for nm in L.BGC_DIAG_CA:
                diag[nm] = up(np.full((4, nc), diag_init), "c")
            for nm in L.BGC_DIAG_C:
                diag[nm] = up(np.full((nc,), diag_init), "c")
        self.eng.bgc_source_sink(nz, nc, self._layout, up(tracers, "kc"), up(temp, "kc"), up(salt, "kc"), up(zmid, "kc"),
                                 up(dz, "kc"), up(zbot, "kc"), up(lat, "c"), up(np.asarray(kmax, dtype=np.int32), "c"),
                                 up(fesed, "kc"), up(dust, "c"), up(sw, "c"), tend, php, pha, diag, alt_co2_use_eco, **self._gkw)
        out = dict(tend=self._down(tend, "kc"), ph_prev=self._down(php, "kc"), ph_prev_alt=self._down(pha, "kc"),
                   kc={}, kca={}, ca={}, c={})
        if diagnostics:
            out["kc"] = {nm: self._down(diag[nm], "kc") for nm in L.BGC_DIAG_KC}
            out["kca"] = {nm: self._down(diag[nm], "kc") for nm in L.BGC_DIAG_KCA}
            out["ca"] = {nm: self._down(diag[nm], "c") for nm in L.BGC_DIAG_CA}
            out["c"] = {nm: self._down(diag[nm], "c") for nm in L.BGC_DIAG_C}
        return out

    # ---- BGC_SourceSink + DMS_SourceSink + MACROS_SourceSink in one call (obgc_fused_step) ----
    def Fused_SourceSink(self, tracers, temp, salt, zmid, dz, zbot, lat, kmax, fesed, dust, sw, ph_prev, ph_prev_alt, sst,
                         dms_tracers=None, macros_tracers=None, alt_co2_use_eco=True, diag_init=0.0, diagnostics=True):
        """dms_tracers / macros_tracers: the reference's 14- / 8-plane arrays [n, ncol, nlev]; only the OWN planes (dms, dmsp /
        prot, poly, lip at the slots of the engine's DMS / MACROS indices) are handed to the library, the shared ones must be
        copies of the BGC tracers (asserted).  Returns {"bgc", "dms", "macros"} shaped like the three separate routines'
        results; the tendency planes of the shared tracers, which the fused step does not write, are returned as zeros."""
        if self.path not in ("device_colfast", "device_tiled"):
            raise ValueError("the fused step runs on device-resident arrays in the native or the column-block layout")
        n, nc, nz = tracers.shape
        self._begin(nc, nz, self._bgc_rows(diagnostics) + 10 + (len(L.DMS_DIAG_KC) + len(L.MACROS_DIAG_KC) if diagnostics else 0))
        up = self._up
        E = self.eng
        tend = up(np.full((30, nc, nz), diag_init), "kc")
        php, pha = up(ph_prev, "kc"), up(ph_prev_alt, "kc")
        diag = None
        if diagnostics:
            diag = {}
            for nm in L.BGC_DIAG_KC:
                diag[nm] = up(np.full((nc, nz), diag_init), "kc")
            for nm in L.BGC_DIAG_KCA:
                diag[nm] = up(np.full((4, nc, nz), diag_init), "kc")
            for nm in L.BGC_DIAG_CA:
                diag[nm] = up(np.full((4, nc), diag_init), "c")
            for nm in L.BGC_DIAG_C:
                diag[nm] = up(np.full((nc,), diag_init), "c")
        own, ddiag, mdiag = {}, None, None
        bslot = {nm: E.bgc.ind[i] - 1 for i, nm in enumerate(L.BGC_TRACERS)}
        if dms_tracers is not None:
            ds = [E.dms.ind[i] - 1 for i in range(14)]
            for j, nm in enumerate(L.DMS_TRACERS[2:], start=2):
                assert np.array_equal(dms_tracers[ds[j]], tracers[bslot[nm]], equal_nan=True), "DMS tracer %s is not the BGC tracer" % nm
            own.update(dms=up(dms_tracers[ds[0]], "kc"), dmsp=up(dms_tracers[ds[1]], "kc"), SST=up(sst, "c"),
                       dms_tend=up(np.full((nc, nz), diag_init), "kc"), dmsp_tend=up(np.full((nc, nz), diag_init), "kc"))
            if diagnostics:
                ddiag = {nm: up(np.full((nc, nz), diag_init), "kc") for nm in L.DMS_DIAG_KC}
        if macros_tracers is not None:
            ms = [E.macros.ind[i] - 1 for i in range(8)]
            for j, nm in enumerate(L.MACROS_TRACERS[3:], start=3):
                assert np.array_equal(macros_tracers[ms[j]], tracers[bslot[nm]], equal_nan=True), "MACROS tracer %s is not the BGC tracer" % nm
            own.update(prot=up(macros_tracers[ms[0]], "kc"), poly=up(macros_tracers[ms[1]], "kc"), lip=up(macros_tracers[ms[2]], "kc"),
                       prot_tend=up(np.full((nc, nz), diag_init), "kc"), poly_tend=up(np.full((nc, nz), diag_init), "kc"),
                       lip_tend=up(np.full((nc, nz), diag_init), "kc"))
            if diagnostics:
                mdiag = {nm: up(np.full((nc, nz), diag_init), "kc") for nm in L.MACROS_DIAG_KC}
        E.fused_step(nz, nc, self._layout, up(tracers, "kc"), up(temp, "kc"), up(salt, "kc"), up(zmid, "kc"), up(dz, "kc"),
                     up(zbot, "kc"), up(lat, "c"), up(np.asarray(kmax, dtype=np.int32), "c"), up(fesed, "kc"), up(dust, "c"),
                     up(sw, "c"), tend, php, pha, diag, alt_co2_use_eco, dms_diag=ddiag, macros_diag=mdiag, **own, **self._gkw)
        dn = self._down
        bgc = dict(tend=dn(tend, "kc"), ph_prev=dn(php, "kc"), ph_prev_alt=dn(pha, "kc"), kc={}, kca={}, ca={}, c={})
        if diagnostics:
            bgc["kc"] = {nm: dn(diag[nm], "kc") for nm in L.BGC_DIAG_KC}
            bgc["kca"] = {nm: dn(diag[nm], "kc") for nm in L.BGC_DIAG_KCA}
            bgc["ca"] = {nm: dn(diag[nm], "c") for nm in L.BGC_DIAG_CA}
            bgc["c"] = {nm: dn(diag[nm], "c") for nm in L.BGC_DIAG_C}
        out = dict(bgc=bgc)
        if dms_tracers is not None:
            t = np.zeros((14, nc, nz))
            t[ds[0]], t[ds[1]] = dn(own["dms_tend"], "kc"), dn(own["dmsp_tend"], "kc")
            out["dms"] = dict(tend=t, kc={nm: dn(v, "kc") for nm, v in ddiag.items()} if ddiag else {})
        if macros_tracers is not None:
            t = np.zeros((8, nc, nz))
            for j, nm in enumerate(("prot_tend", "poly_tend", "lip_tend")):
                t[ms[j]] = dn(own[nm], "kc")
            out["macros"] = dict(tend=t, kc={nm: dn(v, "kc") for nm, v in mdiag.items()} if mdiag else {})
        return out

    # ---- BGC_SurfaceFluxes (BGC_mod.F90:2706) ----
    def BGC_SurfaceFluxes(self, tracers, forcing, depo, river, gas, seaice, net=None, lcalc_o2=True, lcalc_co2=True,
                          diag_init=0.0):
        n, nc, nz = tracers.shape
        self._begin(nc, nz, 30)
        up = self._up
        f = {nm: up(np.asarray(v, dtype=np.float64), "c") for nm, v in forcing.items()}
        d, r, g, s = (up(x, "c") for x in (depo, river, gas, seaice))
        netf = up(np.full((30, nc), diag_init) if net is None else net, "c")
        fd = {nm: up(np.full((nc,), diag_init), "c") for nm in L.BGC_FLUX_DIAG}
        self.eng.bgc_surface_fluxes(nz, nc, self._layout, up(tracers, "kc"), f, d, r, g, s, netf, fd, lcalc_o2, lcalc_co2, **self._gkw)
        dn = self._down
        return dict(forcing={nm: dn(v, "c") for nm, v in f.items()}, depositionFlux=dn(d, "c"), riverFlux=dn(r, "c"),
                    gasFlux=dn(g, "c"), seaIceFlux=dn(s, "c"), netFlux=dn(netf, "c"),
                    fdiag={nm: dn(v, "c") for nm, v in fd.items()})

    # ---- DMS (DMS_mod.F90:156, 778) ----
    def DMS_SourceSink(self, tracers, dz, kmax, sst, sw, diag_init=0.0):
        n, nc, nz = tracers.shape
        self._begin(nc, nz, 14 + 1 + 14 + len(L.DMS_DIAG_KC))
        up = self._up
        tend = up(np.full((14, nc, nz), diag_init), "kc")
        diag = {nm: up(np.full((nc, nz), diag_init), "kc") for nm in L.DMS_DIAG_KC}
        self.eng.dms_source_sink(nz, nc, self._layout, up(tracers, "kc"), up(dz, "kc"),
                                 up(np.asarray(kmax, dtype=np.int32), "c"), up(sst, "c"), up(sw, "c"), tend, diag, **self._gkw)
        return dict(tend=self._down(tend, "kc"), kc={nm: self._down(v, "kc") for nm, v in diag.items()})

    def DMS_SurfaceFluxes(self, tracers, forcing, net=None, lcalc=True, diag_init=0.0):
        n, nc, nz = tracers.shape
        self._begin(nc, nz, 14)
        up = self._up
        f = {nm: up(np.asarray(forcing[nm], dtype=np.float64), "c")
             for nm in ("surfacePressure", "iceFraction", "windSpeedSquared10m", "SST", "SSS")}
        netf = up(np.full((14, nc), diag_init) if net is None else net, "c")
        fd = {nm: up(np.full((nc,), diag_init), "c") for nm in L.DMS_FLUX_DIAG}
        self.eng.dms_surface_fluxes(nz, nc, self._layout, up(tracers, "kc"), f, netf, fd, lcalc, **self._gkw)
        dn = self._down
        return dict(forcing={nm: dn(v, "c") for nm, v in f.items()}, netFlux=dn(netf, "c"),
                    fdiag={nm: dn(v, "c") for nm, v in fd.items()})

    # ---- resident tracers (no reference routine) ----
    def apply_tendencies(self, tracers, tend, kmax, dt, net_flux=None, dz=None, clamp_nonneg=False):
        """Returns x + dt*(tend + [k==1]*net_flux/dz) computed on the device (device paths only)."""
        if self.path == "host":
            raise ValueError("apply_tendencies needs a device path: the point is tracers that stay on the GPU")
        n, nc, nz = tracers.shape
        self._begin(nc, nz, 2 * n + 1)
        up = self._up
        x = up(tracers, "kc")
        self.eng.apply_tendencies(nz, nc, self._layout, x, up(tend, "kc"), up(np.asarray(kmax, dtype=np.int32), "c"), dt,
                                  None if net_flux is None else up(net_flux, "c"), None if dz is None else up(dz, "kc"),
                                  clamp_nonneg, **self._gkw)
        return self._down(x, "kc")

    # ---- MACROS (MACROS_mod.F90:137) ----
    def MACROS_SourceSink(self, tracers, kmax, diag_init=0.0):
        n, nc, nz = tracers.shape
        self._begin(nc, nz, 8 + 8 + len(L.MACROS_DIAG_KC))
        up = self._up
        tend = up(np.full((8, nc, nz), diag_init), "kc")
        diag = {nm: up(np.full((nc, nz), diag_init), "kc") for nm in L.MACROS_DIAG_KC}
        self.eng.macros_source_sink(nz, nc, self._layout, up(tracers, "kc"), up(np.asarray(kmax, dtype=np.int32), "c"),
                                    tend, diag, **self._gkw)
        return dict(tend=self._down(tend, "kc"), kc={nm: self._down(v, "kc") for nm, v in diag.items()})
