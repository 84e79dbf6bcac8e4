"""ctypes binding to oracle/liboracle.so (the plain-C restatement, obgc_oracle.c).

ORACLE / TEST INFRASTRUCTURE ONLY.  Same call surface and result dictionaries as
oracle/refbind.py::Ref so tests can diff the two directly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import fields as F

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


def available():
    return os.path.exists(LIB_PATH)


# mirrors ora_autotroph / ora_bgc_params in obgc_oracle.h
class Autotroph(C.Structure):
    _fields_ = [(n, C.c_int) for n in F.AUTOTROPH_INTS] + [(n, C.c_double) for n in F.AUTOTROPH_REALS]


class BgcParams(C.Structure):
    _fields_ = ([("autotrophs", Autotroph * 4), ("ind", C.c_int * 30),
                 ("sp_ind", C.c_int), ("diat_ind", C.c_int), ("diaz_ind", C.c_int), ("phaeo_ind", C.c_int)] +
                [(n, C.c_double) for n in F.BGC_PARMS] +
                [("parm_scalelen_z", C.c_double * 4), ("parm_scalelen_vals", C.c_double * 4),
                 ("T0_Kelvin_BGC", C.c_double),
                 ("lrest_po4", C.c_int), ("lrest_no3", C.c_int), ("lrest_sio3", C.c_int)])


class DmsParams(C.Structure):
    _fields_ = [("ind", C.c_int * 14)] + [(n, C.c_double) for n in F.DMS_PARMS]


class MacrosParams(C.Structure):
    _fields_ = [("ind", C.c_int * 8)] + [(n, C.c_double) for n in F.MACROS_PARMS]


class Oracle:
    def __init__(self, bgc_indices=None, dms_indices=None, macros_indices=None, T0_Kelvin=273.15):
        if not available():
            build()
        self.lib = L = C.CDLL(LIB_PATH)
        assert L.ora_sizeof_bgc_params() == C.sizeof(BgcParams), "BgcParams layout drifted from obgc_oracle.h"
        assert L.ora_sizeof_dms_params() == C.sizeof(DmsParams)
        assert L.ora_sizeof_macros_params() == C.sizeof(MacrosParams)
        self.bgc = BgcParams()
        self.dms = DmsParams()
        self.mac = MacrosParams()
        bi = np.asarray(bgc_indices if bgc_indices is not None else np.arange(1, 31), dtype=np.int32)
        di = np.asarray(dms_indices if dms_indices is not None else np.arange(1, 15), dtype=np.int32)
        mi = np.asarray(macros_indices if macros_indices is not None else np.arange(1, 9), dtype=np.int32)
        L.ora_bgc_params_init.argtypes = [C.POINTER(BgcParams), _ip, C.c_double]
        L.ora_dms_params_init.argtypes = [C.POINTER(DmsParams), _ip]
        L.ora_macros_params_init.argtypes = [C.POINTER(MacrosParams), _ip]
        L.ora_bgc_params_init(C.byref(self.bgc), bi, T0_Kelvin)
        L.ora_dms_params_init(C.byref(self.dms), di)
        L.ora_macros_params_init(C.byref(self.mac), mi)
        L.ora_talk_row_calls.restype = C.c_longlong

    # ---- parameters in the same shape as Ref.bgc_get_params -----------------
    def bgc_get_params(self):
        areal = np.array([[getattr(self.bgc.autotrophs[a], n) for n in F.AUTOTROPH_REALS] for a in range(4)])
        aint = np.array([[getattr(self.bgc.autotrophs[a], n) for n in F.AUTOTROPH_INTS] for a in range(4)],
                        dtype=np.int32)
        parms = np.array([getattr(self.bgc, n) for n in F.BGC_PARMS])
        return dict(areal=areal, aint=aint, parms=parms, scalelen_z=np.array(self.bgc.parm_scalelen_z[:]),
                    scalelen_vals=np.array(self.bgc.parm_scalelen_vals[:]))

    def bgc_set_params(self, p):
        for a in range(4):
            for i, n in enumerate(F.AUTOTROPH_REALS):
                setattr(self.bgc.autotrophs[a], n, float(p["areal"][a][i]))
            for i, n in enumerate(F.AUTOTROPH_INTS):
                setattr(self.bgc.autotrophs[a], n, int(p["aint"][a][i]))
        for i, n in enumerate(F.BGC_PARMS):
            setattr(self.bgc, n, float(p["parms"][i]))
        for i in range(4):
            self.bgc.parm_scalelen_z[i] = float(p["scalelen_z"][i])
            self.bgc.parm_scalelen_vals[i] = float(p["scalelen_vals"][i])

    def dms_get_parms(self):
        return np.array([getattr(self.dms, n) for n in F.DMS_PARMS])

    def dms_set_parms(self, p):
        for i, n in enumerate(F.DMS_PARMS):
            setattr(self.dms, n, float(p[i]))

    def macros_get_parms(self):
        return np.array([getattr(self.mac, n) for n in F.MACROS_PARMS])

    def macros_set_parms(self, p):
        for i, n in enumerate(F.MACROS_PARMS):
            setattr(self.mac, n, float(p[i]))

    # ---- BGC ------------------------------------------------------------------
    def bgc_source_sink(self, tracers, temp, salt, zmid, dz, zbot, lat, kmax, fesed, dust, sw,
                        ph_prev, ph_prev_alt, alt_co2_use_eco=True, diag_init=0.0, diagnostics=True):
        n, nc, nz = tracers.shape
        assert n == 30
        tend = np.full((30, nc, nz), diag_init)
        php = np.ascontiguousarray(ph_prev, dtype=np.float64).copy()
        pha = np.ascontiguousarray(ph_prev_alt, dtype=np.float64).copy()
        dkc = np.full((len(F.BGC_DIAG_KC), nc, nz), diag_init)
        dkca = np.full((len(F.BGC_DIAG_KCA), 4, nc, nz), diag_init)
        dca = np.full((len(F.BGC_DIAG_CA), 4, nc), diag_init)
        dc = np.full((len(F.BGC_DIAG_C), nc), diag_init)
        fn = self.lib.ora_bgc_source_sink
        vp = C.c_void_p
        fn.argtypes = [C.POINTER(BgcParams), C.c_int, C.c_int] + [_dp] * 7 + [_ip, _dp, _dp, _dp] + \
                      [_dp] * 3 + [vp] * 4 + [C.c_int]
        ca = np.ascontiguousarray
        dptr = (lambda a: a.ctypes.data_as(vp)) if diagnostics else (lambda a: None)
        fn(C.byref(self.bgc), nz, nc, ca(tracers), ca(temp), ca(salt), ca(zmid), ca(dz), ca(zbot), ca(lat),
           ca(kmax, dtype=np.int32), ca(fesed), ca(dust), ca(sw), tend, php, pha,
           dptr(dkc), dptr(dkca), dptr(dca), dptr(dc), 1 if alt_co2_use_eco else 0)
        return dict(tend=tend, ph_prev=php, ph_prev_alt=pha,
                    kc={nm: dkc[i] for i, nm in enumerate(F.BGC_DIAG_KC)},
                    kca={nm: dkca[i] for i, nm in enumerate(F.BGC_DIAG_KCA)},
                    ca={nm: dca[i] for i, nm in enumerate(F.BGC_DIAG_CA)},
                    c={nm: dc[i] for i, nm in enumerate(F.BGC_DIAG_C)})

    def bgc_surface_fluxes(self, tracers, forcing, depo, river, gas, seaice, net=None,
                           lcalc_o2=True, lcalc_co2=True, diag_init=0.0):
        n, nc, nz = tracers.shape
        frc = np.stack([np.asarray(forcing[nm], dtype=np.float64) for nm in F.BGC_SURF_FORCING])
        d, r, g, s = (np.ascontiguousarray(x, dtype=np.float64).copy() for x in (depo, river, gas, seaice))
        netf = np.full((30, nc), diag_init) if net is None else np.ascontiguousarray(net).copy()
        fd = np.full((len(F.BGC_FLUX_DIAG), nc), diag_init)
        fn = self.lib.ora_bgc_surface_fluxes
        fn.argtypes = [C.POINTER(BgcParams), C.c_int, C.c_int] + [_dp] * 8 + [C.c_int, C.c_int]
        fn(C.byref(self.bgc), nz, nc, np.ascontiguousarray(tracers), frc, d, r, g, s, netf, fd,
           1 if lcalc_o2 else 0, 1 if lcalc_co2 else 0)
        return dict(forcing={nm: frc[i] for i, nm in enumerate(F.BGC_SURF_FORCING)},
                    depositionFlux=d, riverFlux=r, gasFlux=g, seaIceFlux=s, netFlux=netf,
                    fdiag={nm: fd[i] for i, nm in enumerate(F.BGC_FLUX_DIAG)})

    # ---- co2calc ------------------------------------------------------------
    def comp_co3terms(self, k, depth, temp, salt, dic, ta, pt, sit, phlo, phhi):
        n = len(k)
        o = [np.zeros(n) for _ in range(4)]
        fn = self.lib.ora_comp_co3terms
        fn.argtypes = [C.c_int, _ip] + [_dp] * 13
        a = lambda x: np.ascontiguousarray(x, dtype=np.float64)
        fn(n, np.ascontiguousarray(k, dtype=np.int32), a(depth), a(temp), a(salt), a(dic), a(ta), a(pt),
           a(sit), a(phlo), a(phhi), *o)
        return dict(ph=o[0], h2co3=o[1], hco3=o[2], co3=o[3], talk_calls=self.lib.ora_talk_row_calls())

    def comp_co3_sat_vals(self, k, depth, temp, salt):
        n = len(k)
        o = [np.zeros(n) for _ in range(2)]
        fn = self.lib.ora_comp_co3_sat_vals
        fn.argtypes = [C.c_int, _ip] + [_dp] * 5
        a = lambda x: np.ascontiguousarray(x, dtype=np.float64)
        fn(n, np.ascontiguousarray(k, dtype=np.int32), a(depth), a(temp), a(salt), *o)
        return dict(calc=o[0], arag=o[1])

    def co2calc_1point(self, depth, temp, salt, dic, ta, pt, sit, phlo, phhi, xco2, atmpres):
        n = len(depth)
        o = [np.zeros(n) for _ in range(5)]
        fn = self.lib.ora_co2calc_1point
        fn.argtypes = [C.c_int] + [_dp] * 16
        a = lambda x: np.ascontiguousarray(x, dtype=np.float64)
        fn(n, a(depth), a(temp), a(salt), a(dic), a(ta), a(pt), a(sit), a(phlo), a(phhi), a(xco2), a(atmpres), *o)
        return dict(ph=o[0], co2star=o[1], dco2star=o[2], pco2surf=o[3], dpco2=o[4])

    # ---- DMS / MACROS -------------------------------------------------------
    def dms_source_sink(self, tracers, dz, kmax, sst, sw, diag_init=0.0):
        n, nc, nz = tracers.shape
        tend = np.full((14, nc, nz), diag_init)
        dkc = np.full((len(F.DMS_DIAG_KC), nc, nz), diag_init)
        fn = self.lib.ora_dms_source_sink
        fn.argtypes = [C.POINTER(DmsParams), C.c_int, C.c_int, _dp, _dp, _ip, _dp, _dp, _dp, _dp]
        ca = np.ascontiguousarray
        fn(C.byref(self.dms), nz, nc, ca(tracers), ca(dz), ca(kmax, dtype=np.int32), ca(sst), ca(sw), tend, dkc)
        return dict(tend=tend, kc={nm: dkc[i] for i, nm in enumerate(F.DMS_DIAG_KC)})

    def dms_surface_fluxes(self, tracers, forcing, net=None, lcalc=True, diag_init=0.0):
        n, nc, nz = tracers.shape
        frc = np.stack([np.asarray(forcing[nm], dtype=np.float64) for nm in F.DMS_SURF_FORCING])
        netf = np.full((14, nc), diag_init) if net is None else np.ascontiguousarray(net).copy()
        fd = np.full((len(F.DMS_FLUX_DIAG), nc), diag_init)
        fn = self.lib.ora_dms_surface_fluxes
        fn.argtypes = [C.POINTER(DmsParams), C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int]
        fn(C.byref(self.dms), nz, nc, np.ascontiguousarray(tracers), frc, netf, fd, 1 if lcalc else 0)
        return dict(forcing={nm: frc[i] for i, nm in enumerate(F.DMS_SURF_FORCING)}, netFlux=netf,
                    fdiag={nm: fd[i] for i, nm in enumerate(F.DMS_FLUX_DIAG)})

    # ---- branch-coverage counters ----------------------------------
    def branch_reset(self):
        self.lib.ora_branch_reset()

    def branch_hits(self):
        self.lib.ora_branch_name.restype = C.c_char_p
        self.lib.ora_branch_hit.restype = C.c_ulonglong
        return {self.lib.ora_branch_name(i).decode(): int(self.lib.ora_branch_hit(i)) for i in range(self.lib.ora_branch_count())}

    def apply_tendencies(self, tracers, tend, kmax, dt, net_flux=None, dz=None, clamp_nonneg=False):
        """x + dt*(tend + [k==1]*net_flux/dz); returns the updated copy (parity unpinned: no reference routine)."""
        n, nc, nz = tracers.shape
        x = np.ascontiguousarray(tracers, dtype=np.float64).copy()
        fn = self.lib.ora_apply_tendencies
        vp = C.c_void_p
        fn.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, vp, vp, _ip, C.c_double, C.c_int]
        fn.restype = None
        nf = np.ascontiguousarray(net_flux, dtype=np.float64) if net_flux is not None else None
        dzc = np.ascontiguousarray(dz, dtype=np.float64) if dz is not None else None
        fn(nz, nc, n, x, np.ascontiguousarray(tend, dtype=np.float64), nf.ctypes.data_as(vp) if nf is not None else None,
           dzc.ctypes.data_as(vp) if dzc is not None else None, np.ascontiguousarray(kmax, dtype=np.int32), float(dt),
           1 if clamp_nonneg else 0)
        return x

    def macros_source_sink(self, tracers, kmax, diag_init=0.0):
        n, nc, nz = tracers.shape
        tend = np.full((8, nc, nz), diag_init)
        dkc = np.full((len(F.MACROS_DIAG_KC), nc, nz), diag_init)
        fn = self.lib.ora_macros_source_sink
        fn.argtypes = [C.POINTER(MacrosParams), C.c_int, C.c_int, _dp, _ip, _dp, _dp]
        fn(C.byref(self.mac), nz, nc, np.ascontiguousarray(tracers), np.ascontiguousarray(kmax, dtype=np.int32),
           tend, dkc)
        return dict(tend=tend, kc={nm: dkc[i] for i, nm in enumerate(F.MACROS_DIAG_KC)})
