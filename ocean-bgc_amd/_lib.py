"""ctypes declarations for libobgc.so (include/obgc.h).  Loading fails loudly: there is
no CPU fallback anywhere in this package."""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# OBGC_LIB selects another build of the same ABI (the reference-order contract build, `make contract`)
LIB_PATH = os.environ.get("OBGC_LIB") or os.path.join(_HERE, "libobgc.so")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")


def _xlist(macro):
    """Expand one X-macro list of include/obgc_fields.h into a list of names."""
    txt = open(os.path.join(INCLUDE, "obgc_fields.h")).read()
    m = re.search(r"#define\s+%s\(X\)(.*?)(?:\n\s*\n|\n#|\n/\*)" % re.escape(macro), txt, re.S)
    if not m:
        raise RuntimeError("X-macro %s not found in obgc_fields.h" % macro)
    return re.findall(r"X\((\w+)\)", m.group(1))


BGC_TRACERS = _xlist("OBGC_BGC_TRACERS")
BGC_DIAG_KC = _xlist("OBGC_BGC_DIAG_KC")
BGC_DIAG_KCA = _xlist("OBGC_BGC_DIAG_KCA")
BGC_DIAG_CA = _xlist("OBGC_BGC_DIAG_CA")
BGC_DIAG_C = _xlist("OBGC_BGC_DIAG_C")
BGC_FLUX_DIAG = _xlist("OBGC_BGC_FLUX_DIAG")
DMS_TRACERS = _xlist("OBGC_DMS_TRACERS")
DMS_DIAG_KC = _xlist("OBGC_DMS_DIAG_KC")
DMS_FLUX_DIAG = _xlist("OBGC_DMS_FLUX_DIAG")
DMS_PARMS = _xlist("OBGC_DMS_PARMS")
MACROS_TRACERS = _xlist("OBGC_MACROS_TRACERS")
MACROS_DIAG_KC = _xlist("OBGC_MACROS_DIAG_KC")
MACROS_PARMS = _xlist("OBGC_MACROS_PARMS")
BGC_PARMS = _xlist("OBGC_BGC_PARMS")
AUTOTROPH_INTS = _xlist("OBGC_AUTOTROPH_INTS")
AUTOTROPH_REALS = _xlist("OBGC_AUTOTROPH_REALS")
BGC_DIAG_KC_UNTOUCHED = ["POC_ACCUM", "DONr_remin", "DOPr_remin"]

OBGC_LAYOUT_KFAST, OBGC_LAYOUT_COLFAST, OBGC_LAYOUT_TILED = 0, 1, 2
OBGC_MEM_HOST, OBGC_MEM_DEVICE = 0, 1
OBGC_ASYNC = 1
STATUS = {0: "OBGC_OK", -1: "OBGC_ERR_INVALID", -2: "OBGC_ERR_NO_DEVICE", -3: "OBGC_ERR_HIP",
          -4: "OBGC_ERR_ALLOC", -5: "OBGC_ERR_UNSUPPORTED"}

_dptr = C.c_void_p


class Grid(C.Structure):
    _fields_ = [("nlev", C.c_int), ("ncol", C.c_int), ("layout", C.c_int), ("memspace", C.c_int),
                ("ld", C.c_longlong), ("plane", C.c_longlong), ("ldc", C.c_longlong)]


class Autotroph(C.Structure):
    _fields_ = [(n, C.c_int) for n in AUTOTROPH_INTS] + [(n, C.c_double) for n in AUTOTROPH_REALS]


class BgcParams(C.Structure):
    _fields_ = ([("autotrophs", Autotroph * 4), ("ind", C.c_int * 30), ("sp_ind", C.c_int), ("diat_ind", C.c_int),
                 ("diaz_ind", C.c_int), ("phaeo_ind", C.c_int)] + [(n, C.c_double) for n in BGC_PARMS] +
                [("parm_scalelen_z", C.c_double * 4), ("parm_scalelen_vals", C.c_double * 4),
                 ("T0_Kelvin_BGC", C.c_double), ("lrest_po4", C.c_int), ("lrest_no3", C.c_int), ("lrest_sio3", C.c_int)])


class DmsParams(C.Structure):
    _fields_ = [("ind", C.c_int * 14)] + [(n, C.c_double) for n in DMS_PARMS]


class MacrosParams(C.Structure):
    _fields_ = [("ind", C.c_int * 8)] + [(n, C.c_double) for n in MACROS_PARMS]


class BgcDiag(C.Structure):
    _fields_ = [(n, _dptr) for n in BGC_DIAG_KC + BGC_DIAG_KCA + BGC_DIAG_CA + BGC_DIAG_C]


class BgcFluxDiag(C.Structure):
    _fields_ = [(n, _dptr) for n in BGC_FLUX_DIAG]


class DmsDiag(C.Structure):
    _fields_ = [(n, _dptr) for n in DMS_DIAG_KC]


class DmsFluxDiag(C.Structure):
    _fields_ = [(n, _dptr) for n in DMS_FLUX_DIAG]


class MacrosDiag(C.Structure):
    _fields_ = [(n, _dptr) for n in MACROS_DIAG_KC]


class BgcSourceSinkArgs(C.Structure):
    _fields_ = [("grid", Grid), ("tracers", _dptr), ("temp", _dptr), ("salt", _dptr), ("zmid", _dptr), ("dz", _dptr),
                ("zbot", _dptr), ("lat", _dptr), ("kmax", _dptr), ("fesedflux", _dptr), ("dust_flux_in", _dptr),
                ("sw_flux", _dptr), ("tend", _dptr), ("ph_prev", _dptr), ("ph_prev_alt", _dptr),
                ("diag", C.POINTER(BgcDiag)), ("alt_co2_use_eco", C.c_int)]


class FusedStepArgs(C.Structure):
    _fields_ = [("bgc", BgcSourceSinkArgs), ("dms", _dptr), ("dmsp", _dptr), ("SST", _dptr), ("dms_tend", _dptr),
                ("dmsp_tend", _dptr), ("dms_diag", C.POINTER(DmsDiag)), ("prot", _dptr), ("poly", _dptr), ("lip", _dptr),
                ("prot_tend", _dptr), ("poly_tend", _dptr), ("lip_tend", _dptr), ("macros_diag", C.POINTER(MacrosDiag))]


class BgcSurfaceFluxesArgs(C.Structure):
    _fields_ = [("grid", Grid), ("tracers", _dptr), ("surfacePressure", _dptr), ("windSpeedSquared10m", _dptr),
                ("atmCO2", _dptr), ("atmCO2_ALT_CO2", _dptr), ("surfaceDepth", _dptr), ("SST", _dptr), ("SSS", _dptr),
                ("iceFraction", _dptr), ("surface_pH", _dptr), ("surface_pH_alt_co2", _dptr),
                ("depositionFlux", _dptr), ("riverFlux", _dptr), ("gasFlux", _dptr), ("seaIceFlux", _dptr),
                ("netFlux", _dptr), ("diag", C.POINTER(BgcFluxDiag)), ("lcalc_O2_gas_flux", C.c_int),
                ("lcalc_CO2_gas_flux", C.c_int)]


class DmsSourceSinkArgs(C.Structure):
    _fields_ = [("grid", Grid), ("tracers", _dptr), ("dz", _dptr), ("kmax", _dptr), ("SST", _dptr), ("sw_flux", _dptr),
                ("tend", _dptr), ("diag", C.POINTER(DmsDiag))]


class DmsSurfaceFluxesArgs(C.Structure):
    _fields_ = [("grid", Grid), ("tracers", _dptr), ("surfacePressure", _dptr), ("windSpeedSquared10m", _dptr),
                ("SST", _dptr), ("SSS", _dptr), ("iceFraction", _dptr), ("netFlux", _dptr),
                ("diag", C.POINTER(DmsFluxDiag)), ("lcalc_DMS_gas_flux", C.c_int)]


class MacrosSourceSinkArgs(C.Structure):
    _fields_ = [("grid", Grid), ("tracers", _dptr), ("kmax", _dptr), ("tend", _dptr), ("diag", C.POINTER(MacrosDiag))]


class ApplyTendenciesArgs(C.Structure):
    _fields_ = [("grid", Grid), ("ntracer", C.c_int), ("tracers", _dptr), ("tend", _dptr), ("net_flux", _dptr), ("dz", _dptr),
                ("kmax", _dptr), ("dt", C.c_double), ("clamp_nonneg", C.c_int)]


class BlockStepArgs(C.Structure):
    _fields_ = [(n, _dptr) for n in (
        "ecosysTracers", "layerThickness", "zMid", "temperature", "salinity", "maxLevelCell", "latCell", "shortWaveHeatFlux",
        "atmosphericPressure", "iceFraction", "windSpeedSquared10m", "atmosphericCO2", "atmosphericCO2_ALT_CO2", "dustFlux",
        "ironSedimentFlux", "depositionFlux", "riverFlux", "seaIceFlux", "ecosysTracersTend", "ecosysSurfaceFlux")] + \
        [("accumulate", C.c_int)] + \
        [(n, _dptr) for n in ("dmsTracers", "dmsTracersTend", "dmsSurfaceFlux", "macrosTracers", "macrosTracersTend")]


class SynthArgs(C.Structure):
    _fields_ = [("nlev", C.c_int), ("ncol", C.c_int), ("col0", C.c_longlong), ("ld", C.c_longlong),
                ("plane", C.c_longlong), ("ldc", C.c_longlong), ("ind", C.POINTER(C.c_int)), ("tracers", _dptr),
                ("temp", _dptr), ("salt", _dptr), ("zmid", _dptr), ("dz", _dptr), ("zbot", _dptr), ("fesedflux", _dptr),
                ("lat", _dptr), ("dust_flux_in", _dptr), ("sw_flux", _dptr), ("kmax", _dptr), ("tiled", C.c_int)]


SYNTH_EXTRA_FIELDS = ("surfacePressure", "iceFraction", "windSpeedSquared10m", "atmCO2", "atmCO2_ALT_CO2", "surfaceDepth", "SST",
                      "SSS", "surface_pH", "surface_pH_alt_co2", "depositionFlux", "riverFlux", "gasFlux", "seaIceFlux",
                      "dms", "dmsp", "prot", "poly", "lip")


class SynthExtraArgs(C.Structure):
    _fields_ = [("nlev", C.c_int), ("ncol", C.c_int), ("col0", C.c_longlong), ("ld", C.c_longlong), ("ldc", C.c_longlong)] + \
        [(n, _dptr) for n in SYNTH_EXTRA_FIELDS] + [("tiled", C.c_int)]


# every symbol include/obgc.h declares, with its prototype
PROTOTYPES = {
    "obgc_abi_version": (C.c_int, []),
    "obgc_sizeof_bgc_params": (C.c_int, []),
    "obgc_sizeof_dms_params": (C.c_int, []),
    "obgc_sizeof_macros_params": (C.c_int, []),
    "obgc_sizeof_bgc_diag": (C.c_int, []),
    "obgc_bgc_params_init": (None, [C.POINTER(BgcParams), C.POINTER(C.c_int), C.c_double]),
    "obgc_bgc_params_derive": (None, [C.POINTER(BgcParams)]),
    "obgc_dms_params_init": (None, [C.POINTER(DmsParams), C.POINTER(C.c_int)]),
    "obgc_macros_params_init": (None, [C.POINTER(MacrosParams), C.POINTER(C.c_int)]),
    "obgc_tracer_name": (C.c_char_p, [C.c_int, C.c_int, C.c_int]),
    "obgc_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_uint]),
    "obgc_destroy": (C.c_int, [C.c_void_p]),
    "obgc_sync": (C.c_int, [C.c_void_p]),
    "obgc_last_error": (C.c_char_p, [C.c_void_p]),
    "obgc_set_bgc_params": (C.c_int, [C.c_void_p, C.POINTER(BgcParams)]),
    "obgc_set_dms_params": (C.c_int, [C.c_void_p, C.POINTER(DmsParams)]),
    "obgc_set_macros_params": (C.c_int, [C.c_void_p, C.POINTER(MacrosParams)]),
    "obgc_stream": (C.c_void_p, [C.c_void_p]),
    "obgc_last_kernel_ms": (C.c_double, [C.c_void_p, C.c_int]),
    "obgc_last_schedule": (C.c_int, [C.c_void_p]),
    "obgc_bgc_source_sink": (C.c_int, [C.c_void_p, C.POINTER(BgcSourceSinkArgs)]),
    "obgc_fused_step": (C.c_int, [C.c_void_p, C.POINTER(FusedStepArgs)]),
    "obgc_bgc_surface_fluxes": (C.c_int, [C.c_void_p, C.POINTER(BgcSurfaceFluxesArgs)]),
    "obgc_dms_source_sink": (C.c_int, [C.c_void_p, C.POINTER(DmsSourceSinkArgs)]),
    "obgc_noalloc_max_cells": (C.c_longlong, []),
    "obgc_dms_source_sink_noalloc": (C.c_int, [C.c_void_p, C.POINTER(DmsSourceSinkArgs)]),
    "obgc_dms_surface_fluxes": (C.c_int, [C.c_void_p, C.POINTER(DmsSurfaceFluxesArgs)]),
    "obgc_macros_source_sink": (C.c_int, [C.c_void_p, C.POINTER(MacrosSourceSinkArgs)]),
    "obgc_apply_tendencies": (C.c_int, [C.c_void_p, C.POINTER(ApplyTendenciesArgs)]),
    "obgc_bgc_diag_accumulate": (C.c_int, [C.c_void_p, C.POINTER(Grid), C.POINTER(BgcDiag), C.POINTER(BgcDiag), C.c_double, C.c_int]),
    "obgc_dms_diag_accumulate": (C.c_int, [C.c_void_p, C.POINTER(Grid), C.POINTER(DmsDiag), C.POINTER(DmsDiag), C.c_double, C.c_int]),
    "obgc_macros_diag_accumulate": (C.c_int, [C.c_void_p, C.POINTER(Grid), C.POINTER(MacrosDiag), C.POINTER(MacrosDiag), C.c_double,
                                              C.c_int]),
    "obgc_accumulate_arrays": (C.c_int, [C.c_void_p, C.POINTER(Grid), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int,
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_double, C.c_int]),
    "obgc_params_write_text": (C.c_int, [C.c_char_p, C.POINTER(BgcParams), C.POINTER(DmsParams), C.POINTER(MacrosParams)]),
    "obgc_params_read_text": (C.c_int, [C.c_char_p, C.POINTER(BgcParams), C.POINTER(DmsParams), C.POINTER(MacrosParams),
                                        C.c_char_p, C.c_int]),
    "obgc_restart_write": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _dptr, _dptr, _dptr, _dptr]),
    "obgc_partition_columns": (C.c_int, [C.POINTER(C.c_int), C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_longlong)]),
    "obgc_restart_read_header": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "obgc_restart_read": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _dptr, _dptr, _dptr, _dptr]),
    "obgc_restart_write_grid": (C.c_int, [C.c_char_p, C.POINTER(Grid), _dptr, _dptr, _dptr, _dptr]),
    "obgc_restart_read_grid": (C.c_int, [C.c_char_p, C.POINTER(Grid), _dptr, _dptr, _dptr, _dptr]),
    "obgc_block_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "obgc_block_destroy": (C.c_int, [C.c_void_p]),
    "obgc_block_step": (C.c_int, [C.c_void_p, C.POINTER(BlockStepArgs)]),
    "obgc_block_step_device": (C.c_int, [C.c_void_p, C.POINTER(BlockStepArgs)]),
    "obgc_block_set_diagnostics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "obgc_block_get_state": (C.c_int, [C.c_void_p, _dptr, _dptr, _dptr, _dptr]),
    "obgc_block_set_state": (C.c_int, [C.c_void_p, _dptr, _dptr, _dptr, _dptr]),
    "obgc_diag_set_rows": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "obgc_diag_set_create": (C.c_int, [C.c_void_p, C.POINTER(Grid), C.c_int, C.POINTER(C.c_char_p), _dptr, _dptr, C.POINTER(C.c_void_p)]),
    "obgc_diag_set_destroy": (C.c_int, [C.c_void_p]),
    "obgc_diag_set_count": (C.c_int, [C.c_void_p]),
    "obgc_diag_set_bgc": (C.POINTER(BgcDiag), [C.c_void_p]),
    "obgc_diag_set_dms": (C.POINTER(DmsDiag), [C.c_void_p]),
    "obgc_diag_set_macros": (C.POINTER(MacrosDiag), [C.c_void_p]),
    "obgc_diag_set_accumulate": (C.c_int, [C.c_void_p, C.c_double, C.c_int]),
    "obgc_diag_set_elems": (C.c_longlong, [C.c_void_p, C.c_char_p]),
    "obgc_diag_set_download": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, _dptr, C.c_longlong]),
    "obgc_diag_set_write": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "obgc_diag_names_read_text": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "obgc_diag_file_info": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "obgc_diag_file_record": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.c_longlong)]),
    "obgc_diag_file_read": (C.c_int, [C.c_char_p, C.c_char_p, _dptr, C.c_longlong]),
    "obgc_synth_fill": (C.c_int, [C.c_void_p, C.POINTER(SynthArgs)]),
    "obgc_synth_fill_extra": (C.c_int, [C.c_void_p, C.POINTER(SynthExtraArgs)]),
}

_lib = None


def load():
    """Load libobgc.so and bind every prototype.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # PyTorch-ROCm wheels bundle their own libamdhip64; importing torch first makes libobgc.so
        # bind to that same runtime instead of pulling a second copy from /opt/rocm (two HIP
        # runtimes in one process leave torch without devices).  A Fortran host has no torch.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libobgc.so not found at %s -- build it with `make -C ocean-bgc_amd/csrc` "
                           "(or __graft_entry__.build()).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.obgc_abi_version() != 1:
        raise RuntimeError("libobgc ABI version mismatch")
    for fn, st in (("obgc_sizeof_bgc_params", BgcParams), ("obgc_sizeof_dms_params", DmsParams),
                   ("obgc_sizeof_macros_params", MacrosParams), ("obgc_sizeof_bgc_diag", BgcDiag)):
        if getattr(lib, fn)() != C.sizeof(st):
            raise RuntimeError("%s layout differs between obgc.h and _lib.py" % st.__name__)
    _lib = lib
    return lib
