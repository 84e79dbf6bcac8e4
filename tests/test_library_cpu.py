"""CPU-only checks of the product side: the C-ABI library loads, exports every symbol
include/obgc.h declares, refuses to run without a GPU (no fallback), and its host-side
parameter / metadata tables equal the reference's (golden dump of BGC_parms_init,
DMS_parms_init, MACROS_parms_init and the *_init name tables)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, has_gpu
from oracle import fields as F

GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def L(pkg):
    from ocean_bgc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "ocean-bgc_amd", "csrc")])
    return _lib


def test_header_symbols_are_exported(L):
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "obgc.h")).read()
    declared = set(re.findall(r"\b(obgc_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"obgc_handle"}
    assert declared == set(L.PROTOTYPES), declared ^ set(L.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.obgc_abi_version() == 1


def test_field_lists_agree_with_oracle_lists(L):
    pairs = [(L.BGC_TRACERS, F.BGC_TRACERS), (L.BGC_DIAG_KC, F.BGC_DIAG_KC), (L.BGC_DIAG_KCA, F.BGC_DIAG_KCA),
             (L.BGC_DIAG_CA, F.BGC_DIAG_CA), (L.BGC_DIAG_C, F.BGC_DIAG_C), (L.BGC_FLUX_DIAG, F.BGC_FLUX_DIAG),
             (L.DMS_TRACERS, F.DMS_TRACERS), (L.DMS_DIAG_KC, F.DMS_DIAG_KC), (L.DMS_FLUX_DIAG, F.DMS_FLUX_DIAG),
             (L.DMS_PARMS, F.DMS_PARMS), (L.MACROS_TRACERS, F.MACROS_TRACERS), (L.MACROS_DIAG_KC, F.MACROS_DIAG_KC),
             (L.MACROS_PARMS, F.MACROS_PARMS), (L.BGC_PARMS, F.BGC_PARMS), (L.AUTOTROPH_INTS, F.AUTOTROPH_INTS),
             (L.AUTOTROPH_REALS, F.AUTOTROPH_REALS)]
    for a, b in pairs:
        assert a == b
    assert len(L.BGC_DIAG_KC) == 61 and len(L.BGC_DIAG_KCA) == 18 and len(L.BGC_DIAG_C) == 17


def test_oracle_header_lists_agree_with_fields_py():
    hdr = open(os.path.join(ROOT, "oracle", "obgc_oracle.h")).read()

    def xl(macro):
        m = re.search(r"#define\s+%s\(X\)(.*?)\n(?!\s+X)" % macro, hdr, re.S)
        return re.findall(r"X\((\w+)\)", m.group(1))

    assert xl("ORA_DIAG_KC_LIST") == F.BGC_DIAG_KC
    assert xl("ORA_DIAG_KCA_LIST") == F.BGC_DIAG_KCA
    assert xl("ORA_DIAG_C_LIST") == F.BGC_DIAG_C
    assert xl("ORA_DMS_DIAG_KC_LIST") == F.DMS_DIAG_KC
    assert xl("ORA_SURF_FORCING_LIST") == F.BGC_SURF_FORCING


def test_parameter_tables_equal_reference_dump(L):
    lib = L.load()
    g = dict(np.load(os.path.join(GOLD, "params.npz")))
    p = L.BgcParams()
    lib.obgc_bgc_params_init(C.byref(p), None, 273.15)
    areal = np.array([[getattr(p.autotrophs[a], n) for n in L.AUTOTROPH_REALS] for a in range(4)])
    aint = np.array([[getattr(p.autotrophs[a], n) for n in L.AUTOTROPH_INTS] for a in range(4)])
    assert np.array_equal(areal, g["areal"])
    assert np.array_equal(aint, g["aint"])          # includes BGC_init's derived Chl/C/Fe/Si/CaCO3 slots
    assert np.array_equal(np.array([getattr(p, n) for n in L.BGC_PARMS]), g["parms"])
    assert np.array_equal(np.array(p.parm_scalelen_z[:]), g["scalelen_z"])
    assert np.array_equal(np.array(p.parm_scalelen_vals[:]), g["scalelen_vals"])
    d = L.DmsParams()
    lib.obgc_dms_params_init(C.byref(d), None)
    assert np.array_equal(np.array([getattr(d, n) for n in L.DMS_PARMS]), g["dms_parms"])
    m = L.MacrosParams()
    lib.obgc_macros_params_init(C.byref(m), None)
    assert np.array_equal(np.array([getattr(m, n) for n in L.MACROS_PARMS]), g["macros_parms"])
    # permuted slots flow into the derived autotroph indices
    perm = (C.c_int * 30)(*range(30, 0, -1))
    lib.obgc_bgc_params_init(C.byref(p), perm, 273.15)
    assert p.autotrophs[0].C_ind == 31 - 17 and p.autotrophs[1].Si_ind == 31 - 24 and p.autotrophs[2].Si_ind == 0


def test_tracer_metadata_equals_reference_init(L):
    lib = L.load()
    g = dict(np.load(os.path.join(GOLD, "params.npz")))
    for module, key, n in ((0, "bgc_names", 30), (1, "dms_names", 14), (2, "macros_names", 8)):
        for i in range(n):
            for w in range(3):
                got = lib.obgc_tracer_name(module, i + 1, w).decode()
                assert got.strip() == str(g[key][i][w]).strip(), (module, i, w, got)
    assert lib.obgc_tracer_name(0, 31, 0) is None


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a machine without a GPU")
def test_no_gpu_means_loud_failure_not_fallback(L, pkg):
    from ocean_bgc_amd.engine import Engine, ObgcError
    with pytest.raises(ObgcError, match="no CPU fallback"):
        Engine(0)
    lib = L.load()
    h = C.c_void_p()
    assert lib.obgc_create(C.byref(h), 0, 0) == -2 and not h.value      # OBGC_ERR_NO_DEVICE


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under ocean-bgc_amd/ or include/ may reference it."""
    bad = []
    for base in ("ocean-bgc_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dp or "__pycache__" in dp:
                continue
            for fn in files:
                if fn.endswith((".so", ".o", ".mod", ".pyc")):
                    continue
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|oracle/|liboracle|libbgc_ref|obgc_oracle", txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_partition_columns(L):
    """obgc_partition_columns: contiguous ranges that tile the columns; with balance=cells no
    range exceeds the mean number of active cells by more than one column's worth."""
    from ocean_bgc_amd.engine import ObgcError, partition_columns
    rng = np.random.default_rng(3)
    # deep basin, a wide shelf, land: equal column counts would give the first device 20x the cells of the last
    kmax = np.concatenate([np.full(1000, 60), rng.integers(1, 6, 5000), np.zeros(700, dtype=int), np.full(300, -1)]).astype(np.int32)
    cells = np.maximum(kmax, 0)
    for nparts in (1, 2, 3, 8, 64):
        b = partition_columns(kmax, nparts, "columns")
        assert b[0] == 0 and b[-1] == kmax.size and all(y >= x for x, y in zip(b, b[1:]))
        assert max(y - x for x, y in zip(b, b[1:])) - min(y - x for x, y in zip(b, b[1:])) <= 1
        b = partition_columns(kmax, nparts, "cells")
        assert b[0] == 0 and b[-1] == kmax.size and all(y >= x for x, y in zip(b, b[1:]))
        load = [int(cells[x:y].sum()) for x, y in zip(b, b[1:])]
        assert sum(load) == int(cells.sum())
        assert max(load) <= cells.sum() / nparts + cells.max()
    # the cut is a pure function of the prefix sums: any split glued back gives every column exactly once
    b = partition_columns(kmax, 5, "cells")
    assert np.array_equal(np.concatenate([np.arange(x, y) for x, y in zip(b, b[1:])]), np.arange(kmax.size))
    # all-land and empty inputs fall back to equal counts
    assert partition_columns(np.zeros(10, dtype=np.int32), 4, "cells") == [0, 2, 5, 7, 10]
    assert partition_columns(np.zeros(0, dtype=np.int32), 3, "cells") == [0, 0, 0, 0]
    with pytest.raises(ObgcError):
        partition_columns(kmax, 0, "cells")


def test_fortran_binding_covers_the_resident_and_fused_entry_points():
    """The generated ISO_C_BINDING module must declare, with the C struct's member order, every entry point the Fortran
    host with resident tracers uses (no GPU needed: text of the generated module against the header)."""
    import re
    api = open(os.path.join(ROOT, "ocean-bgc_amd", "fortran", "obgc_c_api.F90")).read()
    hdr = open(os.path.join(ROOT, "include", "obgc.h")).read()
    for fn in ("obgc_fused_step", "obgc_synth_fill", "obgc_synth_fill_extra", "obgc_apply_tendencies", "obgc_last_schedule",
               "obgc_last_kernel_ms", "obgc_bgc_surface_fluxes", "obgc_bgc_source_sink", "obgc_dms_source_sink_noalloc"):
        assert re.search(r"bind\(C, name='%s'\)" % fn, api), fn
        assert re.search(r"\b%s\s*\(" % fn, hdr), fn
    # member order of the fused-step block: C header vs Fortran type (pointers only, so the order is the layout)
    c_body = re.search(r"typedef struct \{([^}]*)\} obgc_fused_step_args;", hdr).group(1)
    c_names = re.findall(r"\*\s*(\w+)", re.sub(r"/\*.*?\*/", "", c_body, flags=re.S))
    f_body = re.search(r"type, bind\(C\) :: obgc_fused_step_args(.*?)end type", api, re.S).group(1).replace("&\n", " ")
    f_names = [n.strip() for ln in f_body.splitlines() if "type(c_ptr)" in ln for n in ln.split("::")[1].split(",")]
    assert c_names == f_names, (c_names, f_names)
    dev = open(os.path.join(ROOT, "ocean-bgc_amd", "fortran", "obgc_device.F90")).read()
    for name in ("hipMalloc", "hipFree", "hipMemcpy", "hipMemset"):
        assert "bind(C, name='%s')" % name in dev


def test_build_guards_the_column_kernels_register_budget(tmp_path):
    """csrc/check_resources.py fails the build when a bgc_column_kernel instantiation needs scratch, more than 256 registers
    or more LDS than two workgroups per CU allow (the budget hangs on an internal compiler switch, csrc/Makefile)."""
    import subprocess
    import sys
    chk = os.path.join(ROOT, "ocean-bgc_amd", "csrc", "check_resources.py")
    rem = os.path.join(ROOT, "ocean-bgc_amd", "csrc", "build", "obgc_ecosys.remarks")
    if not os.path.exists(rem):
        pytest.skip("the library was not built here")
    ok = subprocess.run([sys.executable, chk, rem], capture_output=True, text=True)
    assert ok.returncode == 0 and "inside the budget" in ok.stdout, ok.stderr
    text = open(rem, errors="replace").read()
    for doctored in (re.sub(r"ScratchSize \[bytes/lane\]: 0", "ScratchSize [bytes/lane]: 152", text),
                     re.sub(r"VGPRs: 2\d\d", "VGPRs: 300", text), "no kernels here"):
        p = tmp_path / "bad.remarks"
        p.write_text(doctored)
        bad = subprocess.run([sys.executable, chk, str(p)], capture_output=True, text=True)
        assert bad.returncode != 0 and "budget" in bad.stderr + bad.stdout
