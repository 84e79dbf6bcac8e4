"""parameter text files and the restart file (host-only helpers of libobgc.so, no GPU)."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def L(pkg):
    from ocean_bgc_amd import _lib
    _lib.load()
    return _lib


def _defaults(L):
    lib = L.load()
    b, d, m = L.BgcParams(), L.DmsParams(), L.MacrosParams()
    lib.obgc_bgc_params_init(C.byref(b), None, 273.15)
    lib.obgc_dms_params_init(C.byref(d), None)
    lib.obgc_macros_params_init(C.byref(m), None)
    return lib, b, d, m


def test_parameter_file_round_trips_bit_for_bit(L, tmp_path):
    lib, b, d, m = _defaults(L)
    b.parm_o2_min = 4.0 + 1e-13          # not representable in a short decimal
    b.autotrophs[2].kFe = 0.1 / 3.0
    d.k_S_z = np.pi
    path = str(tmp_path / "params.txt").encode()
    assert lib.obgc_params_write_text(path, C.byref(b), C.byref(d), C.byref(m)) == 0
    lib2, b2, d2, m2 = _defaults(L)
    err = C.create_string_buffer(256)
    assert lib.obgc_params_read_text(path, C.byref(b2), C.byref(d2), C.byref(m2), err, 256) == 0, err.value
    assert bytes(b) == bytes(b2) and bytes(d) == bytes(d2) and bytes(m) == bytes(m2)
    text = open(path.decode()).read()
    assert "[autotroph 3]" in text and "parm_o2_min = 0x" in text and "[dms]" in text and "[macros]" in text


def test_partial_file_overrides_only_what_it_lists(L, tmp_path):
    lib, b, d, m = _defaults(L)
    ref = bytes(b)
    p = tmp_path / "edit.txt"
    p.write_text("# a host's tuning file\n[bgc]\nparm_POC_diss = 95.0   # cm -> see BGC_parms\n[autotroph 2]\nkSiO3 = 0.9\nNfixer = 0\n")
    err = C.create_string_buffer(256)
    assert lib.obgc_params_read_text(str(p).encode(), C.byref(b), None, None, err, 256) == 0, err.value
    assert b.parm_POC_diss == 95.0 and b.autotrophs[1].kSiO3 == 0.9
    b.parm_POC_diss, b.autotrophs[1].kSiO3 = L.BgcParams.from_buffer_copy(ref).parm_POC_diss, L.BgcParams.from_buffer_copy(ref).autotrophs[1].kSiO3
    assert bytes(b) == ref


@pytest.mark.parametrize("text,needle", [("[bgc]\nparm_bogus = 1\n", "parm_bogus"), ("[ocean]\n", "unknown section"),
                                         ("parm_o2_min = 1\n", "before any section"), ("[bgc]\nparm_o2_min = abc\n", "malformed"),
                                         ("[autotroph 7]\n", "unknown section")])
def test_malformed_files_are_errors_with_a_line_number(L, tmp_path, text, needle):
    lib, b, d, m = _defaults(L)
    p = tmp_path / "bad.txt"
    p.write_text(text)
    err = C.create_string_buffer(256)
    assert lib.obgc_params_read_text(str(p).encode(), C.byref(b), C.byref(d), C.byref(m), err, 256) == -1
    assert needle.encode() in err.value and b"bad.txt:" in err.value


def test_restart_file_round_trip_and_validation(L, tmp_path):
    lib = L.load()
    nz, nc = 7, 13
    rng = np.random.default_rng(3)
    a = [rng.standard_normal((nc, nz)), rng.standard_normal((nc, nz)), rng.standard_normal(nc), rng.standard_normal(nc)]
    path = str(tmp_path / "ph.rst").encode()
    ptr = lambda x: x.ctypes.data_as(C.c_void_p)
    assert lib.obgc_restart_write(path, nz, nc, *[ptr(x) for x in a]) == 0
    import os
    assert os.path.getsize(path) == 64 + 8 * (2 * nz * nc + 2 * nc)
    fl, fc = C.c_int(), C.c_int()
    assert lib.obgc_restart_read_header(path, C.byref(fl), C.byref(fc)) == 0 and (fl.value, fc.value) == (nz, nc)
    b = [np.zeros_like(x) for x in a]
    assert lib.obgc_restart_read(path, nz, nc, *[ptr(x) for x in b]) == 0
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert lib.obgc_restart_read(path, nz + 1, nc, *[ptr(x) for x in b]) == -1      # shape mismatch
    with open(path, "r+b") as f:
        f.write(b"NOTOBGC1")
    assert lib.obgc_restart_read_header(path, C.byref(fl), C.byref(fc)) == -1        # bad magic
