"""Parameter text files and the restart file (host-only helpers of libobgc.so, no GPU)."""
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


def test_read_rederives_the_autotroph_slots_and_ignores_stale_ind_lines(L, tmp_path):
    """kSiO3 / imp_calcifier decide which autotroph owns a Si / CaCO3 pool (BGC_mod.F90:303-320): after a read the slots
    the kernels branch on follow from the table read, not from *_ind lines of the file."""
    lib, b, d, m = _defaults(L)
    assert b.autotrophs[1].Si_ind == 24 and b.autotrophs[0].Si_ind == 0 and b.autotrophs[0].CaCO3_ind == 20
    p = tmp_path / "si.txt"
    # small phytoplankton become silicifiers, diatoms stop; the file also carries a stale, contradicting Si_ind line
    p.write_text("[autotroph 1]\nkSiO3 = 0.5\nimp_calcifier = 0\nSi_ind = 0\n[autotroph 2]\nkSiO3 = 0.0\nSi_ind = 24\nCaCO3_ind = 20\n")
    err = C.create_string_buffer(256)
    assert lib.obgc_params_read_text(str(p).encode(), C.byref(b), None, None, err, 256) == 0, err.value
    assert b.autotrophs[0].Si_ind == 24 and b.autotrophs[1].Si_ind == 0
    assert b.autotrophs[0].CaCO3_ind == 0 and b.autotrophs[1].CaCO3_ind == 0


def test_long_lines_and_decimal_comma_locales(L, tmp_path):
    lib, b, d, m = _defaults(L)
    p = tmp_path / "long.txt"
    p.write_text("[bgc]\nparm_o2_min = 3.5   #" + "x" * 5000 + "\nparm_o2_min_delta = 2.25\n")
    err = C.create_string_buffer(256)
    assert lib.obgc_params_read_text(str(p).encode(), C.byref(b), None, None, err, 256) == 0, err.value
    assert b.parm_o2_min == 3.5 and b.parm_o2_min_delta == 2.25
    # a process locale with a decimal comma must neither be written nor expected
    import locale
    old = locale.setlocale(locale.LC_NUMERIC)
    try:
        for cand in ("de_DE.UTF-8", "fr_FR.UTF-8", "de_DE", "fr_FR"):
            try:
                locale.setlocale(locale.LC_NUMERIC, cand)
                break
            except locale.Error:
                continue
        else:
            pytest.skip("no decimal-comma locale installed")
        path = str(tmp_path / "loc.txt").encode()
        b.parm_o2_min = 0.1 + 0.2
        assert lib.obgc_params_write_text(path, C.byref(b), None, None) == 0
        assert "," not in open(path.decode()).read()
        lib2, b2, _, _ = _defaults(L)
        assert lib.obgc_params_read_text(path, C.byref(b2), None, None, err, 256) == 0, err.value
        assert b2.parm_o2_min == 0.1 + 0.2
    finally:
        locale.setlocale(locale.LC_NUMERIC, old)


def test_restart_file_in_the_native_layout_and_order_word(L, tmp_path):
    lib = L.load()
    nz, nc, ld = 5, 9, 12
    rng = np.random.default_rng(5)
    k_fast = [rng.standard_normal((nc, nz)), rng.standard_normal((nc, nz))]       # X(k,column): reference order
    sp = [rng.standard_normal(nc), rng.standard_normal(nc)]
    ptr = lambda x: x.ctypes.data_as(C.c_void_p)
    # the same state held column-fastest with a padded leading dimension
    col_fast = [np.full((nz, ld), np.nan) for _ in range(2)]
    for a, bb in zip(col_fast, k_fast):
        a[:, :nc] = bb.T
    g = L.Grid()
    g.nlev, g.ncol, g.layout, g.memspace, g.ld = nz, nc, L.OBGC_LAYOUT_COLFAST, L.OBGC_MEM_HOST, ld
    p1, p2 = str(tmp_path / "a.rst").encode(), str(tmp_path / "b.rst").encode()
    assert lib.obgc_restart_write(p1, nz, nc, ptr(k_fast[0]), ptr(k_fast[1]), ptr(sp[0]), ptr(sp[1])) == 0
    assert lib.obgc_restart_write_grid(p2, C.byref(g), ptr(col_fast[0]), ptr(col_fast[1]), ptr(sp[0]), ptr(sp[1])) == 0
    assert open(p1.decode(), "rb").read() == open(p2.decode(), "rb").read()          # one file format
    back = [np.full((nz, ld), -1.0) for _ in range(2)]
    s2 = [np.zeros(nc), np.zeros(nc)]
    assert lib.obgc_restart_read_grid(p1, C.byref(g), ptr(back[0]), ptr(back[1]), ptr(s2[0]), ptr(s2[1])) == 0
    for a, bb in zip(back, k_fast):
        assert np.array_equal(a[:, :nc], bb.T) and (a[:, nc:] == -1.0).all()
    assert np.array_equal(s2[0], sp[0]) and np.array_equal(s2[1], sp[1])
    g.memspace = L.OBGC_MEM_DEVICE                                                   # host arrays only
    assert lib.obgc_restart_read_grid(p1, C.byref(g), ptr(back[0]), ptr(back[1]), ptr(s2[0]), ptr(s2[1])) == -1
    # an order word other than 0 (level-fastest) is not defined: rejected
    raw = bytearray(open(p1.decode(), "rb").read())
    raw[16] = 1
    open(p1.decode(), "wb").write(bytes(raw))
    fl, fc = C.c_int(), C.c_int()
    assert lib.obgc_restart_read_header(p1, C.byref(fl), C.byref(fc)) == -1


def test_diagnostics_section_names_and_file_helpers(L, tmp_path):
    """[diagnostics] of a parameter file: names are checked by obgc_params_read_text and returned, qualified, by
    obgc_diag_names_read_text; a diagnostics file laid out as csrc/obgc_io.cpp documents is read by the host-only helpers."""
    import ctypes as C
    import struct
    import numpy as np
    from ocean_bgc_amd.engine import DiagSet, ObgcError, diag_file_read, diag_names_from_file
    lib = L.load()
    p = str(tmp_path / "p.txt")
    open(p, "w").write("[bgc]\nparm_Fe_bioavail = 0.5\n[diagnostics]\nbgc = CO3, diag_pH_3D , photoC\ndms = DMS_R_TOTAL\nbgc=Jint_Ctot\n")
    assert diag_names_from_file(p) == ["bgc.CO3", "bgc.pH_3D", "bgc.photoC", "dms.DMS_R_TOTAL", "bgc.Jint_Ctot"]
    b = L.BgcParams()
    lib.obgc_bgc_params_init(C.byref(b), None, 273.15)
    err = C.create_string_buffer(256)
    assert lib.obgc_params_read_text(p.encode(), C.byref(b), None, None, err, 256) == 0 and b.parm_Fe_bioavail == 0.5
    open(p, "w").write("[diagnostics]\nbgc = CO3, no_such_member\n")
    assert lib.obgc_params_read_text(p.encode(), C.byref(b), None, None, err, 256) != 0 and b"no_such_member" in err.value
    with pytest.raises(ObgcError, match="no_such_member"):
        diag_names_from_file(p)
    assert DiagSet.rows(["CO3", "photoC", "bgc.Jint_Ctot", "photoC_zint", "dms.DMS_R_TOTAL"]) == (6, 5)
    with pytest.raises(ObgcError, match="unknown or ambiguous"):
        DiagSet.rows(["macros.CO3"])
    nz, nc = 3, 2
    a = np.arange(24, dtype=np.float64).reshape(4, nc, nz)
    path = str(tmp_path / "hand.obgcdia")
    with open(path, "wb") as f:
        f.write(struct.pack("<8s4i10i", b"OBGCDIA1", nz, nc, 1, 0, *([0] * 10)))
        f.write(struct.pack("<40s4iq", b"bgc.photoC", 0, 1, 4, 0, a.size) + a.tobytes())
    hdr, rec = diag_file_read(path)
    assert hdr == dict(nlev=nz, ncol=nc, nrec=1, which=0) and np.array_equal(rec["bgc.photoC"], a)
