"""The batched MPAS-style host adapter (ocean-bgc_amd/fortran/obgc_mpas_adapter.F90).

The adapter only uses the reference's Fortran module API, so the SAME source is linked twice:
  * oracle/_ref/test_mpas_adapter_ref   against the unmodified reference objects (CPU, built by `make -C oracle
    ref_adapter`) -- checks the adapter's packing / unit conversions / state persistence against an independent
    numpy restatement of the same conversions fed to the C oracle: must agree BIT FOR BIT;
  * ocean-bgc_amd/fortran/build/test_mpas_adapter   against the drop-in modules + libobgc.so (GPU) -- parity bars
    of tests/parity.py, and the batched call must equal the historical one-call-per-cell loop bit for bit.
Two steps are run so that PH_PREV_* / surface_pH* persistence (warm brackets on step 2) is exercised.
"""
import os
import subprocess

import numpy as np
import pytest

import parity as P
from conftest import ROOT, has_gpu

FDIR = os.path.join(ROOT, "ocean-bgc_amd", "fortran")
EXE_GPU = os.path.join(FDIR, "build", "test_mpas_adapter")
EXE_REF = os.path.join(ROOT, "oracle", "_ref", "test_mpas_adapter_ref")

M_TO_CM, CM_TO_M = 100.0, 0.01
PA_TO_ATM = 1.0 / 101325.0
RAD_TO_DEG = 180.0 / 3.14159265358979323846


def mpas_block(synth, nc, nz, col0):
    """A block in the ocean model's conventions (metres, radians, Pa, tracer-fastest) derived from the synthetic
    generator, plus the package-side arrays the adapter is expected to build from it."""
    syn = synth.generate(nc, nz, col0=col0)
    tr = synth.bgc_tracer_array(syn)                       # [n, c, k]
    f = syn["forcing"]
    m = dict(
        tracers=np.ascontiguousarray(tr.transpose(1, 2, 0)),   # [c, k, n]  == Fortran (n, k, c)
        h=syn["dz"] / 100.0, zmid=-syn["zmid"] / 100.0, temp=syn["temp"], salt=syn["salt"],
        lat=syn["lat"] / RAD_TO_DEG, kmax=syn["kmax"].astype(np.int32),
        sw=syn["sw"], patm=f["surfacePressure"] * 101325.0, ice=f["iceFraction"], wind=f["windSpeedSquared10m"] / 1.0e4,
        co2=f["atmCO2"], co2alt=f["atmCO2_ALT_CO2"], dust=syn["dust"] * 10.0, fesed=syn["fesed"],
        depo=syn["depositionFlux"].T.copy(), river=syn["riverFlux"].T.copy(), seaice=syn["seaIceFlux"].T.copy())   # [c, n]
    # what the adapter must hand to the package (same operations, same order, fp64)
    dz = m["h"] * M_TO_CM
    pk = dict(tr=tr, temp=m["temp"], salt=m["salt"], dz=dz, zmid=-m["zmid"] * M_TO_CM, zbot=np.cumsum(dz, axis=1),
              lat=m["lat"] * RAD_TO_DEG, kmax=m["kmax"], fesed=m["fesed"], dust=m["dust"] * 0.1, sw=m["sw"],
              forcing=dict(surfacePressure=m["patm"] * PA_TO_ATM, iceFraction=m["ice"].copy(),
                           windSpeedSquared10m=m["wind"] * M_TO_CM * M_TO_CM, atmCO2=m["co2"], atmCO2_ALT_CO2=m["co2alt"],
                           surface_pH=np.zeros(nc), surface_pH_alt_co2=np.zeros(nc), surfaceDepth=0.5 * m["h"][:, 0],
                           SST=m["temp"][:, 0].copy(), SSS=m["salt"][:, 0].copy()),
              depo=m["depo"].T.copy(), river=m["river"].T.copy(), seaice=m["seaice"].T.copy())
    return m, pk


def write_block(path, m, nc, nz):
    with open(path, "wb") as fh:
        np.array([nz, nc], dtype=np.int32).tofile(fh)
        for k in ("tracers", "h", "zmid", "temp", "salt", "lat"):
            np.ascontiguousarray(m[k], dtype=np.float64).tofile(fh)
        m["kmax"].tofile(fh)
        for k in ("sw", "patm", "ice", "wind", "co2", "co2alt", "dust", "fesed"):
            np.ascontiguousarray(m[k], dtype=np.float64).tofile(fh)
        for k in ("depo", "river", "seaice"):          # Fortran (nCells, nTracers) == C [n, c]
            np.ascontiguousarray(m[k].T, dtype=np.float64).tofile(fh)


def read_result(path, nc, nz):
    raw = np.fromfile(path, dtype=np.float64)
    pos = [0]

    def take(*shape):
        n = int(np.prod(shape))
        a = raw[pos[0]:pos[0] + n].reshape(shape)
        pos[0] += n
        return a

    out = dict(tend=take(nc, nz, 30), sflux=take(nc, 30), php=take(nc, nz), pha=take(nc, nz), sph=take(nc), spha=take(nc))
    assert pos[0] == raw.size
    return out


def expected(oracle, pk, m, nc, nz, nsteps):
    """numpy restatement of obgc_mpas_block_step on top of the C oracle."""
    php, pha = np.zeros((nc, nz)), np.zeros((nc, nz))
    frc = {k: np.array(v, dtype=np.float64) for k, v in pk["forcing"].items()}
    for _ in range(nsteps):
        frc["iceFraction"] = m["ice"].copy()                    # re-packed every step (clamped in place by the package)
        s = oracle.bgc_surface_fluxes(pk["tr"], frc, pk["depo"], pk["river"], np.zeros((30, nc)), pk["seaice"])
        frc["surface_pH"], frc["surface_pH_alt_co2"] = s["forcing"]["surface_pH"], s["forcing"]["surface_pH_alt_co2"]
        w = oracle.bgc_source_sink(pk["tr"], pk["temp"], pk["salt"], pk["zmid"], pk["dz"], pk["zbot"], pk["lat"], pk["kmax"],
                                   pk["fesed"], pk["dust"], pk["sw"], php, pha, alt_co2_use_eco=False)
        php, pha = w["ph_prev"], w["ph_prev_alt"]
    active = np.arange(nz)[None, :] < pk["kmax"][:, None]
    tend = np.where(active[:, :, None], m["h"][:, :, None] * w["tend"].transpose(1, 2, 0), 0.0)    # [c, k, n]
    return dict(tend=tend, sflux=s["netFlux"].T * CM_TO_M, php=php, pha=pha, sph=frc["surface_pH"],
                spha=frc["surface_pH_alt_co2"])


def run(exe, fin, fout, mode, nsteps):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:/opt/rocm/lib/llvm/lib:" + env.get("LD_LIBRARY_PATH", "")
    res = subprocess.run([exe, fin, fout, str(mode), str(nsteps)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "test_mpas_adapter OK" in res.stdout, res.stdout + res.stderr


def test_adapter_sources_are_generated():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_shim", os.path.join(FDIR, "gen_shim.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    for fn, lines in (("obgc_mpas_adapter.F90", g.mpas_adapter()), ("test_mpas_adapter.F90", g.mpas_test_program()),
                      ("obgc_mpas_resident.F90", g.mpas_resident())):
        assert open(os.path.join(FDIR, fn)).read() == "\n".join(g.wrap(lines)) + "\n", fn
    src = open(os.path.join(FDIR, "obgc_mpas_adapter.F90")).read()
    # only the reference's public module API
    assert "use obgc_c_api" not in src and "use BGC_parms" in src and "use BGC_mod" in src


def test_adapter_on_reference_modules_matches_numpy_restatement(tmp_path, synth, oracle):
    """CPU: the adapter linked against the UNMODIFIED reference reproduces the independent numpy + oracle
    restatement bit for bit (the C oracle is bit-identical to the reference), batched and per column."""
    if not os.path.exists(EXE_REF):
        pytest.skip("oracle/_ref/test_mpas_adapter_ref not built (make -C oracle ref_adapter)")
    nc, nz, nsteps = 96, 60, 2
    m, pk = mpas_block(synth, nc, nz, col0=977)
    fin = str(tmp_path / "blk.bin")
    write_block(fin, m, nc, nz)
    want = expected(oracle, pk, m, nc, nz, nsteps)
    for mode in (0, 1):
        fout = str(tmp_path / ("out%d.bin" % mode))
        run(EXE_REF, fin, fout, mode, nsteps)
        got = read_result(fout, nc, nz)
        for k in want:
            assert np.array_equal(got[k], want[k]), (mode, k, np.abs(got[k] - want[k]).max())


@pytest.mark.gpu
def test_adapter_on_gpu_batched_equals_per_cell_and_matches_oracle(tmp_path, synth, oracle):
    if not has_gpu():
        pytest.fail("needs a GPU")
    if not os.path.exists(EXE_GPU):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    nc, nz, nsteps = 640, 60, 2
    m, pk = mpas_block(synth, nc, nz, col0=120000)
    fin = str(tmp_path / "blk.bin")
    write_block(fin, m, nc, nz)
    res = {}
    for mode in (0, 1):
        fout = str(tmp_path / ("out%d.bin" % mode))
        run(EXE_GPU, fin, fout, mode, nsteps)
        res[mode] = read_result(fout, nc, nz)
    for k in res[0]:            # one batched call == the historical one-call-per-cell loop, bit for bit
        assert np.array_equal(res[0][k], res[1][k]), k
    # the device-resident twin (Fortran module obgc_mpas_resident over obgc_block_*): same tendencies and fluxes
    fout = str(tmp_path / "out2.bin")
    run(EXE_GPU, fin, fout, 2, nsteps)
    res2 = read_result(fout, nc, nz)
    assert np.array_equal(res2["tend"], res[0]["tend"]) and np.array_equal(res2["sflux"], res[0]["sflux"])
    want = expected(oracle, pk, m, nc, nz, nsteps)
    rep = P.Report()
    P.compare_tend(rep, "mpas.tend", res[0]["tend"].transpose(2, 0, 1), want["tend"].transpose(2, 0, 1))
    rep.note("diag", "surface flux", P.e_norm(res[0]["sflux"], want["sflux"]), P.TOL_DIAG)
    # warm second step: both sides stop the Newton iteration at |dx| < 1e-10, so pH agrees to the solver's own slop
    for k in ("php", "pha", "sph", "spha"):
        rep.note("ph", k, P.e_norm(res[0][k], want[k]), P.TOL_PH)
    assert rep.ok(), "\n".join(rep.fail[:10])


def block_args(L, m, nc, nz, tend, accumulate):
    """-> (obgc_block_step_args, arrays it points to, the surface-flux output array)"""
    import ctypes as C
    a = L.BlockStepArgs()
    keep = []

    def put(name, arr, dtype=np.float64):
        x = np.ascontiguousarray(arr, dtype=dtype)
        keep.append(x)
        setattr(a, name, x.ctypes.data_as(C.c_void_p))
        return x

    put("ecosysTracers", m["tracers"]); put("layerThickness", m["h"]); put("zMid", m["zmid"]); put("temperature", m["temp"])
    put("salinity", m["salt"]); put("maxLevelCell", m["kmax"], np.int32); put("latCell", m["lat"])
    put("shortWaveHeatFlux", m["sw"]); put("atmosphericPressure", m["patm"]); put("iceFraction", m["ice"])
    put("windSpeedSquared10m", m["wind"]); put("atmosphericCO2", m["co2"]); put("atmosphericCO2_ALT_CO2", m["co2alt"])
    put("dustFlux", m["dust"]); put("ironSedimentFlux", m["fesed"])
    for k, nm in (("depo", "depositionFlux"), ("river", "riverFlux"), ("seaice", "seaIceFlux")):
        put(nm, m[k].T)                                  # Fortran (nCells, nTracers)
    sflux = np.zeros((nc, 30))
    a.ecosysTracersTend = tend.ctypes.data_as(C.c_void_p)
    a.ecosysSurfaceFlux = sflux.ctypes.data_as(C.c_void_p)
    a.accumulate = 1 if accumulate else 0
    return a, keep, sflux


def block_step(L, eng, blk, m, nc, nz, tend, accumulate):
    import ctypes as C
    a, keep, sflux = block_args(L, m, nc, nz, tend, accumulate)
    rc = eng.lib.obgc_block_step(blk, C.byref(a))
    assert rc == 0, (rc, eng.lib.obgc_last_error(eng.h))
    return sflux


@pytest.mark.gpu
@pytest.mark.parametrize("chunks,threads", [(1, "1"), (3, "1"), (3, "0")])
def test_resident_block_equals_host_adapter_and_matches_oracle(tmp_path, synth, oracle, pkg, monkeypatch, chunks, threads):
    """The device-resident block (obgc_block_*): same bits as the Fortran adapter through the host path, parity bars
    against the numpy + oracle restatement, and the accumulate switch.  chunks = 3: the pipelined step (cell ranges
    of 256, 256 and 128 cells going up, through the kernels and down concurrently)."""
    import ctypes as C
    if not has_gpu():
        pytest.fail("needs a GPU")
    monkeypatch.setenv("OBGC_BLOCK_CHUNKS", str(chunks))
    monkeypatch.setenv("OBGC_BLOCK_MIN_CELLS", "0")
    monkeypatch.setenv("OBGC_HOST_THREADS", threads)
    if not os.path.exists(EXE_GPU):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine
    nc, nz, nsteps = 640, 60, 2
    m, pk = mpas_block(synth, nc, nz, col0=120000)
    E = Engine(0)
    blk = C.c_void_p()
    assert E.lib.obgc_block_create(E.h, nc, nz, C.byref(blk)) == 0
    tend = np.full((nc, nz, 30), 123.0)
    for _ in range(nsteps):
        tend[:] = 123.0                                   # overwritten, not accumulated
        sflux = block_step(L, E, blk, m, nc, nz, tend, accumulate=False)
    fin, fout = str(tmp_path / "blk.bin"), str(tmp_path / "out.bin")
    write_block(fin, m, nc, nz)
    run(EXE_GPU, fin, fout, 0, nsteps)
    host = read_result(fout, nc, nz)
    assert np.array_equal(tend, host["tend"]) and np.array_equal(sflux, host["sflux"])
    # the state the block carries = the arrays the host adapter keeps, in the model's order
    st = [np.full((nc, nz), -1.0), np.full((nc, nz), -1.0), np.full(nc, -1.0), np.full(nc, -1.0)]
    assert E.lib.obgc_block_get_state(blk, *[x.ctypes.data_as(C.c_void_p) for x in st]) == 0
    for got, key in zip(st, ("php", "pha", "sph", "spha")):
        assert np.array_equal(got, host[key]), key
    want = expected(oracle, pk, m, nc, nz, nsteps)
    rep = P.Report()
    P.compare_tend(rep, "block.tend", tend.transpose(2, 0, 1), want["tend"].transpose(2, 0, 1))
    rep.note("diag", "surface flux", P.e_norm(sflux, want["sflux"]), P.TOL_DIAG)
    assert rep.ok(), "\n".join(rep.fail[:10])
    # accumulate: a third step added on top of what the host already holds, active cells only
    base = np.full((nc, nz, 30), 5.0)
    acc = base.copy()
    block_step(L, E, blk, m, nc, nz, acc, accumulate=True)
    over = np.zeros((nc, nz, 30))
    blk2 = C.c_void_p()
    assert E.lib.obgc_block_create(E.h, nc, nz, C.byref(blk2)) == 0
    for _ in range(nsteps + 1):
        block_step(L, E, blk2, m, nc, nz, over, accumulate=False)
    active = (np.arange(nz)[None, :] < m["kmax"][:, None])[:, :, None]
    assert np.array_equal(acc, np.where(active, base + over, base))
    # restart: a fresh block given the state of two steps takes the third step of the uninterrupted block
    blk3 = C.c_void_p()
    assert E.lib.obgc_block_create(E.h, nc, nz, C.byref(blk3)) == 0
    assert E.lib.obgc_block_set_state(blk3, *[x.ctypes.data_as(C.c_void_p) for x in st]) == 0
    resumed = np.zeros((nc, nz, 30))
    sflux3 = block_step(L, E, blk3, m, nc, nz, resumed, accumulate=False)
    assert np.array_equal(resumed, over) and np.abs(sflux3).max() > 0
    # ... and ends in the state of the uninterrupted one (the warm brackets were the same), which a cold block does not
    def state_of(b):
        x = [np.empty((nc, nz)), np.empty((nc, nz)), np.empty(nc), np.empty(nc)]
        assert E.lib.obgc_block_get_state(b, *[y.ctypes.data_as(C.c_void_p) for y in x]) == 0
        return x
    for a3, a2 in zip(state_of(blk3), state_of(blk2)):
        assert np.array_equal(a3, a2)
    blk4 = C.c_void_p()
    assert E.lib.obgc_block_create(E.h, nc, nz, C.byref(blk4)) == 0
    block_step(L, E, blk4, m, nc, nz, np.zeros((nc, nz, 30)), accumulate=False)
    assert not np.array_equal(state_of(blk4)[0], state_of(blk2)[0])
    for b in (blk, blk2, blk3, blk4):
        assert E.lib.obgc_block_destroy(b) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("chunks", [1, 3])
def test_resident_block_with_dms_and_macros_in_the_fused_step(synth, oracle, pkg, monkeypatch, chunks):
    """A host that runs all three packages on host arrays: obgc_block_step with the own tracers of DMS and MACROS -- the
    source-sink part is ONE obgc_fused_step on the block, DMS_SurfaceFluxes follows BGC_SurfaceFluxes.  Same bits as the
    block's BGC-only step plus the separate DMS / MACROS entry points fed the package-side arrays through the host path, and
    inside the parity bars against the oracle; the accumulate switch covers all three tendency arrays."""
    import ctypes as C
    if not has_gpu():
        pytest.fail("needs a GPU")
    monkeypatch.setenv("OBGC_BLOCK_CHUNKS", str(chunks))
    monkeypatch.setenv("OBGC_BLOCK_MIN_CELLS", "0")
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine
    from ocean_bgc_amd.modules import ColumnEngine
    nc, nz = 640, 60
    m, pk = mpas_block(synth, nc, nz, col0=120000)
    syn = synth.generate(nc, nz, col0=120000)
    dtr, mtr = synth.dms_tracer_array(syn), synth.macros_tracer_array(syn)        # [14 | 8, c, k], shared slots = BGC tracers
    own_d = np.ascontiguousarray(dtr[0:2].transpose(1, 2, 0))                      # (2, k, c) tracer-fastest
    own_m = np.ascontiguousarray(mtr[0:3].transpose(1, 2, 0))
    E = Engine(0)
    blk, blk0 = C.c_void_p(), C.c_void_p()
    assert E.lib.obgc_block_create(E.h, nc, nz, C.byref(blk)) == 0 and E.lib.obgc_block_create(E.h, nc, nz, C.byref(blk0)) == 0

    def step(block, tend, dtend, mtend, accumulate, with_own=True):
        a, keep, sflux = block_args(L, m, nc, nz, tend, accumulate)
        dsf = np.full((nc, 2), -9.0)
        if with_own:
            a.dmsTracers = own_d.ctypes.data_as(C.c_void_p); a.dmsTracersTend = dtend.ctypes.data_as(C.c_void_p)
            a.dmsSurfaceFlux = dsf.ctypes.data_as(C.c_void_p)
            a.macrosTracers = own_m.ctypes.data_as(C.c_void_p); a.macrosTracersTend = mtend.ctypes.data_as(C.c_void_p)
        rc = E.lib.obgc_block_step(block, C.byref(a))
        assert rc == 0, (rc, E.lib.obgc_last_error(E.h))
        return sflux, dsf

    tend, dtend, mtend = np.full((nc, nz, 30), 123.0), np.full((nc, nz, 2), 123.0), np.full((nc, nz, 3), 123.0)
    sflux, dsf = step(blk, tend, dtend, mtend, False)
    # BGC: the bits of the BGC-only block step
    tend0 = np.full((nc, nz, 30), 123.0)
    sflux0, _ = step(blk0, tend0, None, None, False, with_own=False)
    assert np.array_equal(tend, tend0) and np.array_equal(sflux, sflux0)
    # DMS / MACROS: the bits of the separate entry points on the package-side arrays (host path), thickness-weighted
    H = ColumnEngine(0, path="host")
    d = H.DMS_SourceSink(dtr, pk["dz"], pk["kmax"], pk["forcing"]["SST"], pk["sw"])
    mm = H.MACROS_SourceSink(mtr, pk["kmax"])
    active = (np.arange(nz)[None, :] < pk["kmax"][:, None])[:, :, None]
    want_d = np.where(active, m["h"][:, :, None] * d["tend"][0:2].transpose(1, 2, 0), 0.0)
    want_m = np.where(active, m["h"][:, :, None] * mm["tend"][0:3].transpose(1, 2, 0), 0.0)
    assert np.array_equal(dtend, want_d) and np.array_equal(mtend, want_m)
    frc = {k: np.array(v, dtype=np.float64) for k, v in pk["forcing"].items()}
    ds = H.DMS_SurfaceFluxes(dtr, frc)
    assert np.array_equal(dsf, ds["netFlux"][0:2].T * CM_TO_M)
    # and the oracle, inside the bars
    od = oracle.dms_source_sink(dtr, pk["dz"], pk["kmax"], pk["forcing"]["SST"], pk["sw"])
    om = oracle.macros_source_sink(mtr, pk["kmax"])
    rep = P.Report()
    P.compare_tend(rep, "block.dms", dtend.transpose(2, 0, 1), np.where(active, m["h"][:, :, None] * od["tend"][0:2].transpose(1, 2, 0), 0.0).transpose(2, 0, 1))
    P.compare_tend(rep, "block.macros", mtend.transpose(2, 0, 1), np.where(active, m["h"][:, :, None] * om["tend"][0:3].transpose(1, 2, 0), 0.0).transpose(2, 0, 1))
    os_ = oracle.dms_surface_fluxes(dtr, frc)
    rep.note("diag", "dms surface flux", P.e_norm(dsf, os_["netFlux"][0:2].T * CM_TO_M), P.TOL_DIAG)
    assert rep.ok(), "\n".join(rep.fail[:10])
    # accumulate on top of what the host holds, active cells only, all three arrays
    a30, a2, a3 = np.full((nc, nz, 30), 5.0), np.full((nc, nz, 2), 5.0), np.full((nc, nz, 3), 5.0)
    step(blk0, a30, a2, a3, True)                                 # blk0 has run one step: this is its second (warm pH: BGC only differs)
    o30, o2, o3 = np.zeros((nc, nz, 30)), np.zeros((nc, nz, 2)), np.zeros((nc, nz, 3))
    step(blk, o30, o2, o3, False)                                 # blk's second step, overwriting
    assert np.array_equal(a30, np.where(active, 5.0 + o30, 5.0))
    assert np.array_equal(a2, np.where(active, 5.0 + o2, 5.0)) and np.array_equal(a3, np.where(active, 5.0 + o3, 5.0))
    assert E.lib.obgc_block_destroy(blk) == 0 and E.lib.obgc_block_destroy(blk0) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("with_own", [False, True])
def test_resident_block_device_pointers_same_bits_as_host_pointers(synth, pkg, monkeypatch, with_own):
    """A model whose arrays already live on the GPU: obgc_block_step_device on device copies of the model's arrays gives the
    bits of obgc_block_step on the host arrays -- overwrite, a warm second step that accumulates, with and without the own
    tracers of DMS and MACROS -- and leaves every input array as it was (the package's in-place Fe scaling of the flux
    arrays included)."""
    import ctypes as C
    import torch
    if not has_gpu():
        pytest.fail("needs a GPU")
    monkeypatch.setenv("OBGC_BLOCK_CHUNKS", "1")
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine
    nc, nz = 700, 60                                        # not a multiple of 64: ragged last workgroup of every kernel
    m, pk = mpas_block(synth, nc, nz, col0=120000)
    syn = synth.generate(nc, nz, col0=120000)
    own_d = np.ascontiguousarray(synth.dms_tracer_array(syn)[0:2].transpose(1, 2, 0))
    own_m = np.ascontiguousarray(synth.macros_tracer_array(syn)[0:3].transpose(1, 2, 0))
    E = Engine(0)
    blk_h, blk_d = C.c_void_p(), C.c_void_p()
    assert E.lib.obgc_block_create(E.h, nc, nz, C.byref(blk_h)) == 0 and E.lib.obgc_block_create(E.h, nc, nz, C.byref(blk_d)) == 0
    dev = torch.device("cuda:0")

    def host_step(tend, dtend, mtend, accumulate):
        a, keep, sflux = block_args(L, m, nc, nz, tend, accumulate)
        dsf = np.full((nc, 2), -9.0)
        if with_own:
            a.dmsTracers = own_d.ctypes.data_as(C.c_void_p); a.dmsTracersTend = dtend.ctypes.data_as(C.c_void_p)
            a.dmsSurfaceFlux = dsf.ctypes.data_as(C.c_void_p)
            a.macrosTracers = own_m.ctypes.data_as(C.c_void_p); a.macrosTracersTend = mtend.ctypes.data_as(C.c_void_p)
        rc = E.lib.obgc_block_step(blk_h, C.byref(a))
        assert rc == 0, (rc, E.lib.obgc_last_error(E.h))
        return sflux, dsf

    def device_step(tend, dtend, mtend, accumulate):
        a, keep, _ = block_args(L, m, nc, nz, tend, accumulate)
        names = [n for n, _t in L.BlockStepArgs._fields_ if n not in ("accumulate",)]
        host = {"ecosysTracersTend": tend, "ecosysSurfaceFlux": np.full((nc, 30), -9.0)}
        if with_own:
            host.update(dmsTracers=own_d, dmsTracersTend=dtend, dmsSurfaceFlux=np.full((nc, 2), -9.0), macrosTracers=own_m,
                        macrosTracersTend=mtend)
        # every array the struct points to, as a device tensor of the same bytes
        by_addr = {x.ctypes.data: x for x in keep}
        t = {}
        for n in names:
            if n in host:
                src = host[n]
            else:
                addr = getattr(a, n)
                if not addr:
                    continue
                src = by_addr[addr]
            t[n] = torch.from_numpy(src).to(dev)
            setattr(a, n, C.c_void_p(t[n].data_ptr()))
        before = {n: x.clone() for n, x in t.items()}
        torch.cuda.synchronize()
        rc = E.lib.obgc_block_step_device(blk_d, C.byref(a))
        assert rc == 0, (rc, E.lib.obgc_last_error(E.h))
        torch.cuda.synchronize()
        outs = ("ecosysTracersTend", "ecosysSurfaceFlux", "dmsTracersTend", "dmsSurfaceFlux", "macrosTracersTend")
        for n in t:
            if n not in outs:
                assert torch.equal(t[n], before[n]), n + " was modified"
        tend[:] = t["ecosysTracersTend"].cpu().numpy()
        if with_own:
            dtend[:] = t["dmsTracersTend"].cpu().numpy(); mtend[:] = t["macrosTracersTend"].cpu().numpy()
        return t["ecosysSurfaceFlux"].cpu().numpy(), (t["dmsSurfaceFlux"].cpu().numpy() if with_own else None)

    for accumulate, fill in ((False, 123.0), (True, 5.0)):
        th, dh, mh = np.full((nc, nz, 30), fill), np.full((nc, nz, 2), fill), np.full((nc, nz, 3), fill)
        td, dd, md = th.copy(), dh.copy(), mh.copy()
        sf_h, dsf_h = host_step(th, dh, mh, accumulate)
        sf_d, dsf_d = device_step(td, dd, md, accumulate)
        assert np.array_equal(th, td) and np.array_equal(sf_h, sf_d)
        assert np.abs(th).max() > 0 and not np.array_equal(th, np.full_like(th, fill))
        if with_own:
            assert np.array_equal(dh, dd) and np.array_equal(mh, md) and np.array_equal(dsf_h, dsf_d)
    # host arrays handed to the device entry point: an error code and a message, no kernel is launched on them
    a, keep, _ = block_args(L, m, nc, nz, np.zeros((nc, nz, 30)), False)
    assert E.lib.obgc_block_step_device(blk_d, C.byref(a)) != 0
    assert b"must be device memory" in E.lib.obgc_last_error(E.h)
    assert E.lib.obgc_block_destroy(blk_h) == 0 and E.lib.obgc_block_destroy(blk_d) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("chunks", [1, 3])
def test_resident_block_fills_a_diagnostic_set(synth, pkg, monkeypatch, chunks):
    """obgc_block_set_diagnostics: a diagnostic set on the block's geometry is filled by the block's steps -- BGC, DMS and
    MACROS members, level-dependent and per-column ones, over cell ranges of a pipelined step -- with the bits the host-memory
    entry points give for the package-side arrays; the running means follow; members outside the selection do not exist; the
    tendencies are those of a block without diagnostics."""
    import ctypes as C
    if not has_gpu():
        pytest.fail("needs a GPU")
    monkeypatch.setenv("OBGC_BLOCK_CHUNKS", str(chunks))
    monkeypatch.setenv("OBGC_BLOCK_MIN_CELLS", "0")
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine
    from ocean_bgc_amd.modules import ColumnEngine
    nc, nz = 640, 60
    m, pk = mpas_block(synth, nc, nz, col0=120000)
    syn = synth.generate(nc, nz, col0=120000)
    dtr, mtr = synth.dms_tracer_array(syn), synth.macros_tracer_array(syn)
    own_d = np.ascontiguousarray(dtr[0:2].transpose(1, 2, 0))
    own_m = np.ascontiguousarray(mtr[0:3].transpose(1, 2, 0))
    E = Engine(0)
    blk, plain = C.c_void_p(), C.c_void_p()
    assert E.lib.obgc_block_create(E.h, nc, nz, C.byref(blk)) == 0 and E.lib.obgc_block_create(E.h, nc, nz, C.byref(plain)) == 0
    names = ["CO3", "pH_3D", "POC_FLUX_IN", "photoC", "photoC_TOT_zint", "photoC_zint", "Jint_Ctot", "zsatcalc", "O2_ZMIN_DEPTH",
             "dms.DMS_R_TOTAL", "macros.LIP_R_TOTAL"]
    ds = E.diag_set(nz, nc, names)
    void = lambda p: C.cast(p, C.c_void_p) if p else None
    assert E.lib.obgc_block_set_diagnostics(blk, void(ds.bgc), void(ds.dms), void(ds.macros)) == 0

    def step(block, tend, dtend, mtend):
        a, keep, sflux = block_args(L, m, nc, nz, tend, False)
        dsf = np.zeros((nc, 2))
        a.dmsTracers = own_d.ctypes.data_as(C.c_void_p); a.dmsTracersTend = dtend.ctypes.data_as(C.c_void_p)
        a.dmsSurfaceFlux = dsf.ctypes.data_as(C.c_void_p)
        a.macrosTracers = own_m.ctypes.data_as(C.c_void_p); a.macrosTracersTend = mtend.ctypes.data_as(C.c_void_p)
        rc = E.lib.obgc_block_step(block, C.byref(a))
        assert rc == 0, (rc, E.lib.obgc_last_error(E.h))

    H = ColumnEngine(0, path="host")
    php, pha = np.zeros((nc, nz)), np.zeros((nc, nz))
    insts = []
    for i in range(2):
        t1, d1, m1 = np.zeros((nc, nz, 30)), np.zeros((nc, nz, 2)), np.zeros((nc, nz, 3))
        t0, d0, m0 = np.zeros((nc, nz, 30)), np.zeros((nc, nz, 2)), np.zeros((nc, nz, 3))
        step(blk, t1, d1, m1)
        step(plain, t0, d0, m0)
        assert np.array_equal(t1, t0) and np.array_equal(d1, d0) and np.array_equal(m1, m0)
        ds.accumulate(0.5, reset=(i == 0))
        # the package-side call of the same step through host memory (warm from its own previous pH, as the block's)
        w = H.BGC_SourceSink(pk["tr"], pk["temp"], pk["salt"], pk["zmid"], pk["dz"], pk["zbot"], pk["lat"], pk["kmax"], pk["fesed"],
                             pk["dust"], pk["sw"], php, pha, alt_co2_use_eco=False)
        php, pha = w["ph_prev"], w["ph_prev_alt"]
        d = H.DMS_SourceSink(dtr, pk["dz"], pk["kmax"], pk["forcing"]["SST"], pk["sw"])
        mm = H.MACROS_SourceSink(mtr, pk["kmax"])
        want = {"CO3": w["kc"]["CO3"], "pH_3D": w["kc"]["pH_3D"], "POC_FLUX_IN": w["kc"]["POC_FLUX_IN"], "photoC": w["kca"]["photoC"],
                "photoC_TOT_zint": w["c"]["photoC_TOT_zint"], "photoC_zint": w["ca"]["photoC_zint"], "Jint_Ctot": w["c"]["Jint_Ctot"],
                "zsatcalc": w["c"]["zsatcalc"], "O2_ZMIN_DEPTH": w["c"]["O2_ZMIN_DEPTH"], "dms.DMS_R_TOTAL": d["kc"]["DMS_R_TOTAL"],
                "macros.LIP_R_TOTAL": mm["kc"]["LIP_R_TOTAL"]}
        active = np.arange(nz)[None, :] < pk["kmax"][:, None]
        for n, ref in want.items():
            got = ds.download(n, "inst")
            if n.startswith(("dms.", "macros.")):            # below the sea floor the reference leaves these members alone
                got, ref = np.where(active, got, 0.0), np.where(active, ref, 0.0)
            assert np.array_equal(got, ref), (i, n)
        insts.append(want)
    for n in ("CO3", "photoC", "photoC_TOT_zint"):
        mean = ds.download(n, "mean")
        ref = 0.5 * insts[0][n] + 0.5 * insts[1][n]
        assert np.allclose(mean, ref, rtol=1e-15, atol=0.0), n
    assert ds.lib.obgc_diag_set_elems(ds.s, b"HCO3") < 0          # not selected: not there
    # detached again: the steps go on, the set keeps what it has
    assert E.lib.obgc_block_set_diagnostics(blk, None, None, None) == 0
    before = ds.download("CO3", "inst")
    step(blk, np.zeros((nc, nz, 30)), np.zeros((nc, nz, 2)), np.zeros((nc, nz, 3)))
    assert np.array_equal(ds.download("CO3", "inst"), before)
    ds.close()
    assert E.lib.obgc_block_destroy(blk) == 0 and E.lib.obgc_block_destroy(plain) == 0


@pytest.mark.gpu
def test_fortran_resident_block_with_diagnostics_and_state(tmp_path, synth, pkg):
    """The Fortran module obgc_mpas_resident with obgc_mpas_resident_select_diagnostics / _accumulate / _get_diagnostic /
    _get_state (test program mode 3): the means of the selected members over the steps and the carried pH state are the bits
    the Python host gets from the same block through the C ABI."""
    import ctypes as C
    if not has_gpu():
        pytest.fail("needs a GPU")
    if not os.path.exists(EXE_GPU):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine
    nc, nz, nsteps = 640, 60, 3
    m, pk = mpas_block(synth, nc, nz, col0=120000)
    fin, fout = str(tmp_path / "blk.bin"), str(tmp_path / "out3.bin")
    write_block(fin, m, nc, nz)
    run(EXE_GPU, fin, fout, 3, nsteps)
    raw = np.fromfile(fout, dtype=np.float64)
    got, pos = {}, 0
    for nm, shape in (("tend", (nc, nz, 30)), ("sflux", (nc, 30)), ("x0", (nc, nz)), ("x1", (nc, nz)), ("x2", (nc,)), ("x3", (nc,)),
                      ("co3", (nc, nz)), ("pcz", (nc,)), ("php", (nc, nz)), ("pha", (nc, nz)), ("sph", (nc,)), ("spha", (nc,))):
        n = int(np.prod(shape))
        got[nm] = raw[pos:pos + n].reshape(shape)
        pos += n
    assert pos == raw.size
    E = Engine(0)
    blk = C.c_void_p()
    assert E.lib.obgc_block_create(E.h, nc, nz, C.byref(blk)) == 0
    ds = E.diag_set(nz, nc, ["CO3", "photoC_TOT_zint", "dms.DMS_R_TOTAL"])
    void = lambda p: C.cast(p, C.c_void_p) if p else None
    assert E.lib.obgc_block_set_diagnostics(blk, void(ds.bgc), void(ds.dms), void(ds.macros)) == 0
    tend = np.zeros((nc, nz, 30))
    for s in range(nsteps):
        tend[:] = 0.0
        sflux = block_step(L, E, blk, m, nc, nz, tend, accumulate=True)      # the Fortran program accumulates into zeros
        ds.accumulate(1.0 / nsteps, reset=(s == 0))
    st = [np.empty((nc, nz)), np.empty((nc, nz)), np.empty(nc), np.empty(nc)]
    assert E.lib.obgc_block_get_state(blk, *[x.ctypes.data_as(C.c_void_p) for x in st]) == 0
    assert np.array_equal(got["tend"], tend) and np.array_equal(got["sflux"], sflux)
    assert np.array_equal(got["co3"], ds.download("CO3")) and np.array_equal(got["pcz"], ds.download("photoC_TOT_zint"))
    for key, ref in zip(("php", "pha", "sph", "spha"), st):
        assert np.array_equal(got[key], ref), key
    assert (got["co3"] > 0).any() and np.isfinite(got["co3"]).all()
    ds.close()
    assert E.lib.obgc_block_destroy(blk) == 0
