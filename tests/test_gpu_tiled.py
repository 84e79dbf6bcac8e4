"""The column-block layout (OBGC_LAYOUT_TILED: blocks of 64 columns, every level-dependent array of a call a set of rows
of ONE arena) through the C ABI: against the golden output of the unmodified reference, bit for bit against the native
column-fastest layout (same kernels, other addresses), with column counts that do not fill the last block, for the
surface fluxes, the explicit update and the running means, and the entry points that refuse it."""
import os

import numpy as np
import pytest

import parity as P
from conftest import has_gpu
from oracle import fields as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def engines(pkg):
    if not has_gpu():
        pytest.fail("needs a GPU")
    from ocean_bgc_amd.modules import ColumnEngine
    return {p: ColumnEngine(0, path=p) for p in ("device_tiled", "device_colfast")}


def _args(syn, tr):
    return (tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"], syn["dust"],
            syn["sw"])


def _same(a, b, what):
    assert np.array_equal(a["tend"], b["tend"]), what + " tend"
    for grp in ("kc", "kca", "ca", "c"):
        for nm, v in a.get(grp, {}).items():
            assert np.array_equal(v, b[grp][nm]), "%s %s.%s" % (what, grp, nm)
    for nm in ("ph_prev", "ph_prev_alt"):
        if nm in a:
            assert np.array_equal(a[nm], b[nm]), what + " " + nm


@pytest.mark.parametrize("fixture", ["config1_1x60", "branch_cols_60", "branch_cols_80", "branch_cols_100"])
def test_bgc_source_sink_tiled_vs_reference_golden(engines, fixture):
    E = engines["device_tiled"]
    g = dict(np.load(os.path.join(GOLD, fixture + ".npz")))
    nc, nz = g["in.temp"].shape
    z = np.zeros((nc, nz))
    common = (g["in.temp"], g["in.salt"], g["in.zmid"], g["in.dz"], g["in.zbot"], g["in.lat"], g["in.kmax"], g["in.fesed"],
              g["in.dust"], g["in.sw"])
    a = E.BGC_SourceSink(g["in.bgc_tracers"], *common, z, z, alt_co2_use_eco=True, diag_init=-7.0)
    rep = P.compare_bgc(a, P.golden_bgc(g, "bgc_cold"))
    assert rep.ok(), "\n".join(rep.fail[:12])
    assert E.eng.last_schedule() in (3, 4)     # the column-block kernel, the carbonate solve inside or beside it
    for nm in F.BGC_DIAG_KC_UNTOUCHED:
        assert (a["kc"][nm] == -7.0).all(), nm
    below = np.arange(nz)[None, :] >= g["in.kmax"][:, None]
    assert (a["tend"][:, below] == 0).all() and (a["ph_prev"][below] == 0).all()
    b = E.BGC_SourceSink(g["in.bgc_tracers_warm"], *common, g["bgc_cold.ph_prev"],
                         g["bgc_cold.ph_prev_alt"] + 0.01 * (g["bgc_cold.ph_prev_alt"] != 0), alt_co2_use_eco=False, diag_init=-7.0)
    rep = P.compare_bgc(b, P.golden_bgc(g, "bgc_warm"))
    assert rep.ok(), "\n".join(rep.fail[:12])


@pytest.mark.parametrize("fixture", ["branch_cols_60", "branch_cols_100"])
def test_fused_step_tiled_vs_reference_golden(engines, fixture):
    E = engines["device_tiled"]
    g = dict(np.load(os.path.join(GOLD, fixture + ".npz")))
    nc, nz = g["in.temp"].shape
    z = np.zeros((nc, nz))
    common = (g["in.temp"], g["in.salt"], g["in.zmid"], g["in.dz"], g["in.zbot"], g["in.lat"], g["in.kmax"], g["in.fesed"],
              g["in.dust"], g["in.sw"])
    r = E.Fused_SourceSink(g["in.bgc_tracers"], *common, z, z, g["in.forcing.SST"], g["in.dms_tracers"], g["in.macros_tracers"],
                           diag_init=-7.0)
    rep = P.compare_bgc(r["bgc"], P.golden_bgc(g, "bgc_cold"))
    P.compare_simple(r["dms"], dict(tend=g["dms.tend"], kc={nm: g["dms.kc." + nm] for nm in F.DMS_DIAG_KC}), "dms", rep)
    P.compare_simple(r["macros"], dict(tend=g["macros.tend"], kc={nm: g["macros.kc." + nm] for nm in F.MACROS_DIAG_KC}), "macros", rep)
    assert rep.ok(), "\n".join(rep.fail[:12])


@pytest.mark.parametrize("nc,nz,diagnostics", [(20_000, 80, True), (4_999, 60, True), (130, 100, False), (1, 60, True), (63, 7, True)])
def test_tiled_equals_native_layout_bit_for_bit(engines, synth, nc, nz, diagnostics):
    """the same kernels on other addresses: every output of the fused step, ragged last block included"""
    syn = synth.generate(nc, nz, col0=250_000)
    tr, dtr, mtr = synth.bgc_tracer_array(syn), synth.dms_tracer_array(syn), synth.macros_tracer_array(syn)
    z = np.zeros((nc, nz))
    sst = syn["forcing"]["SST"]
    os.environ["OBGC_SCHED"] = "2"            # the native layout with the carbonate solve inside too: the same launch
    try:
        from ocean_bgc_amd.modules import ColumnEngine
        ref = ColumnEngine(0, path="device_colfast").Fused_SourceSink(*_args(syn, tr), z, z, sst, dtr, mtr, diag_init=-3.0,
                                                                      diagnostics=diagnostics)
    finally:
        del os.environ["OBGC_SCHED"]
    got = engines["device_tiled"].Fused_SourceSink(*_args(syn, tr), z, z, sst, dtr, mtr, diag_init=-3.0, diagnostics=diagnostics)
    _same(ref["bgc"], got["bgc"], "bgc")
    _same(ref["dms"], got["dms"], "dms")
    _same(ref["macros"], got["macros"], "macros")
    # and the default arrangement of the native layout (carbonate kernel beside the column kernel below 10 M cells)
    dflt = engines["device_colfast"].BGC_SourceSink(*_args(syn, tr), z, z, diag_init=-3.0, diagnostics=diagnostics)
    _same(dflt, got["bgc"], "bgc, default schedule")


def test_saturation_horizons_without_the_callers_co3_arrays(engines, synth):
    """zsatcalc / zsatarag requested, CO3 / co3_sat_* not: the native layout keeps them in scratch planes for the
    saturation-horizon kernel, the column-block layout has no scratch in the caller's index space and searches inside the
    column kernel (schedule 3) -- the same bits either way."""
    import torch
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import TileArena
    nc, nz = 700, 60
    syn = synth.generate(nc, nz, col0=31_000)
    tr = synth.bgc_tracer_array(syn)
    z = np.zeros((nc, nz))
    want = engines["device_colfast"].BGC_SourceSink(*_args(syn, tr), z, z)
    E = engines["device_tiled"].eng
    A = TileArena(torch, "cuda:0", nc, nz, 30 + 6 + 30 + 2 + 16)
    t = A.take(30).put(tr)
    f = {nm: A.take(1, (nc, nz)).put(syn[nm]) for nm in ("temp", "salt", "zmid", "dz", "zbot", "fesed")}
    tend, php, pha = A.take(30), A.take(1, (nc, nz)), A.take(1, (nc, nz))
    diag = {nm: A.take(4) for nm in ("photoC", "photoC_NO3", "CaCO3_form", "bSi_form")}
    for nm in ("zsatcalc", "zsatarag", "O2_ZMIN_DEPTH", "Jint_Ntot"):
        diag[nm] = torch.full((nc,), -5.0, dtype=torch.float64, device="cuda")
    cv = {nm: torch.from_numpy(np.ascontiguousarray(syn[nm])).cuda() for nm in ("lat", "dust", "sw")}
    kmax = torch.from_numpy(np.asarray(syn["kmax"], dtype=np.int32)).cuda()
    torch.cuda.synchronize()
    E.bgc_source_sink(nz, nc, L.OBGC_LAYOUT_TILED, t, f["temp"], f["salt"], f["zmid"], f["dz"], f["zbot"], cv["lat"], kmax, f["fesed"],
                      cv["dust"], cv["sw"], tend, php, pha, diag, True, ld=A.ld, plane=64)
    assert E.last_schedule() == 3
    assert np.array_equal(tend.get(), want["tend"]) and np.array_equal(php.get(), want["ph_prev"])
    for nm in ("zsatcalc", "zsatarag", "O2_ZMIN_DEPTH", "Jint_Ntot"):
        assert np.array_equal(diag[nm].cpu().numpy(), want["c"][nm]), nm
    assert np.array_equal(diag["photoC"].get(), want["kca"]["photoC"])


def test_surface_fluxes_update_and_means_in_the_tiled_layout(engines, synth):
    nc, nz = 3_000, 60
    syn = synth.generate(nc, nz, col0=7_000)
    tr, dtr = synth.bgc_tracer_array(syn), synth.dms_tracer_array(syn)
    f = syn["forcing"]
    flux = [syn[nm] for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")]
    T, C = engines["device_tiled"], engines["device_colfast"]
    a, b = T.BGC_SurfaceFluxes(tr, f, *flux), C.BGC_SurfaceFluxes(tr, f, *flux)
    for nm in ("netFlux", "gasFlux", "depositionFlux"):
        assert np.array_equal(a[nm], b[nm]), nm
    for nm, v in a["fdiag"].items():
        assert np.array_equal(v, b["fdiag"][nm]), nm
    a, b = T.DMS_SurfaceFluxes(dtr, f), C.DMS_SurfaceFluxes(dtr, f)
    assert np.array_equal(a["netFlux"], b["netFlux"])
    # explicit update of resident tracers (with the surface flux and the clamp)
    rng = np.random.default_rng(5)
    tend = rng.standard_normal(tr.shape) * 1e-6
    nf = rng.standard_normal((30, nc)) * 1e-4
    for n_use, c_use in ((nc, None), (nc - 1, None)):          # an odd column count takes the 8-byte kernel
        sl = slice(0, n_use)
        x1 = T.apply_tendencies(tr[:, sl], tend[:, sl], syn["kmax"][sl], 3600.0, nf[:, sl].copy(), syn["dz"][sl], True)
        x2 = C.apply_tendencies(tr[:, sl], tend[:, sl], syn["kmax"][sl], 3600.0, nf[:, sl].copy(), syn["dz"][sl], True)
        assert np.array_equal(x1, x2)


def test_diag_accumulate_in_the_tiled_layout(engines):
    import torch
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import TileArena
    nc, nz = 200, 9
    E = engines["device_tiled"].eng
    A = TileArena(torch, "cuda:0", nc, nz, 16)
    rng = np.random.default_rng(3)
    names = ["CO3", "photoC", "NITRIF"]
    vals = {"CO3": rng.standard_normal((nc, nz)), "photoC": rng.standard_normal((4, nc, nz)), "NITRIF": rng.standard_normal((nc, nz))}
    inst = {nm: A.take(4 if nm == "photoC" else 1, vals[nm].shape).put(vals[nm]) for nm in names}
    mean = {nm: A.take(4 if nm == "photoC" else 1, vals[nm].shape).put(np.full(vals[nm].shape, 2.0)) for nm in names}
    cvec = torch.from_numpy(rng.standard_normal(nc)).cuda()
    cmean = torch.full((nc,), 2.0, dtype=torch.float64, device="cuda")
    inst["zsatcalc"], mean["zsatcalc"] = cvec, cmean
    torch.cuda.synchronize()
    E.bgc_diag_accumulate(nz, nc, L.OBGC_LAYOUT_TILED, inst, mean, 0.25, reset=False, ld=A.ld, plane=64)
    for nm in names:
        assert np.array_equal(mean[nm].get(), 2.0 + 0.25 * vals[nm]), nm
    assert np.array_equal(cmean.cpu().numpy(), 2.0 + 0.25 * cvec.cpu().numpy())
    # nothing outside the selected rows moved (the arena's other rows are zero, the pad lanes of the last block too)
    assert float(A.buf[:, :, 12:, :].abs().sum()) == 0.0
    assert float(A.buf[-1, :, :, nc % 64:].abs().sum()) == 0.0


def test_synth_fill_tiled_is_the_numpy_generator(engines, synth):
    import torch
    from ocean_bgc_amd.engine import TileArena
    nc, nz, col0 = 1_000, 60, 123_456
    E = engines["device_tiled"].eng
    A = TileArena(torch, "cuda:0", nc, nz, 44)
    tr = A.take(30)
    fld = {nm: A.take(1, (nc, nz)) for nm in ("temp", "salt", "zmid", "dz", "zbot", "fesedflux")}
    own = {nm: A.take(1, (nc, nz)) for nm in ("dms", "dmsp", "prot", "poly", "lip")}
    cvec = {nm: torch.zeros(nc, dtype=torch.float64, device="cuda") for nm in ("lat", "dust_flux_in", "sw_flux")}
    kmax = torch.zeros(nc, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    E.synth_fill(nz, nc, col0, ld=A.ld, plane=64, tiled=True, tracers=tr, kmax=kmax, **fld, **cvec)
    E.synth_fill_extra(nz, nc, col0, ld=A.ld, tiled=True, **own)
    syn = synth.generate(nc, nz, col0=col0)
    assert np.array_equal(tr.get(), synth.bgc_tracer_array(syn))
    for nm, key in (("temp", "temp"), ("salt", "salt"), ("zmid", "zmid"), ("dz", "dz"), ("zbot", "zbot"), ("fesedflux", "fesed")):
        assert np.array_equal(fld[nm].get(), syn[key]), nm
    assert np.array_equal(kmax.cpu().numpy(), np.asarray(syn["kmax"], dtype=np.int32))
    dtr, mtr = synth.dms_tracer_array(syn), synth.macros_tracer_array(syn)
    assert np.array_equal(own["dms"].get(), dtr[0]) and np.array_equal(own["dmsp"].get(), dtr[1])
    assert np.array_equal(own["prot"].get(), mtr[0]) and np.array_equal(own["lip"].get(), mtr[2])


def test_entry_points_that_refuse_the_tiled_layout(engines, synth):
    from ocean_bgc_amd.engine import ObgcError
    nc, nz = 70, 10
    syn = synth.generate(nc, nz, col0=0)
    T = engines["device_tiled"]
    with pytest.raises(ObgcError, match="UNSUPPORTED"):
        T.DMS_SourceSink(synth.dms_tracer_array(syn), syn["dz"], syn["kmax"], syn["forcing"]["SST"], syn["sw"])
    with pytest.raises(ObgcError, match="UNSUPPORTED"):
        T.MACROS_SourceSink(synth.macros_tracer_array(syn), syn["kmax"])
    # diagnostics without the four production profiles the column integrals read
    import torch
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import TileArena
    A = TileArena(torch, "cuda:0", nc, nz, 80)
    tr = A.take(30).put(synth.bgc_tracer_array(syn))
    f = {nm: A.take(1, (nc, nz)).put(syn[nm]) for nm in ("temp", "salt", "zmid", "dz", "zbot", "fesed")}
    tend, php, pha, co3 = A.take(30), A.take(1, (nc, nz)), A.take(1, (nc, nz)), A.take(1, (nc, nz))
    cv = {nm: torch.from_numpy(np.ascontiguousarray(syn[nm])).cuda() for nm in ("lat", "dust", "sw")}
    cv["kmax"] = torch.from_numpy(np.asarray(syn["kmax"], dtype=np.int32)).cuda()
    torch.cuda.synchronize()
    with pytest.raises(ObgcError, match="photoC"):
        T.eng.bgc_source_sink(nz, nc, L.OBGC_LAYOUT_TILED, tr, f["temp"], f["salt"], f["zmid"], f["dz"], f["zbot"], cv["lat"], cv["kmax"],
                              f["fesed"], cv["dust"], cv["sw"], tend, php, pha, {"CO3": co3}, True, ld=A.ld, plane=64)
    # a tiled grid without its level stride, or in host memory, is a bad grid
    with pytest.raises(ObgcError, match="INVALID"):
        T.eng.bgc_source_sink(nz, nc, L.OBGC_LAYOUT_TILED, tr, f["temp"], f["salt"], f["zmid"], f["dz"], f["zbot"], cv["lat"], cv["kmax"],
                              f["fesed"], cv["dust"], cv["sw"], tend, php, pha, None, True, ld=0, plane=64)
