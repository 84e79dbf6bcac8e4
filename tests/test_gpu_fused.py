"""obgc_fused_step -- BGC_SourceSink + DMS_SourceSink + MACROS_SourceSink in one call (BASELINE config #4) -- through
the C ABI: against the golden output of the unmodified reference, bit for bit against the three separate entry points,
and on columns where DMS cannot share BGC's light pass (a pool the zero-mask removes while it still holds chlorophyll,
BGC_mod.F90:826-844 against DMS_mod.F90:539; a DMS PAR fraction other than BGC's)."""
import os

import numpy as np
import pytest

import parity as P
from conftest import has_gpu
from oracle import fields as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def E(pkg):
    if not has_gpu():
        pytest.fail("needs a GPU")
    from ocean_bgc_amd.modules import ColumnEngine
    return ColumnEngine(0, path="device_colfast")


def _args(syn, tr):
    return (tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"], syn["dust"],
            syn["sw"])


def _same(a, b, what):
    """two result dictionaries of one routine, bit for bit (NaN-free by construction)"""
    assert np.array_equal(a["tend"], b["tend"]), what + " tend"
    for grp in ("kc", "kca", "ca", "c"):
        for nm, v in a.get(grp, {}).items():
            assert np.array_equal(v, b[grp][nm]), "%s %s.%s" % (what, grp, nm)
    for nm in ("ph_prev", "ph_prev_alt"):
        if nm in a:
            assert np.array_equal(a[nm], b[nm]), what + " " + nm


def _separate(E, syn, tr, dtr, mtr, php, pha, fill, **kw):
    """the three entry points; tendency planes of the shared tracers zeroed like Fused_SourceSink returns them"""
    sst = syn["forcing"]["SST"]
    b = E.BGC_SourceSink(*_args(syn, tr), php, pha, diag_init=fill, **kw)
    d = E.DMS_SourceSink(dtr, syn["dz"], syn["kmax"], sst, syn["sw"], diag_init=fill)
    m = E.MACROS_SourceSink(mtr, syn["kmax"], diag_init=fill)
    return b, d, m


@pytest.mark.parametrize("fixture", ["config1_1x60", "branch_cols_60", "branch_cols_80", "branch_cols_100"])
def test_fused_step_vs_reference_golden(E, fixture):
    g = dict(np.load(os.path.join(GOLD, fixture + ".npz")))
    nc, nz = g["in.temp"].shape
    z = np.zeros((nc, nz))
    common = (g["in.temp"], g["in.salt"], g["in.zmid"], g["in.dz"], g["in.zbot"], g["in.lat"], g["in.kmax"], g["in.fesed"],
              g["in.dust"], g["in.sw"])
    r = E.Fused_SourceSink(g["in.bgc_tracers"], *common, z, z, g["in.forcing.SST"], g["in.dms_tracers"], g["in.macros_tracers"],
                           diag_init=-7.0)
    rep = P.compare_bgc(r["bgc"], P.golden_bgc(g, "bgc_cold"))
    # below the sea floor the reference leaves DMS / MACROS diagnostics alone (-7 in the fixture too) and zeroes tendencies
    P.compare_simple(r["dms"], dict(tend=g["dms.tend"], kc={nm: g["dms.kc." + nm] for nm in F.DMS_DIAG_KC}), "dms", rep)
    P.compare_simple(r["macros"], dict(tend=g["macros.tend"], kc={nm: g["macros.kc." + nm] for nm in F.MACROS_DIAG_KC}), "macros", rep)
    assert rep.ok(), "\n".join(rep.fail[:12])
    for nm in F.BGC_DIAG_KC_UNTOUCHED:
        assert (r["bgc"]["kc"][nm] == -7.0).all(), nm


@pytest.mark.parametrize("diagnostics", [True, False])
def test_fused_step_equals_the_three_calls_bit_for_bit(E, synth, diagnostics):
    nc, nz = 20_000, 80
    syn = synth.generate(nc, nz, col0=60_000)
    tr, dtr, mtr = synth.bgc_tracer_array(syn), synth.dms_tracer_array(syn), synth.macros_tracer_array(syn)
    z = np.zeros((nc, nz))
    sst = syn["forcing"]["SST"]
    f = E.Fused_SourceSink(*_args(syn, tr), z, z, sst, dtr, mtr, diag_init=-3.0, diagnostics=diagnostics)
    b, d, m = _separate(E, syn, tr, dtr, mtr, z, z, -3.0, diagnostics=diagnostics)
    _same(b, f["bgc"], "bgc cold")
    if diagnostics:
        _same(d, f["dms"], "dms")
        _same(m, f["macros"], "macros")
    else:
        assert np.array_equal(d["tend"], f["dms"]["tend"]) and np.array_equal(m["tend"], f["macros"]["tend"])
    # warm step on moved tracers, alt_co2_use_eco off
    tr2 = tr * 1.002
    dtr2, mtr2 = dtr.copy(), mtr.copy()
    for j, nm in enumerate(F.DMS_TRACERS[2:], start=2):
        dtr2[j] = tr2[F.BGC_TRACERS.index(nm)]
    for j, nm in enumerate(F.MACROS_TRACERS[3:], start=3):
        mtr2[j] = tr2[F.BGC_TRACERS.index(nm)]
    f2 = E.Fused_SourceSink(*_args(syn, tr2), b["ph_prev"], b["ph_prev_alt"], sst, dtr2, mtr2, alt_co2_use_eco=False, diag_init=-3.0,
                            diagnostics=diagnostics)
    b2 = E.BGC_SourceSink(*_args(syn, tr2), b["ph_prev"], b["ph_prev_alt"], alt_co2_use_eco=False, diag_init=-3.0, diagnostics=diagnostics)
    _same(b2, f2["bgc"], "bgc warm")
    assert np.array_equal(E.DMS_SourceSink(dtr2, syn["dz"], syn["kmax"], sst, syn["sw"])["tend"], f2["dms"]["tend"])
    # one module at a time
    only_d = E.Fused_SourceSink(*_args(syn, tr), z, z, sst, dtr, None, diag_init=-3.0, diagnostics=diagnostics)
    only_m = E.Fused_SourceSink(*_args(syn, tr), z, z, sst, None, mtr, diag_init=-3.0, diagnostics=diagnostics)
    assert "macros" not in only_d and "dms" not in only_m
    _same(f["dms"], only_d["dms"], "dms alone")
    _same(f["macros"], only_m["macros"], "macros alone")
    _same(f["bgc"], only_d["bgc"], "bgc beside dms")
    _same(f["bgc"], only_m["bgc"], "bgc beside macros")


def test_columns_where_dms_cannot_share_the_light_pass(E, oracle, synth):
    """A carbon / iron pool at zero removes its autotroph from BGC's chlorophyll total (zero-mask) but not from DMS's: such
    columns get a DMS light pass of their own, every other column shares BGC's."""
    nc, nz = 3_000, 60
    syn = synth.generate(nc, nz, col0=17)
    t = syn["tracers"]
    rng = np.random.RandomState(5)
    hit = rng.rand(nc, nz) < 0.002                      # a few cells in ~11 % of the columns
    hit[3::10, 1] = True                                 # and the second level of every tenth column, where there is light
    t["spFe"] = np.where(hit, 0.0, t["spFe"])            # sp pool masked out, spChl stays
    hit2 = rng.rand(nc, nz) < 0.001
    t["diatC"] = np.where(hit2, -1.0e-3, t["diatC"])     # negative carbon: clamped to 0 -> masked, diatChl stays
    flagged = (hit | hit2)[:, :].copy()
    flagged &= np.arange(nz)[None, :] < syn["kmax"][:, None]
    assert 0.02 < flagged.any(axis=1).mean() < 0.5
    tr, dtr, mtr = synth.bgc_tracer_array(syn), synth.dms_tracer_array(syn), synth.macros_tracer_array(syn)
    z = np.zeros((nc, nz))
    sst = syn["forcing"]["SST"]
    f = E.Fused_SourceSink(*_args(syn, tr), z, z, sst, dtr, mtr, diag_init=-3.0)
    b, d, m = _separate(E, syn, tr, dtr, mtr, z, z, -3.0)
    _same(b, f["bgc"], "bgc")
    _same(d, f["dms"], "dms")
    _same(m, f["macros"], "macros")
    rep = P.compare_simple(f["dms"], oracle.dms_source_sink(dtr, syn["dz"], syn["kmax"], sst, syn["sw"], diag_init=-3.0), "dms")
    P.compare_bgc(f["bgc"], oracle.bgc_source_sink(*_args(syn, tr), z, z, diag_init=-3.0), rep)
    assert rep.ok(), "\n".join(rep.fail[:10])
    # the light really differs below a flagged cell: DMS's photolysis term with BGC's light would be wrong there
    lit = [c for c in range(3, nc, 10) if syn["sw"][c] > 1.0 and syn["kmax"][c] > 3 and t["spChl"][c, 1] > 0.05]
    assert len(lit) > 50
    col = lit[0]
    par_bgc = f["bgc"]["kc"]["PAR_avg"][col]
    par_dms = f["dms"]["kc"]["DMS_R_PHOT"][col] / dtr[0][col] / E.eng.dms.j_dms_perI     # DMS's own PAR_avg (DMS_mod.F90:562, 702)
    assert abs(par_dms[0] / par_bgc[0] - 1.0) < 1e-12            # above the masked cell the two light passes agree ...
    assert abs(par_dms[2] / par_bgc[2] - 1.0) > 1e-6             # ... below it DMS has lost more light than BGC


def test_a_dms_par_fraction_of_its_own(pkg, oracle, synth):
    """f_qsw_par_DMS (DMS_parms.F90:191-192) is a host-settable module variable; BGC's 0.45 is a constant: when they differ
    every column takes the DMS light pass of its own."""
    from ocean_bgc_amd.modules import ColumnEngine
    from oracle import orabind
    E2 = ColumnEngine(0, path="device_colfast")
    O = orabind.Oracle()
    dp = O.dms_get_parms()
    dp[F.DMS_PARMS.index("f_qsw_par_DMS")] = 0.40
    O.dms_set_parms(dp)
    E2.dms_set_parms(dp)
    nc, nz = 2_000, 80
    syn = synth.generate(nc, nz, col0=4_000)
    tr, dtr, mtr = synth.bgc_tracer_array(syn), synth.dms_tracer_array(syn), synth.macros_tracer_array(syn)
    z = np.zeros((nc, nz))
    sst = syn["forcing"]["SST"]
    f = E2.Fused_SourceSink(*_args(syn, tr), z, z, sst, dtr, mtr, diag_init=-3.0)
    d = E2.DMS_SourceSink(dtr, syn["dz"], syn["kmax"], sst, syn["sw"], diag_init=-3.0)
    _same(d, f["dms"], "dms")
    rep = P.compare_simple(f["dms"], O.dms_source_sink(dtr, syn["dz"], syn["kmax"], sst, syn["sw"], diag_init=-3.0), "dms")
    assert rep.ok(), "\n".join(rep.fail[:10])


def test_fused_step_refuses_what_it_does_not_serve(pkg):
    import torch
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine, ObgcError
    eng = Engine(0)
    nz, nc = 60, 64
    f64 = dict(dtype=torch.float64, device="cuda")
    z = lambda *s: torch.zeros(s, **f64)
    km = torch.full((nc,), nz, dtype=torch.int32, device="cuda")
    with pytest.raises(ObgcError, match="native column-fastest or column-block layout only"):
        eng.fused_step(nz, nc, L.OBGC_LAYOUT_KFAST, z(30, nc, nz), z(nc, nz), z(nc, nz), z(nc, nz), z(nc, nz), z(nc, nz), z(nc), km,
                       z(nc, nz), z(nc), z(nc), z(30, nc, nz), z(nc, nz), z(nc, nz), None, True, dms=z(nc, nz), dmsp=z(nc, nz), SST=z(nc),
                       dms_tend=z(nc, nz), dmsp_tend=z(nc, nz))
    with pytest.raises(ObgcError, match="null DMS array"):
        eng.fused_step(nz, nc, L.OBGC_LAYOUT_COLFAST, z(30, nz, nc), z(nz, nc), z(nz, nc), z(nz, nc), z(nz, nc), z(nz, nc), z(nc), km,
                       z(nz, nc), z(nc), z(nc), z(30, nz, nc), z(nz, nc), z(nz, nc), None, True, dms=z(nz, nc), dmsp=z(nz, nc))
