"""GPU parity tests (run with `-m gpu` on a real MI355X).  Every call goes through the
C ABI of libobgc.so; the checker is the golden output of the unmodified reference
(tests/golden) or the C oracle on the same seeded inputs.  Tolerances: tests/parity.py.
"""
import os

import numpy as np
import pytest

import parity as P
from conftest import has_gpu
from oracle import fields as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PATHS = ("device_colfast", "device_kfast", "host")


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


@pytest.fixture(scope="module")
def engines(pkg):
    if not has_gpu():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False (no CPU fallback exists)")
    from ocean_bgc_amd.modules import ColumnEngine
    return {p: ColumnEngine(0, path=p) for p in PATHS}


def _assert(rep, what):
    assert rep.ok(), what + ":\n  " + "\n  ".join(rep.fail[:12])


def bgc_common(g):
    return (g["in.temp"], g["in.salt"], g["in.zmid"], g["in.dz"], g["in.zbot"], g["in.lat"], g["in.kmax"],
            g["in.fesed"], g["in.dust"], g["in.sw"])


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("fixture", ["config1_1x60", "branch_cols_60", "branch_cols_80", "branch_cols_100"])
def test_bgc_source_sink_vs_reference_golden(engines, path, fixture):
    E = engines[path]
    g = load(fixture)
    nc, nz = g["in.temp"].shape
    z = np.zeros((nc, nz))
    a = E.BGC_SourceSink(g["in.bgc_tracers"], *bgc_common(g), z, z, alt_co2_use_eco=True, diag_init=-7.0)
    _assert(P.compare_bgc(a, P.golden_bgc(g, "bgc_cold")), "%s cold %s" % (fixture, path))
    for nm in F.BGC_DIAG_KC_UNTOUCHED:                      # never assigned by the reference
        assert (a["kc"][nm] == -7.0).all(), nm
    # cells below the sea floor: tendencies and diagnostics are zero, PH_PREV untouched (quirk 11)
    km = g["in.kmax"]
    below = np.arange(nz)[None, :] >= km[:, None]
    assert (a["tend"][:, below] == 0).all()
    assert (a["ph_prev"][below] == 0).all()
    # warm call: feed the REFERENCE's previous pH so both sides start from the same bracket
    b = E.BGC_SourceSink(g["in.bgc_tracers_warm"], *bgc_common(g), g["bgc_cold.ph_prev"],
                         g["bgc_cold.ph_prev_alt"] + 0.01 * (g["bgc_cold.ph_prev_alt"] != 0), alt_co2_use_eco=False,
                         diag_init=-7.0)
    _assert(P.compare_bgc(b, P.golden_bgc(g, "bgc_warm")), "%s warm %s" % (fixture, path))
    assert (b["tend"][7] == 0.0).all()                      # alt_co2_use_eco = .false.


@pytest.mark.parametrize("path", PATHS)
def test_bgc_surface_fluxes_vs_reference_golden(engines, path):
    E = engines[path]
    g = load("branch_cols_60")
    f = {nm: g["in.forcing." + nm] for nm in F.BGC_SURF_FORCING}
    flux = [g["in." + nm] for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")]
    s1 = E.BGC_SurfaceFluxes(g["in.bgc_tracers"], f, *flux, diag_init=-7.0)

    def gold(pre):
        return dict(forcing={nm: g["%s.forcing.%s" % (pre, nm)] for nm in F.BGC_SURF_FORCING},
                    depositionFlux=g[pre + ".depositionFlux"], riverFlux=g[pre + ".riverFlux"], gasFlux=g[pre + ".gasFlux"],
                    seaIceFlux=g[pre + ".seaIceFlux"], netFlux=g[pre + ".netFlux"],
                    fdiag={nm: g["%s.fdiag.%s" % (pre, nm)] for nm in F.BGC_FLUX_DIAG})

    _assert(P.compare_surface(s1, gold("surf_cold")), "surface cold " + path)
    w = gold("surf_cold")
    s2 = E.BGC_SurfaceFluxes(g["in.bgc_tracers_warm"], w["forcing"], w["depositionFlux"], w["riverFlux"], w["gasFlux"],
                             w["seaIceFlux"], diag_init=-7.0)
    _assert(P.compare_surface(s2, gold("surf_warm")), "surface warm " + path)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("fixture", ["branch_cols_60", "branch_cols_80", "branch_cols_100"])
def test_dms_macros_vs_reference_golden(engines, path, fixture):
    E = engines[path]
    g = load(fixture)
    f = {nm: g["in.forcing." + nm] for nm in F.BGC_SURF_FORCING}
    d = E.DMS_SourceSink(g["in.dms_tracers"], g["in.dz"], g["in.kmax"], f["SST"], g["in.sw"], diag_init=-7.0)
    want = dict(tend=g["dms.tend"], kc={nm: g["dms.kc." + nm] for nm in F.DMS_DIAG_KC})
    _assert(P.compare_simple(d, want, "dms"), "dms " + path)
    ds = E.DMS_SurfaceFluxes(g["in.dms_tracers"], f, diag_init=-7.0)
    rep = P.Report()
    rep.note("diag", "dms.netFlux", P.e_norm(ds["netFlux"][:2], g["dms_surf.netFlux"][:2]), P.TOL_DIAG)
    assert (ds["netFlux"][2:] == -7.0).all()                # only the dms / dmsp rows are written
    for nm in F.DMS_FLUX_DIAG:
        rep.note("diag", "dms.fdiag." + nm, P.e_norm(ds["fdiag"][nm], g["dms_surf.fdiag." + nm]), P.TOL_DIAG)
    _assert(rep, "dms surface " + path)
    m = E.MACROS_SourceSink(g["in.macros_tracers"], g["in.kmax"], diag_init=-7.0)
    want = dict(tend=g["macros.tend"], kc={nm: g["macros.kc." + nm] for nm in F.MACROS_DIAG_KC})
    _assert(P.compare_simple(m, want, "macros"), "macros " + path)


def test_batch_vs_oracle_all_modules(engines, oracle, synth):
    """20k seeded columns through the native path against the C oracle; permutation of tracer slots."""
    E = engines["device_colfast"]
    nc, nz = 20000, 60
    syn = synth.generate(nc, nz, col0=31337)
    tr = synth.bgc_tracer_array(syn)
    args = (tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"],
            syn["dust"], syn["sw"])
    z = np.zeros((nc, nz))
    want = oracle.bgc_source_sink(*args, z, z)
    got = E.BGC_SourceSink(*args, z, z)
    rep = P.compare_bgc(got, want)
    _assert(rep, "20k batch")
    print("parity summary", rep.summary())
    # tendencies-only mode must give the same tendencies and pH state bit for bit
    lean = E.BGC_SourceSink(*args, z, z, diagnostics=False)
    assert np.array_equal(lean["tend"], got["tend"]) and np.array_equal(lean["ph_prev"], got["ph_prev"])
    f = syn["forcing"]
    dtr, mtr = synth.dms_tracer_array(syn), synth.macros_tracer_array(syn)
    _assert(P.compare_simple(E.DMS_SourceSink(dtr, syn["dz"], syn["kmax"], f["SST"], syn["sw"]),
                             oracle.dms_source_sink(dtr, syn["dz"], syn["kmax"], f["SST"], syn["sw"]), "dms"), "dms 20k")
    _assert(P.compare_simple(E.MACROS_SourceSink(mtr, syn["kmax"]), oracle.macros_source_sink(mtr, syn["kmax"]), "macros"),
            "macros 20k")
    flux = [syn[nm] for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")]
    _assert(P.compare_surface(E.BGC_SurfaceFluxes(tr, f, *flux), oracle.bgc_surface_fluxes(tr, f, *flux)), "surface 20k")


def test_permuted_slots_and_parameters(pkg, synth):
    from ocean_bgc_amd.modules import ColumnEngine
    from oracle import orabind
    rng = np.random.RandomState(11)
    perm = rng.permutation(30) + 1
    E = ColumnEngine(0, bgc_indices=perm, T0_Kelvin=273.16, path="device_colfast")
    O = orabind.Oracle(bgc_indices=perm, T0_Kelvin=273.16)
    p = O.bgc_get_params()
    assert np.array_equal(p["aint"], E.bgc_get_params()["aint"])
    p["parms"] = p["parms"] * (1.0 + 0.1 * rng.rand(p["parms"].size))
    p["areal"] = p["areal"] * (1.0 + 0.05 * rng.rand(*p["areal"].shape))
    O.bgc_set_params(p)
    E.bgc_set_params(p)
    nc, nz = 2000, 60
    syn = synth.generate(nc, nz, col0=999)
    indices = {nm: int(perm[i]) for i, nm in enumerate(F.BGC_TRACERS)}
    tr = synth.bgc_tracer_array(syn, indices)
    args = (tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"],
            syn["dust"], syn["sw"])
    z = np.zeros((nc, nz))
    _assert(P.compare_bgc(E.BGC_SourceSink(*args, z, z), O.bgc_source_sink(*args, z, z)), "permuted")


@pytest.mark.parametrize("variant", ["temperature functions swapped", "one grazee class", "all grazee classes distinct",
                                     "two N fixers, no calcifier"])
def test_autotroph_flag_variants(pkg, synth, variant):
    """The integer side of the autotroph table (tests/test_oracle_vs_reference.py pins the oracle on the same variants):
    nothing in the kernels may assume which group fixes N2, calcifies, shares a grazee class or follows which
    temperature function."""
    from ocean_bgc_amd.modules import ColumnEngine
    from oracle import orabind
    from test_oracle_vs_reference import FLAG_VARIANTS, with_flags
    E = ColumnEngine(0, path="device_colfast")
    O = orabind.Oracle()
    p = with_flags(O.bgc_get_params(), FLAG_VARIANTS[variant])
    O.bgc_set_params(p)
    E.bgc_set_params(p)
    nc, nz = 1500, 60
    syn = synth.generate(nc, nz, col0=4242)
    args = (synth.bgc_tracer_array(syn), syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"],
            syn["fesed"], syn["dust"], syn["sw"])
    z = np.zeros((nc, nz))
    _assert(P.compare_bgc(E.BGC_SourceSink(*args, z, z), O.bgc_source_sink(*args, z, z)), variant)


@pytest.mark.parametrize("path", PATHS)
def test_dms_macros_permuted_slots_and_parameters(pkg, synth, path):
    """DMS / MACROS with their tracer slots shuffled and every parameter off its default (the oracle is pinned on the
    same case in tests/test_oracle_vs_reference.py)."""
    from ocean_bgc_amd.modules import ColumnEngine
    from oracle import orabind
    from test_oracle_vs_reference import dms_macros_case
    rng = np.random.RandomState(23)
    dperm, mperm, syn, dtr, mtr = dms_macros_case(synth, rng)
    E = ColumnEngine(0, dms_indices=dperm, macros_indices=mperm, path=path)
    O = orabind.Oracle(dms_indices=dperm, macros_indices=mperm)
    dp = O.dms_get_parms() * (1.0 + 0.1 * rng.rand(len(F.DMS_PARMS)))
    mp = O.macros_get_parms() * (1.0 + 0.1 * rng.rand(len(F.MACROS_PARMS)))
    O.dms_set_parms(dp); E.dms_set_parms(dp)
    O.macros_set_parms(mp); E.macros_set_parms(mp)
    f = syn["forcing"]
    rep = P.Report()
    P.compare_simple(E.DMS_SourceSink(dtr, syn["dz"], syn["kmax"], f["SST"], syn["sw"], diag_init=-2.0),
                     O.dms_source_sink(dtr, syn["dz"], syn["kmax"], f["SST"], syn["sw"], diag_init=-2.0), "dms", rep)
    P.compare_simple(E.MACROS_SourceSink(mtr, syn["kmax"], diag_init=-2.0), O.macros_source_sink(mtr, syn["kmax"], diag_init=-2.0),
                     "macros", rep)
    a, b = E.DMS_SurfaceFluxes(dtr, f, diag_init=-2.0), O.dms_surface_fluxes(dtr, f, diag_init=-2.0)
    written = [int(dperm[0]) - 1, int(dperm[1]) - 1]                      # the dms / dmsp rows, wherever the host put them
    rep.note("diag", "dms.netFlux", P.e_norm(a["netFlux"][written], b["netFlux"][written]), P.TOL_DIAG)
    assert np.array_equal(a["netFlux"], b["netFlux"]) or (np.delete(a["netFlux"], written, axis=0) == -2.0).all()
    _assert(rep, "dms/macros permuted " + path)


@pytest.mark.parametrize("path", PATHS)
def test_gas_flux_switches(engines, oracle, synth, path):
    """lcalc_O2_gas_flux / lcalc_CO2_gas_flux / lcalc_DMS_gas_flux = .false. (BGC_mod.F90:2848, 2884; DMS_mod.F90:846):
    the fluxes and their diagnostics that are then skipped keep what the reference leaves in them."""
    E = engines[path]
    nc, nz = 3000, 60
    syn = synth.generate(nc, nz, col0=777)
    tr = synth.bgc_tracer_array(syn)
    f = syn["forcing"]
    flux = [syn[nm] for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")]
    for kw in (dict(lcalc_o2=False), dict(lcalc_co2=False), dict(lcalc_o2=False, lcalc_co2=False)):
        _assert(P.compare_surface(E.BGC_SurfaceFluxes(tr, f, *flux, diag_init=-2.0, **kw),
                                  oracle.bgc_surface_fluxes(tr, f, *flux, diag_init=-2.0, **kw)), "%s %s" % (path, kw))
    dtr = synth.dms_tracer_array(syn)
    a, b = E.DMS_SurfaceFluxes(dtr, f, lcalc=False, diag_init=-2.0), oracle.dms_surface_fluxes(dtr, f, lcalc=False, diag_init=-2.0)
    rep = P.Report()
    rep.note("diag", "dms.netFlux", P.e_norm(a["netFlux"], b["netFlux"]), P.TOL_DIAG)
    for nm in b["fdiag"]:
        rep.note("diag", "dms.fdiag." + nm, P.e_norm(a["fdiag"][nm], b["fdiag"][nm]), P.TOL_DIAG)
    _assert(rep, "dms lcalc=False " + path)


def test_device_generator_is_bit_identical_to_numpy(pkg, synth):
    import torch
    from ocean_bgc_amd.engine import Engine
    eng = Engine(0)
    nc, nz, col0 = 5000, 60, 123456
    dev = "cuda:0"
    t = {nm: torch.empty((nz, nc), dtype=torch.float64, device=dev) for nm in ("temp", "salt", "zmid", "dz", "zbot", "fesedflux")}
    tr = torch.empty((30, nz, nc), dtype=torch.float64, device=dev)
    c = {nm: torch.empty((nc,), dtype=torch.float64, device=dev) for nm in ("lat", "dust_flux_in", "sw_flux")}
    km = torch.empty((nc,), dtype=torch.int32, device=dev)
    eng.synth_fill(nz, nc, col0, tracers=tr, kmax=km, **t, **c)
    syn = synth.generate(nc, nz, col0=col0)
    assert np.array_equal(km.cpu().numpy(), syn["kmax"])
    for nm, key in (("temp", "temp"), ("salt", "salt"), ("zmid", "zmid"), ("dz", "dz"), ("zbot", "zbot"), ("fesedflux", "fesed")):
        assert np.array_equal(t[nm].cpu().numpy().T, syn[key]), nm
    for nm, key in (("lat", "lat"), ("dust_flux_in", "dust"), ("sw_flux", "sw")):
        assert np.array_equal(c[nm].cpu().numpy(), syn[key]), nm
    ref = synth.bgc_tracer_array(syn)
    got = tr.cpu().numpy().transpose(0, 2, 1)
    for i, nm in enumerate(synth.BGC_TRACERS):
        assert np.array_equal(got[i], ref[i]), nm


def test_reference_layout_tiled_transposes(pkg, oracle, synth, monkeypatch):
    """Reference-layout callers go through LDS-tiled transposes in column tiles: force several tiles
    (incl. a short last one) and require the same bits as the native-layout path."""
    from ocean_bgc_amd.modules import ColumnEngine
    monkeypatch.setenv("OBGC_KFAST_TILE", "704")
    Ek = ColumnEngine(0, path="device_kfast")
    Eh = ColumnEngine(0, path="host")
    monkeypatch.delenv("OBGC_KFAST_TILE")
    Ec = ColumnEngine(0, path="device_colfast")
    nc, nz = 2000, 60
    syn = synth.generate(nc, nz, col0=5150)
    tr = synth.bgc_tracer_array(syn)
    args = (tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"],
            syn["dust"], syn["sw"])
    z = np.zeros((nc, nz))
    a, b, c = Ek.BGC_SourceSink(*args, z, z, diag_init=-7.0), Ec.BGC_SourceSink(*args, z, z, diag_init=-7.0), \
        Eh.BGC_SourceSink(*args, z, z, diag_init=-7.0)
    for other in (a, c):
        assert np.array_equal(other["tend"], b["tend"]) and np.array_equal(other["ph_prev"], b["ph_prev"])
        for grp in ("kc", "kca", "ca", "c"):
            for nm in b[grp]:
                assert np.array_equal(other[grp][nm], b[grp][nm]), (grp, nm)
    _assert(P.compare_bgc(a, oracle.bgc_source_sink(*args, z, z, diag_init=-7.0)), "tiled reference layout")


@pytest.mark.parametrize("threads", ["1", "0"])
def test_host_path_pipelined_in_column_chunks(pkg, synth, monkeypatch, threads):
    """Host-memory callers in the reference layout are served chunk by chunk (upload, kernels and download of
    different column chunks overlap).  Force five chunks with a short last one, plain and with padded leading
    dimensions, and require the bits of the one-piece path; the padding and the never-written members stay as they were."""
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine
    nc, nz = 1500, 23
    syn = synth.generate(nc, nz, col0=31)
    tr = synth.bgc_tracer_array(syn)
    kc_names = ("temp", "salt", "zmid", "dz", "zbot", "fesed")

    def run(eng, ld, extra):
        plane = ld * (nc + extra)

        def pad(x, npl):
            buf = np.full((npl * plane,), np.nan)
            view = np.lib.stride_tricks.as_strided(buf, (npl, nc, nz), (plane * 8, ld * 8, 8))
            view[...] = np.asarray(x).reshape(npl, nc, nz)
            return buf, view
        t = {"tr": pad(tr, 30)}
        for nm in kc_names:
            t[nm] = pad(syn[nm], 1)
        t["tend"] = pad(np.zeros((30, nc, nz)), 30)
        t["php"], t["pha"] = pad(np.zeros((nc, nz)), 1), pad(np.zeros((nc, nz)), 1)
        diag = {nm: pad(np.full((nc, nz), -7.0), 1) for nm in L.BGC_DIAG_KC}
        diag.update({nm: pad(np.full((4, nc, nz), -7.0), 4) for nm in L.BGC_DIAG_KCA})
        dvec = {nm: np.full((4, nc), -7.0) for nm in L.BGC_DIAG_CA}
        dvec.update({nm: np.full((nc,), -7.0) for nm in L.BGC_DIAG_C})
        d = {nm: b for nm, (b, _) in diag.items()}
        d.update(dvec)
        eng.bgc_source_sink(nz, nc, L.OBGC_LAYOUT_KFAST, t["tr"][0], *(t[nm][0] for nm in kc_names[:5]), syn["lat"],
                            syn["kmax"], t["fesed"][0], syn["dust"], syn["sw"], t["tend"][0], t["php"][0], t["pha"][0], d, True,
                            ld=ld, plane=plane)
        out = {"tend": t["tend"], "php": t["php"], "pha": t["pha"]}
        out.update(diag)
        return out, dvec

    monkeypatch.setenv("OBGC_HOST_CHUNKS", "0")
    one = Engine(0)
    monkeypatch.setenv("OBGC_HOST_CHUNKS", "5")
    monkeypatch.setenv("OBGC_HOST_PIPELINE_MIN_MB", "0")
    monkeypatch.setenv("OBGC_HOST_THREADS", threads)          # "0": the fallback without a helper thread
    piped = Engine(0)
    for ld, extra in ((nz, 0), (nz + 3, 2)):
        a, av = run(one, ld, extra)
        b, bv = run(piped, ld, extra)
        for nm in a:
            assert np.array_equal(a[nm][0], b[nm][0], equal_nan=True), (ld, nm)          # whole buffers, padding included
            inside = np.zeros(a[nm][0].shape, dtype=bool)
            np.lib.stride_tricks.as_strided(inside, a[nm][1].shape, tuple(s // 8 for s in a[nm][1].strides))[...] = True
            assert np.isnan(b[nm][0][~inside]).all(), (ld, nm)
        for nm in av:
            assert np.array_equal(av[nm], bv[nm]), nm
        for nm in L.BGC_DIAG_KC_UNTOUCHED:
            assert (b[nm][1] == -7.0).all(), nm
        assert not (b["tend"][1] == 0).all()


@pytest.mark.parametrize("layout", ["colfast", "kfast"])
def test_padded_strides_leave_the_padding_alone(pkg, synth, layout):
    """Callers may pass arrays larger than the batch (ld > extent, plane > ld*extent; the reference's
    numLevelsMax / numColumnsMax).  Results must equal the dense call bit for bit and the padding must
    stay untouched."""
    import torch
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine
    eng = Engine(0)
    nc, nz = 1000, 37
    syn = synth.generate(nc, nz, col0=77)
    tr = synth.bgc_tracer_array(syn)
    kc_names = ("temp", "salt", "zmid", "dz", "zbot", "fesed")
    f64 = dict(dtype=torch.float64, device="cuda")
    if layout == "colfast":
        ld, rows = nc + 13, nz
        lay = L.OBGC_LAYOUT_COLFAST
        def dense(x): return np.ascontiguousarray(np.swapaxes(x, -1, -2))          # [.., nz, nc]
    else:
        ld, rows = nz + 3, nc
        lay = L.OBGC_LAYOUT_KFAST
        def dense(x): return np.ascontiguousarray(x)                                # [.., nc, nz]
    plane = ld * (rows + 5) + 7
    ldc = nc + 9

    def padded(x, nplanes):
        """-> (flat tensor with NaN padding, dense view into it)"""
        d = torch.from_numpy(dense(x)).cuda()
        flat = torch.full((nplanes * plane,), float("nan"), **f64)
        view = torch.as_strided(flat, (nplanes, d.shape[-2], d.shape[-1]), (plane, ld, 1))
        view.copy_(d.reshape(nplanes, d.shape[-2], d.shape[-1]))
        return flat, view

    def run(pad):
        t = {}
        if pad:
            t["tr"], _ = padded(tr, 30)
            for nm in kc_names:
                t[nm], _ = padded(syn[nm][None], 1)
            t["tend"], tv = padded(np.zeros_like(tr), 30)
            t["php"], pv = padded(np.zeros((1, nc, nz)), 1)
            t["pha"], av = padded(np.zeros((1, nc, nz)), 1)
            t["lim"], lv = padded(np.zeros((4, nc, nz)), 4)
            zint_flat = torch.full((4 * ldc,), float("nan"), **f64)
            zint = torch.as_strided(zint_flat, (4, nc), (ldc, 1))
            kw = dict(ld=ld, plane=plane, ldc=ldc)
        else:
            t["tr"] = torch.from_numpy(dense(tr)).cuda()
            for nm in kc_names:
                t[nm] = torch.from_numpy(dense(syn[nm])).cuda()
            tv = t["tend"] = torch.zeros_like(t["tr"])
            pv = t["php"] = torch.zeros_like(t["temp"]); av = t["pha"] = torch.zeros_like(t["temp"])
            lv = t["lim"] = torch.zeros((4,) + tuple(t["temp"].shape), **f64)
            zint_flat = zint = torch.zeros((4, nc), **f64)
            kw = {}
        c = {nm: torch.from_numpy(np.ascontiguousarray(syn[k])).cuda() for nm, k in (("lat", "lat"), ("dust", "dust"), ("sw", "sw"))}
        km = torch.from_numpy(syn["kmax"]).cuda()
        diag = {"N_lim": t["lim"], "photoC_zint": zint_flat, "Jint_Ctot": torch.zeros((nc,), **f64)}
        torch.cuda.synchronize()
        eng.bgc_source_sink(nz, nc, lay, t["tr"], t["temp"], t["salt"], t["zmid"], t["dz"], t["zbot"], c["lat"], km, t["fesed"],
                            c["dust"], c["sw"], t["tend"], t["php"], t["pha"], diag, True, **kw)
        return t, tv.clone(), pv.clone(), lv.clone(), zint.clone(), zint_flat

    tp, tend_p, ph_p, lim_p, zint_p, zflat = run(True)
    td, tend_d, ph_d, lim_d, zint_d, _ = run(False)
    assert torch.equal(tend_p.reshape(tend_d.shape), tend_d)
    assert torch.equal(ph_p.reshape(ph_d.shape), ph_d)
    assert torch.equal(lim_p.reshape(lim_d.shape), lim_d)
    assert torch.equal(zint_p, zint_d)
    # padding: everything outside the strided views is still NaN
    for key, npl in (("tend", 30), ("php", 1), ("lim", 4)):
        flat = tp[key]
        mask = torch.ones_like(flat, dtype=torch.bool)
        view_mask = torch.as_strided(mask, (npl, tend_p.shape[-2] if key != "php" else ph_p.shape[-2], ph_p.shape[-1]), (plane, ld, 1))
        view_mask.fill_(False)
        assert torch.isnan(flat[mask]).all(), key
    zmask = torch.ones_like(zflat, dtype=torch.bool)
    torch.as_strided(zmask, (4, nc), (ldc, 1)).fill_(False)
    assert torch.isnan(zflat[zmask]).all()
