"""The engine driven from the library's own files, on the GPU:

  * a parameter file written by obgc_params_write_text, read back by obgc_params_read_text, pushed with
    obgc_set_*_params, run by the kernels, compared with the oracle given the same table;
  * a restart file of the only persistent state (PH_PREV_3D, PH_PREV_ALT_CO2_3D, surface_pH, surface_pH_alt_co2;
    BGC_mod.F90:943-949, 956, 978, 2872-2888): cold step -> file -> fresh handle -> warm step must equal the
    uninterrupted cold step -> warm step bit for bit.
"""
import ctypes as C

import numpy as np
import pytest

import parity as P
from conftest import has_gpu
from oracle import fields as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(pkg):
    if not has_gpu():
        pytest.fail("needs a GPU")
    from ocean_bgc_amd import _lib as L
    return L


def test_engine_runs_from_a_parameter_file(gpu, synth, tmp_path):
    L = gpu
    from ocean_bgc_amd.modules import ColumnEngine
    from oracle import orabind
    rng = np.random.RandomState(41)
    O = orabind.Oracle()
    p = O.bgc_get_params()
    p["parms"] = p["parms"] * (1.0 + 0.1 * rng.rand(p["parms"].size))
    p["areal"] = p["areal"] * (1.0 + 0.05 * rng.rand(*p["areal"].shape))
    p["scalelen_vals"] = p["scalelen_vals"] * 1.1
    dp = O.dms_get_parms() * (1.0 + 0.1 * rng.rand(len(F.DMS_PARMS)))
    mp = O.macros_get_parms() * (1.0 + 0.1 * rng.rand(len(F.MACROS_PARMS)))
    O.bgc_set_params(p); O.dms_set_parms(dp); O.macros_set_parms(mp)
    # the same table goes into a file through a first engine's structs ...
    W = ColumnEngine(0, path="device_colfast")
    W.bgc_set_params(p); W.dms_set_parms(dp); W.macros_set_parms(mp)
    path = str(tmp_path / "tuned.params").encode()
    lib = L.load()
    assert lib.obgc_params_write_text(path, C.byref(W.eng.bgc), C.byref(W.eng.dms), C.byref(W.eng.macros)) == 0
    # ... and a second engine, holding the defaults, is configured from that file only
    E = ColumnEngine(0, path="device_colfast")
    assert E.bgc_get_params()["parms"][0] != p["parms"][0]
    err = C.create_string_buffer(512)
    assert lib.obgc_params_read_text(path, C.byref(E.eng.bgc), C.byref(E.eng.dms), C.byref(E.eng.macros), err, 512) == 0, err.value
    E.eng.push_params()
    assert bytes(E.eng.bgc) == bytes(W.eng.bgc) and bytes(E.eng.dms) == bytes(W.eng.dms) and bytes(E.eng.macros) == bytes(W.eng.macros)
    nc, nz = 1536, 80
    syn = synth.generate(nc, nz, col0=2718)
    tr = synth.bgc_tracer_array(syn)
    args = (tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"],
            syn["dust"], syn["sw"])
    z = np.zeros((nc, nz))
    rep = P.compare_bgc(E.BGC_SourceSink(*args, z, z), O.bgc_source_sink(*args, z, z))
    f = syn["forcing"]
    dtr, mtr = synth.dms_tracer_array(syn), synth.macros_tracer_array(syn)
    P.compare_simple(E.DMS_SourceSink(dtr, syn["dz"], syn["kmax"], f["SST"], syn["sw"]),
                     O.dms_source_sink(dtr, syn["dz"], syn["kmax"], f["SST"], syn["sw"]), "dms", rep)
    P.compare_simple(E.MACROS_SourceSink(mtr, syn["kmax"]), O.macros_source_sink(mtr, syn["kmax"]), "macros", rep)
    flux = [syn[nm] for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")]
    P.compare_surface(E.BGC_SurfaceFluxes(tr, f, *flux), O.bgc_surface_fluxes(tr, f, *flux), rep)
    assert rep.ok(), "\n".join(rep.fail[:10])
    # the file made a difference: the default table gives other tendencies
    D = ColumnEngine(0, path="device_colfast")
    tuned = E.BGC_SourceSink(*args, z, z, diagnostics=False)["tend"]
    assert not np.array_equal(D.BGC_SourceSink(*args, z, z, diagnostics=False)["tend"], tuned)
    # ... and a handle that has already run follows a change of its table (what it derives from the table is redone)
    D.bgc_set_params(p)
    assert np.array_equal(D.BGC_SourceSink(*args, z, z, diagnostics=False)["tend"], tuned)


def test_warm_step_resumed_from_a_restart_file_is_bit_identical(gpu, synth, tmp_path):
    L = gpu
    import torch
    from ocean_bgc_amd.engine import Engine
    nz, nc, col0 = 60, 4096, 31
    CF = L.OBGC_LAYOUT_COLFAST
    f64 = dict(dtype=torch.float64, device="cuda")

    def fresh():
        eng = Engine(0)
        b = dict(eng=eng, tr=torch.empty((30, nz, nc), **f64), kmax=torch.empty((nc,), dtype=torch.int32, device="cuda"),
                 kc={nm: torch.empty((nz, nc), **f64) for nm in ("temp", "salt", "zmid", "dz", "zbot", "fesedflux")},
                 cc={nm: torch.empty((nc,), **f64) for nm in ("lat", "dust_flux_in", "sw_flux")})
        eng.synth_fill(nz, nc, col0, tracers=b["tr"], kmax=b["kmax"], **b["kc"], **b["cc"])
        b["forcing"] = {nm: torch.empty((nc,), **f64) for nm in ("surfacePressure", "iceFraction", "windSpeedSquared10m", "atmCO2",
                                                                 "atmCO2_ALT_CO2", "surfaceDepth", "SST", "SSS", "surface_pH",
                                                                 "surface_pH_alt_co2")}
        b["flux"] = {nm: torch.empty((30, nc), **f64) for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")}
        eng.synth_fill_extra(nz, nc, col0, **b["forcing"], **b["flux"])
        b["ph"] = torch.zeros((2, nz, nc), **f64)
        torch.cuda.synchronize()
        return b

    def step(b, scale):
        """BGC_SurfaceFluxes then BGC_SourceSink on tracers scaled by `scale` (the model moved between the steps)."""
        eng = b["eng"]
        tr = b["tr"] * scale
        tend = torch.full((30, nz, nc), -7.0, **f64)
        net = torch.full((30, nc), -7.0, **f64)
        flux = [b["flux"][nm].clone() for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")]
        torch.cuda.synchronize()
        eng.bgc_surface_fluxes(nz, nc, CF, tr, b["forcing"], *flux, net)
        eng.bgc_source_sink(nz, nc, CF, tr, b["kc"]["temp"], b["kc"]["salt"], b["kc"]["zmid"], b["kc"]["dz"], b["kc"]["zbot"],
                            b["cc"]["lat"], b["kmax"], b["kc"]["fesedflux"], b["cc"]["dust_flux_in"], b["cc"]["sw_flux"], tend,
                            b["ph"][0], b["ph"][1], None, True)
        eng.sync()
        return tend, net

    # uninterrupted: cold step, then a warm step on moved tracers
    a = fresh()
    step(a, 1.0)
    state = [a["ph"][0].cpu().numpy(), a["ph"][1].cpu().numpy(), a["forcing"]["surface_pH"].cpu().numpy(),
             a["forcing"]["surface_pH_alt_co2"].cpu().numpy()]
    assert (state[0] != 0).any() and (state[2] != 0).any()
    # the state goes to disk in the engine's native (column-fastest) order through the grid variant
    g = L.Grid()
    g.nlev, g.ncol, g.layout, g.memspace = nz, nc, CF, L.OBGC_MEM_HOST
    path = str(tmp_path / "state.rst").encode()
    lib = L.load()
    ptr = lambda x: x.ctypes.data_as(C.c_void_p)
    assert lib.obgc_restart_write_grid(path, C.byref(g), *[ptr(x) for x in state]) == 0
    tend_a, net_a = step(a, 1.001)
    # resumed: a new handle, new arrays, the pH state from the file
    b = fresh()
    fl, fc = C.c_int(), C.c_int()
    assert lib.obgc_restart_read_header(path, C.byref(fl), C.byref(fc)) == 0 and (fl.value, fc.value) == (nz, nc)
    back = [np.zeros((nz, nc)), np.zeros((nz, nc)), np.zeros(nc), np.zeros(nc)]
    assert lib.obgc_restart_read_grid(path, C.byref(g), *[ptr(x) for x in back]) == 0
    b["ph"][0].copy_(torch.from_numpy(back[0])); b["ph"][1].copy_(torch.from_numpy(back[1]))
    b["forcing"]["surface_pH"].copy_(torch.from_numpy(back[2])); b["forcing"]["surface_pH_alt_co2"].copy_(torch.from_numpy(back[3]))
    # iceFraction was clamped in place by the first step of the uninterrupted run (BGC_mod.F90:2837-2838): idempotent
    tend_b, net_b = step(b, 1.001)
    assert torch.equal(tend_b, tend_a) and torch.equal(net_b, net_a)
    assert torch.equal(b["ph"], a["ph"])
    assert torch.equal(b["forcing"]["surface_pH"], a["forcing"]["surface_pH"])
    # and the warm start mattered: a cold second step leaves a different pH state (the solver stops at |dx| < 1e-10)
    c = fresh()
    step(c, 1.001)
    assert not torch.equal(c["ph"], a["ph"])
