"""GPU tests at BASELINE.json's full sizes (configs 2, 3 and 4) through size-independent properties:
determinism, shard invariance (bit-identical per-column results for any batch offset), the
reference's zero-fill of land / sub-floor cells, conservation of C, N, P, Si (the Jint_*
integrals vanish to round-off), warm-start consistency of the pH solve, and a strided column
sample checked against the oracle."""
import numpy as np
import pytest

import parity as P
from conftest import has_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(pkg):
    if not has_gpu():
        pytest.fail("needs a GPU")
    import torch
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine
    return torch, L, Engine(0)


def make_batch(torch, L, eng, nz, nc, col0, diag):
    f64 = dict(dtype=torch.float64, device="cuda")
    b = dict(nz=nz, nc=nc)
    b["tr"] = torch.empty((30, nz, nc), **f64)
    b["kc"] = {nm: torch.empty((nz, nc), **f64) for nm in ("temp", "salt", "zmid", "dz", "zbot", "fesedflux")}
    b["cc"] = {nm: torch.empty((nc,), **f64) for nm in ("lat", "dust_flux_in", "sw_flux")}
    b["kmax"] = torch.empty((nc,), dtype=torch.int32, device="cuda")
    eng.synth_fill(nz, nc, col0, tracers=b["tr"], kmax=b["kmax"], **b["kc"], **b["cc"])
    b["tend"] = torch.full((30, nz, nc), -7.0, **f64)
    b["ph"] = torch.zeros((2, nz, nc), **f64)
    b["diag"] = None
    if diag:
        d = {}
        for nm in L.BGC_DIAG_KC:
            d[nm] = torch.full((nz, nc), -7.0, **f64)
        for nm in L.BGC_DIAG_KCA:
            d[nm] = torch.full((4, nz, nc), -7.0, **f64)
        for nm in L.BGC_DIAG_CA:
            d[nm] = torch.full((4, nc), -7.0, **f64)
        for nm in L.BGC_DIAG_C:
            d[nm] = torch.full((nc,), -7.0, **f64)
        b["diag"] = d
    torch.cuda.synchronize()
    return b


def run(L, eng, b):
    kc, cc = b["kc"], b["cc"]
    eng.bgc_source_sink(b["nz"], b["nc"], L.OBGC_LAYOUT_COLFAST, b["tr"], kc["temp"], kc["salt"], kc["zmid"], kc["dz"],
                        kc["zbot"], cc["lat"], b["kmax"], kc["fesedflux"], cc["dust_flux_in"], cc["sw_flux"], b["tend"],
                        b["ph"][0], b["ph"][1], b["diag"], True)


def test_cfg2_100k_x_60_properties(gpu, oracle, synth):
    torch, L, eng = gpu
    nz, nc = 60, 100_000
    b = make_batch(torch, L, eng, nz, nc, 0, diag=True)
    run(L, eng, b)
    tend1, ph1 = b["tend"].clone(), b["ph"].clone()
    assert torch.isfinite(tend1).all() and torch.isfinite(ph1).all()
    # determinism: the same inputs give the same bits
    b["ph"].zero_()
    torch.cuda.synchronize()
    run(L, eng, b)
    assert torch.equal(b["tend"], tend1) and torch.equal(b["ph"], ph1)
    # zero fill below the sea floor and on land (BGC_mod.F90:570, 625-727, 802)
    k = torch.arange(nz, device="cuda")[:, None]
    below = k >= b["kmax"][None, :]
    assert (tend1[:, below] == 0).all() and (ph1[:, below] == 0).all()
    for nm in ("photoC_TOT", "CO3", "POC_REMIN"):
        assert (b["diag"][nm][below] == 0).all(), nm
    assert (b["diag"]["photoC"][:, below] == 0).all()
    for nm in L.BGC_DIAG_KC_UNTOUCHED:
        assert (b["diag"][nm] == -7.0).all()
    land = b["kmax"] == 0
    assert land.any() and (b["diag"]["O2_ZMIN"][land] == 0).all() and (b["diag"]["zsatcalc"][land] == 0).all()
    # conservation: vertical integrals of the C, N, P, Si budgets close to round-off
    scale = float(b["diag"]["photoC_TOT_zint"].abs().max())
    for nm, fac in (("Jint_Ctot", 1.0), ("Jint_Ntot", 0.137), ("Jint_Ptot", 0.00855), ("Jint_Sitot", 0.137)):
        assert float(b["diag"][nm].abs().max()) <= 1e-10 * scale * fac, nm
    # shard invariance: a 1024-column window generated and solved on its own is bit-identical
    w0, wn = 40_000, 1024
    s = make_batch(torch, L, eng, nz, wn, w0, diag=True)
    run(L, eng, s)
    assert torch.equal(s["tend"], tend1[:, :, w0:w0 + wn])
    assert torch.equal(s["ph"], ph1[:, :, w0:w0 + wn])
    for nm in ("Jint_Ctot", "zsatarag", "photoC_NO3_TOT_zint"):
        assert torch.equal(s["diag"][nm], b["diag"][nm][w0:w0 + wn]), nm
    assert torch.equal(s["diag"]["light_lim"], b["diag"]["light_lim"][:, :, w0:w0 + wn])
    # warm start from the converged state reproduces the pH to the reference solver's accuracy:
    # xacc = 1e-10 on [H+] ~ 1e-8 (co2calc.F90:47-53, "3 significant figures") -> |dpH| <~ 4e-3
    cold = ph1.clone()
    run(L, eng, b)       # b["ph"] now holds the cold solution -> narrow bracket
    live = ~below
    assert float((b["ph"][0][live] - cold[0][live]).abs().max()) < 5e-3
    # strided sample against the oracle (inputs are bit-identical to the numpy generator)
    cols = np.arange(0, nc, 397)
    tr_h = b["tr"][:, :, cols].cpu().numpy().transpose(0, 2, 1)
    kc_h = {nm: v[:, cols].cpu().numpy().T for nm, v in b["kc"].items()}
    cc_h = {nm: v[cols].cpu().numpy() for nm, v in b["cc"].items()}
    z = np.zeros((cols.size, nz))
    want = oracle.bgc_source_sink(tr_h, kc_h["temp"], kc_h["salt"], kc_h["zmid"], kc_h["dz"], kc_h["zbot"], cc_h["lat"],
                                  b["kmax"][cols].cpu().numpy(), kc_h["fesedflux"], cc_h["dust_flux_in"], cc_h["sw_flux"], z, z,
                                  diagnostics=False)
    rep = P.Report()
    P.compare_tend(rep, "tend", tend1[:, :, cols].cpu().numpy().transpose(0, 2, 1), want["tend"])
    rep.note("ph", "ph_prev", P.e_norm(ph1[0][:, cols].cpu().numpy().T, want["ph_prev"]), P.TOL_PH)
    assert rep.ok(), "\n".join(rep.fail[:8])


def make_surface(torch, L, eng, nz, nc, col0):
    f64 = dict(dtype=torch.float64, device="cuda")
    s = dict(forcing={nm: torch.empty((nc,), **f64) for nm in ("surfacePressure", "iceFraction", "windSpeedSquared10m", "atmCO2",
                                                               "atmCO2_ALT_CO2", "surfaceDepth", "SST", "SSS", "surface_pH",
                                                               "surface_pH_alt_co2")},
             flux={nm: torch.empty((30, nc), **f64) for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")},
             net=torch.full((30, nc), -7.0, **f64), diag={nm: torch.full((nc,), -7.0, **f64) for nm in L.BGC_FLUX_DIAG})
    eng.synth_fill_extra(nz, nc, col0, **s["forcing"], **s["flux"])
    torch.cuda.synchronize()
    return s


def run_surface(L, eng, b, s):
    eng.bgc_surface_fluxes(b["nz"], b["nc"], L.OBGC_LAYOUT_COLFAST, b["tr"], s["forcing"],
                           *[s["flux"][nm] for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")], s["net"], s["diag"])


def test_cfg3_1M_x_60_surface_fluxes_and_tendencies(gpu, oracle, synth):
    """BASELINE config #3 at full size: BGC_SurfaceFluxes + BGC_SourceSink over 1M columns (no interior diagnostics:
    62 GB of them would fit, but the property is size, not content)."""
    torch, L, eng = gpu
    nz, nc = 60, 1_000_000
    b = make_batch(torch, L, eng, nz, nc, 0, diag=False)
    sf = make_surface(torch, L, eng, nz, nc, 0)
    ice_in = sf["forcing"]["iceFraction"].clone()
    run_surface(L, eng, b, sf)
    run(L, eng, b)
    assert torch.isfinite(b["tend"]).all()
    k = torch.arange(nz, device="cuda")[:, None]
    below = k >= b["kmax"][None, :]
    assert (b["tend"][:, below] == 0).all()
    w0, wn = 987_000, 640
    s = make_batch(torch, L, eng, nz, wn, w0, diag=False)
    run(L, eng, s)
    assert torch.equal(s["tend"], b["tend"][:, :, w0:w0 + wn]) and torch.equal(s["ph"], b["ph"][:, :, w0:w0 + wn])
    # ---- the surface call at 1M columns: finite, in-place clamp, a window on its own, an oracle sample ----
    assert torch.isfinite(sf["net"]).all() and all(torch.isfinite(v).all() for v in sf["diag"].values())
    assert torch.equal(sf["forcing"]["iceFraction"], ice_in.clamp(0.0, 1.0)) and (ice_in < 0).any() and (ice_in > 1).any()
    ph = sf["forcing"]["surface_pH"]
    assert float(ph.min()) > 4.0 and float(ph.max()) < 10.0          # land columns included: the reference solves them too
    ss = make_surface(torch, L, eng, nz, wn, w0)
    run_surface(L, eng, s, ss)
    assert torch.equal(ss["net"], sf["net"][:, w0:w0 + wn]) and torch.equal(ss["forcing"]["surface_pH"], ph[w0:w0 + wn])
    for nm, v in ss["diag"].items():
        assert torch.equal(v, sf["diag"][nm][w0:w0 + wn]), nm
    cols = np.arange(0, nc, 1999)
    syn_cols = [synth.generate(1, nz, col0=int(c)) for c in cols[:64]]
    for j, sy in enumerate(syn_cols[:8]):                              # device inputs are the numpy generator's, bit for bit
        assert np.array_equal(b["tr"][:, :, int(cols[j])].cpu().numpy(), synth.bgc_tracer_array(sy)[:, 0, :])
    tr_h = np.concatenate([synth.bgc_tracer_array(sy) for sy in syn_cols], axis=1)
    f_h = {nm: np.concatenate([sy["forcing"][nm] for sy in syn_cols]) for nm in syn_cols[0]["forcing"]}
    flux_h = [np.concatenate([sy[nm] for sy in syn_cols], axis=1) for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")]
    want = oracle.bgc_surface_fluxes(tr_h, f_h, *flux_h)
    sel = torch.as_tensor(cols[:64], device="cuda")
    got = dict(forcing={nm: v[sel].cpu().numpy() for nm, v in sf["forcing"].items()}, netFlux=sf["net"][:, sel].cpu().numpy(),
               fdiag={nm: v[sel].cpu().numpy() for nm, v in sf["diag"].items()})
    for nm, a in (("depositionFlux", 0), ("riverFlux", 1), ("gasFlux", 2), ("seaIceFlux", 3)):
        got[nm] = sf["flux"][nm][:, sel].cpu().numpy()
    rep = P.compare_surface(got, want)
    assert rep.ok(), "\n".join(rep.fail[:8])


def test_large_batch_on_the_separate_kernels(gpu, monkeypatch):
    """The separate attenuation / scan / cell / sweep kernels (OBGC_COLUMN_KERNEL=0) above 10 M cells with diagnostics: the
    carbonate system is solved inside the cell kernel's wavefronts (schedule 2).  400 000 columns x 60 with every
    diagnostic, against a 640-column window computed by the default engine (column-block kernel, carbonate kernel beside
    it) -- every output bit-identical, conservation to round-off, zero fill below the floor."""
    torch, L, eng = gpu
    from ocean_bgc_amd.engine import Engine
    monkeypatch.setenv("OBGC_COLUMN_KERNEL", "0")
    eng_sep = Engine(0)
    monkeypatch.delenv("OBGC_COLUMN_KERNEL")
    nz, nc = 60, 400_000
    b = make_batch(torch, L, eng_sep, nz, nc, 0, diag=True)
    run(L, eng_sep, b)
    assert eng_sep.last_schedule() == 2
    k = torch.arange(nz, device="cuda")[:, None]
    below = k >= b["kmax"][None, :]
    assert torch.isfinite(b["tend"]).all() and (b["tend"][:, below] == 0).all()
    scale = float(b["diag"]["photoC_TOT_zint"].abs().max())
    for nm, fac in (("Jint_Ctot", 1.0), ("Jint_Ntot", 0.137), ("Jint_Ptot", 0.00855), ("Jint_Sitot", 0.137)):
        assert float(b["diag"][nm].abs().max()) <= 1e-10 * scale * fac, nm
    w0, wn = 333_312, 640
    s = make_batch(torch, L, eng, nz, wn, w0, diag=True)
    run(L, eng, s)
    assert eng.last_schedule() == 4
    assert torch.equal(s["tend"], b["tend"][:, :, w0:w0 + wn]) and torch.equal(s["ph"], b["ph"][:, :, w0:w0 + wn])
    for nm, v in s["diag"].items():
        assert torch.equal(v, b["diag"][nm][..., w0:w0 + wn]), nm
    del b, s
    torch.cuda.empty_cache()


def make_own(torch, L, eng, b, col0, diag):
    """DMS / MACROS side of the fused step: own tracers, surface temperature, tendencies, diagnostics (device)."""
    nz, nc = b["nz"], b["nc"]
    f64 = dict(dtype=torch.float64, device="cuda")
    o = dict(own=torch.empty((5, nz, nc), **f64), sst=torch.empty((nc,), **f64), tend=torch.full((5, nz, nc), -7.0, **f64))
    eng.synth_fill_extra(nz, nc, col0, dms=o["own"][0], dmsp=o["own"][1], prot=o["own"][2], poly=o["own"][3], lip=o["own"][4],
                         SST=o["sst"])
    o["dms_diag"] = {nm: torch.full((nz, nc), -7.0, **f64) for nm in L.DMS_DIAG_KC} if diag else None
    o["macros_diag"] = {nm: torch.full((nz, nc), -7.0, **f64) for nm in L.MACROS_DIAG_KC} if diag else None
    torch.cuda.synchronize()
    return o


def run_fused(L, eng, b, o):
    kc, cc, own, t = b["kc"], b["cc"], o["own"], o["tend"]
    eng.fused_step(b["nz"], b["nc"], L.OBGC_LAYOUT_COLFAST, b["tr"], kc["temp"], kc["salt"], kc["zmid"], kc["dz"], kc["zbot"],
                   cc["lat"], b["kmax"], kc["fesedflux"], cc["dust_flux_in"], cc["sw_flux"], b["tend"], b["ph"][0], b["ph"][1],
                   b["diag"], True, dms_diag=o["dms_diag"], macros_diag=o["macros_diag"], dms=own[0], dmsp=own[1], SST=o["sst"],
                   dms_tend=t[0], dmsp_tend=t[1], prot=own[2], poly=own[3], lip=own[4], prot_tend=t[2], poly_tend=t[3],
                   lip_tend=t[4])


def test_cfg4_1M_x_80_fused_step_properties(gpu, oracle, synth):
    """BASELINE config #4 at full size through obgc_fused_step (BGC + DMS + MACROS, every diagnostic, ~180 GB on the device):
    finite, deterministic, the reference's zero fill, conservation to round-off, a column window solved on its own
    bit-identical in EVERY output of the three modules, and a strided column sample against the oracle."""
    torch, L, eng = gpu
    torch.cuda.empty_cache()
    nz, nc = 80, 1_000_000
    b = make_batch(torch, L, eng, nz, nc, 0, diag=True)
    o = make_own(torch, L, eng, b, 0, diag=True)
    run_fused(L, eng, b, o)
    eng.sync()
    assert eng.last_schedule() in (3, 4)
    tend1, own1, ph1 = b["tend"].clone(), o["tend"].clone(), b["ph"].clone()
    assert torch.isfinite(tend1).all() and torch.isfinite(own1).all() and torch.isfinite(ph1).all()
    keep = {nm: b["diag"][nm].clone() for nm in ("Jint_Ctot", "zsatcalc", "POC_REMIN", "PAR_avg")}
    keep_d = o["dms_diag"]["DMS_R_PHOT"].clone()
    # determinism
    b["ph"].zero_()
    torch.cuda.synchronize()               # the engine runs on streams of its own
    run_fused(L, eng, b, o)
    eng.sync()
    assert torch.equal(b["tend"], tend1) and torch.equal(o["tend"], own1) and torch.equal(b["ph"], ph1)
    for nm, v in keep.items():
        assert torch.equal(b["diag"][nm], v), nm
    assert torch.equal(o["dms_diag"]["DMS_R_PHOT"], keep_d)
    del keep, keep_d
    # zero fill below the sea floor and on land, in all three modules (BGC_mod.F90:570, DMS_mod.F90:456-468, MACROS_mod.F90:290-299)
    k = torch.arange(nz, device="cuda")[:, None]
    below = k >= b["kmax"][None, :]
    assert (tend1[:, below] == 0).all() and (own1[:, below] == 0).all() and (ph1[:, below] == 0).all()
    for nm in ("photoC_TOT", "CO3", "POC_REMIN"):
        assert (b["diag"][nm][below] == 0).all(), nm
    # DMS and MACROS leave their diagnostics below the floor as they found them (DMS_mod.F90:723-761 and
    # MACROS_mod.F90:384-402 sit inside the loop over the active levels)
    assert (o["dms_diag"]["DMS_R_TOTAL"][below] == -7.0).all() and (o["macros_diag"]["PROT_S_TOTAL"][below] == -7.0).all()
    for nm in L.BGC_DIAG_KC_UNTOUCHED:
        assert (b["diag"][nm] == -7.0).all()
    # conservation
    scale = float(b["diag"]["photoC_TOT_zint"].abs().max())
    for nm, fac in (("Jint_Ctot", 1.0), ("Jint_Ntot", 0.137), ("Jint_Ptot", 0.00855), ("Jint_Sitot", 0.137)):
        assert float(b["diag"][nm].abs().max()) <= 1e-10 * scale * fac, nm
    # a window of 1 000 columns (not a multiple of the 64-column block) on its own: every output of the three modules
    w0, wn = 765_433, 1_000
    s = make_batch(torch, L, eng, nz, wn, w0, diag=True)
    so = make_own(torch, L, eng, s, w0, diag=True)
    run_fused(L, eng, s, so)
    eng.sync()
    assert torch.equal(s["tend"], tend1[:, :, w0:w0 + wn]) and torch.equal(s["ph"], ph1[:, :, w0:w0 + wn])
    assert torch.equal(so["tend"], own1[:, :, w0:w0 + wn])
    for nm, v in s["diag"].items():
        assert torch.equal(v, b["diag"][nm][..., w0:w0 + wn]), nm
    for grp in ("dms_diag", "macros_diag"):
        for nm, v in so[grp].items():
            assert torch.equal(v, o[grp][nm][:, w0:w0 + wn]), nm
    del s, so
    # strided sample against the oracle: BGC tendencies and pH, DMS and MACROS tendencies and diagnostics
    cols = np.arange(0, nc, 3989)
    sel = torch.as_tensor(cols, device="cuda")
    h = lambda x: x.index_select(-1, sel).cpu().numpy()
    tr_h = h(b["tr"]).transpose(0, 2, 1)
    kc_h = {nm: h(v).T for nm, v in b["kc"].items()}
    cc_h = {nm: h(v) for nm, v in b["cc"].items()}
    kmax_h = h(b["kmax"])
    z = np.zeros((cols.size, nz))
    want = oracle.bgc_source_sink(tr_h, kc_h["temp"], kc_h["salt"], kc_h["zmid"], kc_h["dz"], kc_h["zbot"], cc_h["lat"], kmax_h,
                                  kc_h["fesedflux"], cc_h["dust_flux_in"], cc_h["sw_flux"], z, z, diagnostics=False)
    rep = P.Report()
    P.compare_tend(rep, "tend", h(tend1).transpose(0, 2, 1), want["tend"])
    rep.note("ph", "ph_prev", P.e_norm(h(ph1[0]).T, want["ph_prev"]), P.TOL_PH)
    own_h = h(o["own"]).transpose(0, 2, 1)                       # [5, c, k]
    named = {nm: tr_h[i] for i, nm in enumerate(L.BGC_TRACERS)}
    named.update(dms=own_h[0], dmsp=own_h[1], prot=own_h[2], poly=own_h[3], lip=own_h[4])
    dtr = np.stack([named[nm] for nm in L.DMS_TRACERS])
    mtr = np.stack([named[nm] for nm in L.MACROS_TRACERS])
    want_d = oracle.dms_source_sink(dtr, kc_h["dz"], kmax_h, h(o["sst"]), cc_h["sw_flux"], diag_init=-7.0)
    want_m = oracle.macros_source_sink(mtr, kmax_h, diag_init=-7.0)
    got_own = h(own1).transpose(0, 2, 1)
    P.compare_tend(rep, "dms.tend", got_own[0:2], want_d["tend"][0:2])
    P.compare_tend(rep, "macros.tend", got_own[2:5], want_m["tend"][0:3])
    for nm, w in want_d["kc"].items():
        rep.note("diag", "dms.kc." + nm, P.e_norm(h(o["dms_diag"][nm]).T, w), P.TOL_DIAG)
    for nm, w in want_m["kc"].items():
        rep.note("diag", "macros.kc." + nm, P.e_norm(h(o["macros_diag"][nm]).T, w), P.TOL_DIAG)
    assert rep.ok(), "\n".join(rep.fail[:8])
    del b, o, tend1, own1, ph1
    torch.cuda.empty_cache()


def test_reference_layout_equals_native_layout_at_scale(gpu, monkeypatch):
    """The reference (level-fastest) layout on the device -- cell-parallel kernels on the caller's arrays, LDS transposes
    for the column kernels' planes, several column tiles -- must leave exactly the bits of the native layout in every
    output, including the zero fill below the sea floor and the untouched diagnostics."""
    torch, L, eng = gpu
    from ocean_bgc_amd.engine import Engine
    nz, nc = 60, 100_000
    b = make_batch(torch, L, eng, nz, nc, 777, diag=True)
    run(L, eng, b)
    monkeypatch.setenv("OBGC_KFAST_TILE", "32768")        # 4 tiles, the last one short
    eng2 = Engine(0)
    f64 = dict(dtype=torch.float64, device="cuda")
    t = lambda x: x.transpose(-1, -2).contiguous()         # [.., k, c] -> [.., c, k]
    tend = torch.full((30, nc, nz), -7.0, **f64)
    ph = torch.zeros((2, nc, nz), **f64)
    diag = {}
    for nm, v in b["diag"].items():
        diag[nm] = torch.full(tuple(t(v).shape) if v.dim() >= 2 and v.shape[-2] == nz else tuple(v.shape), -7.0, **f64)
    kc = {k: t(v) for k, v in b["kc"].items()}
    torch.cuda.synchronize()
    eng2.bgc_source_sink(nz, nc, L.OBGC_LAYOUT_KFAST, t(b["tr"]), kc["temp"], kc["salt"], kc["zmid"], kc["dz"], kc["zbot"],
                         b["cc"]["lat"], b["kmax"], kc["fesedflux"], b["cc"]["dust_flux_in"], b["cc"]["sw_flux"], tend, ph[0], ph[1],
                         diag, True)
    eng2.sync()
    assert torch.equal(t(tend), b["tend"]) and torch.equal(t(ph), b["ph"])
    for nm, v in b["diag"].items():
        got = t(diag[nm]) if v.dim() >= 2 and v.shape[-2] == nz else diag[nm]
        assert torch.equal(got, v), nm
