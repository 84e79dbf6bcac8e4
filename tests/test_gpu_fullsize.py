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
    # strided sample of the cold call, every output (the warm call below overwrites the diagnostics)
    cols = np.arange(0, nc, 397)
    got = dict(tend=tend1[:, :, cols].cpu().numpy().transpose(0, 2, 1), ph_prev=ph1[0][:, cols].cpu().numpy().T,
               ph_prev_alt=ph1[1][:, cols].cpu().numpy().T,
               kc={nm: b["diag"][nm][:, cols].cpu().numpy().T for nm in L.BGC_DIAG_KC},
               kca={nm: b["diag"][nm][:, :, cols].cpu().numpy().transpose(0, 2, 1) for nm in L.BGC_DIAG_KCA},
               ca={nm: b["diag"][nm][:, cols].cpu().numpy() for nm in L.BGC_DIAG_CA},
               c={nm: b["diag"][nm][cols].cpu().numpy() for nm in L.BGC_DIAG_C})
    # warm start from the converged state reproduces the pH to the reference solver's accuracy:
    # xacc = 1e-10 on [H+] ~ 1e-8 (co2calc.F90:47-53, "3 significant figures") -> |dpH| <~ 4e-3
    cold = ph1.clone()
    run(L, eng, b)       # b["ph"] now holds the cold solution -> narrow bracket
    live = ~below
    assert float((b["ph"][0][live] - cold[0][live]).abs().max()) < 5e-3
    # the sample against the oracle, diagnostics and column integrals included (inputs are bit-identical to the numpy generator)
    tr_h = b["tr"][:, :, cols].cpu().numpy().transpose(0, 2, 1)
    kc_h = {nm: v[:, cols].cpu().numpy().T for nm, v in b["kc"].items()}
    cc_h = {nm: v[cols].cpu().numpy() for nm, v in b["cc"].items()}
    z = np.zeros((cols.size, nz))
    want = oracle.bgc_source_sink(tr_h, kc_h["temp"], kc_h["salt"], kc_h["zmid"], kc_h["dz"], kc_h["zbot"], cc_h["lat"],
                                  b["kmax"][cols].cpu().numpy(), kc_h["fesedflux"], cc_h["dust_flux_in"], cc_h["sw_flux"], z, z,
                                  diag_init=-7.0)
    rep = P.compare_bgc(got, want)
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
    # strided sample against the oracle: every BGC output (diagnostics and column integrals included), DMS and MACROS
    # tendencies and diagnostics
    cols = np.arange(0, nc, 3989)
    sel = torch.as_tensor(cols, device="cuda")
    h = lambda x: x.index_select(-1, sel).cpu().numpy()
    tr_h = h(b["tr"]).transpose(0, 2, 1)
    kc_h = {nm: h(v).T for nm, v in b["kc"].items()}
    cc_h = {nm: h(v) for nm, v in b["cc"].items()}
    kmax_h = h(b["kmax"])
    z = np.zeros((cols.size, nz))
    want = oracle.bgc_source_sink(tr_h, kc_h["temp"], kc_h["salt"], kc_h["zmid"], kc_h["dz"], kc_h["zbot"], cc_h["lat"], kmax_h,
                                  kc_h["fesedflux"], cc_h["dust_flux_in"], cc_h["sw_flux"], z, z, diag_init=-7.0)
    got = dict(tend=h(tend1).transpose(0, 2, 1), ph_prev=h(ph1[0]).T, ph_prev_alt=h(ph1[1]).T,
               kc={nm: h(b["diag"][nm]).T for nm in L.BGC_DIAG_KC},
               kca={nm: h(b["diag"][nm]).transpose(0, 2, 1) for nm in L.BGC_DIAG_KCA},
               ca={nm: h(b["diag"][nm]) for nm in L.BGC_DIAG_CA}, c={nm: h(b["diag"][nm]) for nm in L.BGC_DIAG_C})
    rep = P.compare_bgc(got, want)
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


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config #5, one GPU's share: 1 M columns x 100 levels, every routine of the path (both *_SurfaceFluxes and
# BGC + DMS + MACROS in the fused step), every diagnostic, in the column-block layout the headline is timed in.
# ~205 GB of the 288 GB: nothing level-dependent is cloned whole.
def _gather(field, cols, torch):
    """columns `cols` of a column-block array -> numpy [planes, len(cols), nlev] (the reference's order)"""
    c = torch.as_tensor(cols, device="cuda")
    v = field.view()[c // 64, :, :, c % 64]                 # [ncols, nlev, planes]
    return v.permute(2, 0, 1).contiguous().cpu().numpy()


def _window(field, w0, wn, torch):
    """columns [w0, w0 + wn) of a column-block array -> device tensor [planes, nlev, wn]"""
    b0, b1 = w0 // 64, (w0 + wn + 63) // 64
    v = field.view()[b0:b1].permute(2, 1, 0, 3).reshape(field.nplanes, field.arena.nlev, (b1 - b0) * 64)
    return v[:, :, w0 - b0 * 64:w0 - b0 * 64 + wn].contiguous()


def make_tiled(torch, L, eng, nz, nc, col0, fill=-7.0):
    from ocean_bgc_amd.engine import TileArena
    nrows = 30 + 6 + 5 + 14 + 30 + 2 + 5 + len(L.BGC_DIAG_KC) + 4 * len(L.BGC_DIAG_KCA) + len(L.DMS_DIAG_KC) + len(L.MACROS_DIAG_KC)
    A = TileArena(torch, "cuda:0", nc, nz, nrows)
    f64 = dict(dtype=torch.float64, device="cuda")
    one = lambda: A.take(1, (nc, nz))
    b = dict(nz=nz, nc=nc, A=A, tr=A.take(30), kc={nm: one() for nm in ("temp", "salt", "zmid", "dz", "zbot", "fesedflux")},
             own=A.take(5), dtr=A.take(14))
    first_out = A.used
    b.update(tend=A.take(30), ph=(one(), one()), own_tend=A.take(5))
    d = {nm: one() for nm in L.BGC_DIAG_KC}
    d.update({nm: A.take(4) for nm in L.BGC_DIAG_KCA})
    b["ddiag"] = {nm: one() for nm in L.DMS_DIAG_KC}
    b["mdiag"] = {nm: one() for nm in L.MACROS_DIAG_KC}
    A.buf[:, :, first_out:, :].fill_(fill)
    b["ph"][0].zero_(); b["ph"][1].zero_()
    for nm in L.BGC_DIAG_CA:
        d[nm] = torch.full((4, nc), fill, **f64)
    for nm in L.BGC_DIAG_C:
        d[nm] = torch.full((nc,), fill, **f64)
    b["diag"] = d
    b["cc"] = {nm: torch.empty((nc,), **f64) for nm in ("lat", "dust_flux_in", "sw_flux")}
    b["kmax"] = torch.empty((nc,), dtype=torch.int32, device="cuda")
    b["sst"] = torch.empty((nc,), **f64)
    g = dict(ld=A.ld, plane=64)
    torch.cuda.synchronize()            # the fills above ran on torch's stream; the engine's streams do not wait for it
    eng.synth_fill(nz, nc, col0, tiled=True, tracers=b["tr"], kmax=b["kmax"], **g, **b["kc"], **b["cc"])
    own = b["own"]
    eng.synth_fill_extra(nz, nc, col0, tiled=True, ld=A.ld, dms=own.sub(0), dmsp=own.sub(1), prot=own.sub(2), poly=own.sub(3),
                         lip=own.sub(4), SST=b["sst"])
    # the 14-plane DMS tracer array DMS_SurfaceFluxes reads: own tracers + copies of the shared BGC tracers (DMS_mod.F90:471-482)
    dv, ov, tv = b["dtr"].view(), own.view(), b["tr"].view()
    dv[:, :, 0, :].copy_(ov[:, :, 0, :]); dv[:, :, 1, :].copy_(ov[:, :, 1, :])
    for j, nm in enumerate(L.DMS_TRACERS[2:], start=2):
        dv[:, :, j, :].copy_(tv[:, :, L.BGC_TRACERS.index(nm), :])
    torch.cuda.synchronize()
    return b, g


def run_tiled(L, eng, b, g, sf=None):
    kc, cc, own, t = b["kc"], b["cc"], b["own"], b["own_tend"]
    nz, nc, TL = b["nz"], b["nc"], L.OBGC_LAYOUT_TILED
    if sf is not None:      # the host model calls *_SurfaceFluxes first
        eng.bgc_surface_fluxes(nz, nc, TL, b["tr"], sf["forcing"], *[sf["flux"][nm] for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")],
                               sf["net"], sf["diag"], **g)
        eng.dms_surface_fluxes(nz, nc, TL, b["dtr"], sf["forcing"], sf["dms_net"], sf["dms_diag"], **g)
    eng.fused_step(nz, nc, TL, b["tr"], kc["temp"], kc["salt"], kc["zmid"], kc["dz"], kc["zbot"], cc["lat"], b["kmax"], kc["fesedflux"],
                   cc["dust_flux_in"], cc["sw_flux"], b["tend"], b["ph"][0], b["ph"][1], b["diag"], True, dms_diag=b["ddiag"],
                   macros_diag=b["mdiag"], dms=own.sub(0), dmsp=own.sub(1), SST=b["sst"], dms_tend=t.sub(0), dmsp_tend=t.sub(1),
                   prot=own.sub(2), poly=own.sub(3), lip=own.sub(4), prot_tend=t.sub(2), poly_tend=t.sub(3), lip_tend=t.sub(4), **g)
    eng.sync()


def test_cfg5_share_1M_x_100_every_routine(gpu, oracle, synth):
    torch, L, eng = gpu
    torch.cuda.empty_cache()
    nz, nc = 100, 1_000_000
    f64 = dict(dtype=torch.float64, device="cuda")
    b, g = make_tiled(torch, L, eng, nz, nc, 0)
    sf = make_surface(torch, L, eng, nz, nc, 0)
    sf["dms_net"] = torch.full((14, nc), -7.0, **f64)
    sf["dms_diag"] = {nm: torch.full((nc,), -7.0, **f64) for nm in L.DMS_FLUX_DIAG}
    run_tiled(L, eng, b, g, sf)
    assert eng.last_schedule() == 3
    A = b["A"]
    # finite everywhere, block range by block range (no temporary the size of the arena)
    for b0 in range(0, A.nblk, 1024):
        assert bool(torch.isfinite(A.buf[b0:b0 + 1024]).all()), "non-finite value in blocks %d.." % b0
    assert all(bool(torch.isfinite(v).all()) for v in sf["diag"].values()) and bool(torch.isfinite(sf["net"]).all())
    assert bool(torch.isfinite(sf["dms_net"][:2]).all())
    # the reference's zero fill below the floor and on land: tendencies of the three modules, pH, diagnostics
    kmx = b["kmax"].reshape(A.nblk, 1, 64)
    below = torch.arange(nz, device="cuda").reshape(1, nz, 1) >= kmx                       # [nblk, nlev, 64]
    tv = b["tend"].view()
    for n in range(30):
        assert bool((tv[:, :, n, :][below] == 0).all()), "tend %d" % n
    for n in range(5):
        assert bool((b["own_tend"].view()[:, :, n, :][below] == 0).all()), "own tend %d" % n
    for nm in ("photoC_TOT", "CO3", "POC_REMIN", "calcToSed", "Fe_scavenge"):
        assert bool((b["diag"][nm].view()[:, :, 0, :][below] == 0).all()), nm
    assert bool((b["diag"]["photoC"].view().permute(0, 1, 3, 2)[below] == 0).all())
    assert bool((b["ph"][0].view()[:, :, 0, :][below] == 0).all())
    assert bool((b["ddiag"]["DMS_R_TOTAL"].view()[:, :, 0, :][below] == -7.0).all())       # DMS / MACROS leave theirs alone
    assert bool((b["mdiag"]["PROT_S_TOTAL"].view()[:, :, 0, :][below] == -7.0).all())
    for nm in L.BGC_DIAG_KC_UNTOUCHED:
        assert bool((b["diag"][nm].view() == -7.0).all()), nm
    land = b["kmax"] == 0
    assert bool(land.any()) and bool((b["diag"]["O2_ZMIN"][land] == 0).all()) and bool((b["diag"]["zsatcalc"][land] == 0).all())
    # conservation of C, N, P, Si
    scale = float(b["diag"]["photoC_TOT_zint"].abs().max())
    for nm, fac in (("Jint_Ctot", 1.0), ("Jint_Ntot", 0.137), ("Jint_Ptot", 0.00855), ("Jint_Sitot", 0.137)):
        assert float(b["diag"][nm].abs().max()) <= 1e-10 * scale * fac, nm
    # the calcite lysocline at 3 300 m (BGC_mod.F90:2564) is crossed by the ragged sea floor of this workload
    zb_bottom = 1000.0 * (b["kmax"].double() + 0.025 * (b["kmax"].double() - 1.0) * b["kmax"].double())
    assert bool((zb_bottom[b["kmax"] > 1] < 3300.0e2).any()) and bool((zb_bottom >= 3300.0e2).any())
    # shard invariance: 1 000 columns (not a multiple of the block) generated and solved on their own, every output
    w0, wn = 765_433, 1_000
    s, gs = make_tiled(torch, L, eng, nz, wn, w0)
    ss = make_surface(torch, L, eng, nz, wn, w0)
    ss["dms_net"] = torch.full((14, wn), -7.0, **f64)
    ss["dms_diag"] = {nm: torch.full((wn,), -7.0, **f64) for nm in L.DMS_FLUX_DIAG}
    run_tiled(L, eng, s, gs, ss)
    for key in ("tend", "own_tend"):
        assert torch.equal(_window(s[key], 0, wn, torch), _window(b[key], w0, wn, torch)), key
    for i in (0, 1):
        assert torch.equal(_window(s["ph"][i], 0, wn, torch), _window(b["ph"][i], w0, wn, torch))
    for grp in ("diag", "ddiag", "mdiag"):
        for nm, v in s[grp].items():
            if hasattr(v, "view") and not torch.is_tensor(v):
                assert torch.equal(_window(v, 0, wn, torch), _window(b[grp][nm], w0, wn, torch)), nm
            else:
                assert torch.equal(v, b[grp][nm][..., w0:w0 + wn]), nm
    assert torch.equal(ss["net"], sf["net"][:, w0:w0 + wn]) and torch.equal(ss["dms_net"][:2], sf["dms_net"][:2, w0:w0 + wn])
    for nm, v in ss["forcing"].items():
        assert torch.equal(v, sf["forcing"][nm][w0:w0 + wn]), nm
    # determinism: the window of a second run (cold again) has the same bits
    keep = {key: _window(b[key], w0, wn, torch) for key in ("tend", "own_tend")}
    keep_c = {nm: b["diag"][nm].clone() for nm in ("Jint_Ctot", "zsatarag", "O2_ZMIN_DEPTH", "photoC_NO3_TOT_zint")}
    b["ph"][0].zero_(); b["ph"][1].zero_()
    torch.cuda.synchronize()
    run_tiled(L, eng, b, g)
    for key, v in keep.items():
        assert torch.equal(_window(b[key], w0, wn, torch), v), key
    for nm, v in keep_c.items():
        assert torch.equal(b["diag"][nm], v), nm
    del s, ss, keep
    # strided oracle sample: every BGC output incl. diagnostics and column integrals, DMS / MACROS tendencies and diagnostics
    cols = np.arange(3, nc, 3989)
    sel = torch.as_tensor(cols, device="cuda")
    tr_h = _gather(b["tr"], cols, torch)
    kc_h = {nm: _gather(v, cols, torch)[0] for nm, v in b["kc"].items()}
    cc_h = {nm: v[sel].cpu().numpy() for nm, v in b["cc"].items()}
    kmax_h = b["kmax"][sel].cpu().numpy()
    assert (kmax_h == 100).any() and (kmax_h < 97).any()                      # both sides of the lysocline
    z = np.zeros((cols.size, nz))
    want = oracle.bgc_source_sink(tr_h, kc_h["temp"], kc_h["salt"], kc_h["zmid"], kc_h["dz"], kc_h["zbot"], cc_h["lat"], kmax_h,
                                  kc_h["fesedflux"], cc_h["dust_flux_in"], cc_h["sw_flux"], z, z, diag_init=-7.0)
    got = dict(tend=_gather(b["tend"], cols, torch), ph_prev=_gather(b["ph"][0], cols, torch)[0], ph_prev_alt=_gather(b["ph"][1], cols, torch)[0],
               kc={nm: _gather(b["diag"][nm], cols, torch)[0] for nm in L.BGC_DIAG_KC},
               kca={nm: _gather(b["diag"][nm], cols, torch) for nm in L.BGC_DIAG_KCA},
               ca={nm: b["diag"][nm][:, sel].cpu().numpy() for nm in L.BGC_DIAG_CA},
               c={nm: b["diag"][nm][sel].cpu().numpy() for nm in L.BGC_DIAG_C})
    rep = P.compare_bgc(got, want)
    assert (want["kc"]["calcToSed"] > 0).any() and (want["kc"]["calcToSed"][kmax_h == 100] == 0).all()
    own_h = _gather(b["own"], cols, torch)
    named = {nm: tr_h[i] for i, nm in enumerate(L.BGC_TRACERS)}
    named.update(dms=own_h[0], dmsp=own_h[1], prot=own_h[2], poly=own_h[3], lip=own_h[4])
    dtr = np.stack([named[nm] for nm in L.DMS_TRACERS])
    assert np.array_equal(_gather(b["dtr"], cols, torch), dtr)
    mtr = np.stack([named[nm] for nm in L.MACROS_TRACERS])
    sst_h = b["sst"][sel].cpu().numpy()
    want_d = oracle.dms_source_sink(dtr, kc_h["dz"], kmax_h, sst_h, cc_h["sw_flux"], diag_init=-7.0)
    want_m = oracle.macros_source_sink(mtr, kmax_h, diag_init=-7.0)
    got_own = _gather(b["own_tend"], cols, torch)
    P.compare_tend(rep, "dms.tend", got_own[0:2], want_d["tend"][0:2])
    P.compare_tend(rep, "macros.tend", got_own[2:5], want_m["tend"][0:3])
    for nm, w in want_d["kc"].items():
        rep.note("diag", "dms.kc." + nm, P.e_norm(_gather(b["ddiag"][nm], cols, torch)[0], w), P.TOL_DIAG)
    for nm, w in want_m["kc"].items():
        rep.note("diag", "macros.kc." + nm, P.e_norm(_gather(b["mdiag"][nm], cols, torch)[0], w), P.TOL_DIAG)
    # DMS_SurfaceFluxes of the sample (the BGC surface call at 1 M columns is held to the oracle by the config-3 test)
    syn_cols = [synth.generate(1, nz, col0=int(c)) for c in cols[:48]]
    f_h = {nm: np.concatenate([sy["forcing"][nm] for sy in syn_cols]) for nm in syn_cols[0]["forcing"]}
    want_s = oracle.dms_surface_fluxes(dtr[:, :48], f_h)
    rep.note("diag", "dms.netFlux", P.e_norm(sf["dms_net"][:2, sel[:48]].cpu().numpy(), want_s["netFlux"][:2]), P.TOL_DIAG)
    assert rep.ok(), "\n".join(rep.fail[:8])
    del b, sf
    torch.cuda.empty_cache()
