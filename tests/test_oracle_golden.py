"""CPU tests: the C restatement (oracle/obgc_oracle.c) against the committed golden
fixtures produced by the UNMODIFIED reference (tests/golden/make_golden.py).

The oracle is compiled without FMA contraction exactly like the reference, and in
practice reproduces it bit for bit; the asserted bar is the one tests/parity.py states for
oracle-vs-reference (e_norm <= 1e-14), with integer-like fields exact.
"""
import os

import numpy as np
import pytest

from conftest import e_norm
from oracle import fields as F

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-14


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def check(name, got, want, tol=TOL):
    if np.array_equal(got, want):
        return
    assert np.isfinite(got).all() == np.isfinite(want).all(), name
    err = e_norm(got, want)
    assert err <= tol, "%s: e_norm %.3e > %.1e" % (name, err, tol)


def forcing_of(g, prefix):
    return {nm: g["%s.%s" % (prefix, nm)] for nm in F.BGC_SURF_FORCING}


@pytest.mark.parametrize("fixture", ["config1_1x60", "branch_cols_60", "branch_cols_80", "branch_cols_100"])
def test_bgc_source_sink_matches_golden(oracle, fixture):
    g = load(fixture)
    nc, nz = g["in.temp"].shape
    z = np.zeros((nc, nz))
    common = (g["in.temp"], g["in.salt"], g["in.zmid"], g["in.dz"], g["in.zbot"], g["in.lat"], g["in.kmax"],
              g["in.fesed"], g["in.dust"], g["in.sw"])
    a = oracle.bgc_source_sink(g["in.bgc_tracers"], *common, z, z, alt_co2_use_eco=True, diag_init=-7.0)
    b = oracle.bgc_source_sink(g["in.bgc_tracers_warm"], *common, g["bgc_cold.ph_prev"],
                               g["bgc_cold.ph_prev_alt"] + 0.01 * (g["bgc_cold.ph_prev_alt"] != 0),
                               alt_co2_use_eco=False, diag_init=-7.0)
    for pre, res in (("bgc_cold", a), ("bgc_warm", b)):
        check(pre + ".tend", res["tend"], g[pre + ".tend"])
        check(pre + ".ph_prev", res["ph_prev"], g[pre + ".ph_prev"])
        check(pre + ".ph_prev_alt", res["ph_prev_alt"], g[pre + ".ph_prev_alt"])
        for grp in ("kc", "kca", "ca", "c"):
            for nm, v in res[grp].items():
                check("%s.%s.%s" % (pre, grp, nm), v, g["%s.%s.%s" % (pre, grp, nm)])
    # quirk 10: three diagnostics are never touched -> still the sentinel
    for nm in F.BGC_DIAG_KC_UNTOUCHED:
        assert (a["kc"][nm] == -7.0).all()
    # warm call had alt_co2_use_eco = .false. -> DIC_ALT tendency identically 0 (BGC_mod.F90:1741-1745)
    assert (b["tend"][7] == 0.0).all()


@pytest.mark.parametrize("fixture", ["config1_1x60", "branch_cols_60"])
def test_bgc_surface_fluxes_matches_golden(oracle, fixture):
    g = load(fixture)
    flux = [g["in." + nm] for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")]
    s1 = oracle.bgc_surface_fluxes(g["in.bgc_tracers"], forcing_of(g, "in.forcing"), *flux, diag_init=-7.0)
    s2 = oracle.bgc_surface_fluxes(g["in.bgc_tracers_warm"], s1["forcing"], s1["depositionFlux"], s1["riverFlux"],
                                   s1["gasFlux"], s1["seaIceFlux"], diag_init=-7.0)
    for pre, s in (("surf_cold", s1), ("surf_warm", s2)):
        for nm, v in s["forcing"].items():
            check("%s.forcing.%s" % (pre, nm), v, g["%s.forcing.%s" % (pre, nm)])
        for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux", "netFlux"):
            check("%s.%s" % (pre, nm), s[nm], g["%s.%s" % (pre, nm)])
        for nm, v in s["fdiag"].items():
            check("%s.fdiag.%s" % (pre, nm), v, g["%s.fdiag.%s" % (pre, nm)])
    # in-place clamps (BGC_mod.F90:2837-2838)
    assert s1["forcing"]["iceFraction"].min() >= 0.0 and s1["forcing"]["iceFraction"].max() <= 1.0


@pytest.mark.parametrize("fixture", ["config1_1x60", "branch_cols_60", "branch_cols_80", "branch_cols_100"])
def test_dms_macros_match_golden(oracle, fixture):
    g = load(fixture)
    f = forcing_of(g, "in.forcing")
    d = oracle.dms_source_sink(g["in.dms_tracers"], g["in.dz"], g["in.kmax"], f["SST"], g["in.sw"], diag_init=-7.0)
    check("dms.tend", d["tend"], g["dms.tend"])
    for nm, v in d["kc"].items():
        check("dms.kc." + nm, v, g["dms.kc." + nm])
    ds = oracle.dms_surface_fluxes(g["in.dms_tracers"], f, diag_init=-7.0)
    check("dms_surf.netFlux", ds["netFlux"], g["dms_surf.netFlux"])
    for nm, v in ds["fdiag"].items():
        check("dms_surf.fdiag." + nm, v, g["dms_surf.fdiag." + nm])
    m = oracle.macros_source_sink(g["in.macros_tracers"], g["in.kmax"], diag_init=-7.0)
    check("macros.tend", m["tend"], g["macros.tend"])
    for nm, v in m["kc"].items():
        check("macros.kc." + nm, v, g["macros.kc." + nm])
    # only 2 of 14 / 3 of 8 tendency planes are ever non-zero (DMS_mod.F90:718-720, MACROS_mod.F90:388-392)
    assert (d["tend"][2:] == 0).all() and (m["tend"][3:] == 0).all()


def test_known_answers(oracle):
    g = load("kat")
    # the only check value printed in the reference tree: BGC_mod.F90:3028-3029
    assert abs(g["O2SAT_10C_35psu"][0] - 282.015) < 5e-4
    k = g["grid.k"].astype(np.int32)
    args = (k, g["grid.depth"], g["grid.temp"], g["grid.salt"], g["grid.dic"], g["grid.ta"], g["grid.pt"], g["grid.sit"])
    n = k.size
    c = oracle.comp_co3terms(*args, np.full(n, 6.0), np.full(n, 9.0))
    w = oracle.comp_co3terms(*args, g["co3terms_cold.ph"] - 0.2, g["co3terms_cold.ph"] + 0.2)
    gr = oracle.comp_co3terms(*args, np.full(n, 8.9), np.full(n, 9.0))
    for nm in ("ph", "h2co3", "hco3", "co3"):
        check("co3terms_cold." + nm, c[nm], g["co3terms_cold." + nm])
        check("co3terms_warm." + nm, w[nm], g["co3terms_warm." + nm])
        check("co3terms_grow." + nm, gr[nm], g["co3terms_grow." + nm])
    s = oracle.comp_co3_sat_vals(k, g["grid.depth"], g["grid.temp"], g["grid.salt"])
    check("co3_sat.calc", s["calc"], g["co3_sat.calc"])
    check("co3_sat.arag", s["arag"], g["co3_sat.arag"])
    one = oracle.co2calc_1point(g["grid.depth"], g["grid.temp"], g["grid.salt"], g["grid.dic"], g["grid.ta"],
                                g["grid.pt"], g["grid.sit"], np.full(n, 7.0), np.full(n, 9.0), np.full(n, 400.0),
                                np.full(n, 1.0))
    for nm, v in one.items():
        check("co2calc_1point." + nm, v, g["co2calc_1point." + nm])
    # Newton work: warm bracket needs far fewer talk_row evaluations than the cold one
    assert w["talk_calls"] < c["talk_calls"]


def test_parameter_tables_match_reference_dump(oracle):
    g = load("params")
    p = oracle.bgc_get_params()
    for key in ("areal", "aint", "parms", "scalelen_z", "scalelen_vals"):
        assert np.array_equal(p[key], g[key]), key
    assert np.array_equal(oracle.dms_get_parms(), g["dms_parms"])
    assert np.array_equal(oracle.macros_get_parms(), g["macros_parms"])
    # single-precision literals widened to double
    assert g["consts"][0] == np.float64(np.float32(1.0e-8))
    assert g["consts"][1] == np.float64(np.float32(3.17e-8))
