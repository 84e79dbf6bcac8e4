"""CPU tests: C restatement vs the LIVE compiled reference (oracle/_ref/libbgc_ref.so).

Skipped automatically where the reference library is absent.  Wider than the golden
fixtures: thousands of columns, permuted tracer indices, non-default parameters.
"""
import numpy as np
import pytest

from conftest import e_norm
from oracle import fields as F

TOL = 1e-14


def assert_same(name, got, want):
    if np.array_equal(got, want):
        return
    err = e_norm(got, want)
    assert err <= TOL, "%s: e_norm %.3e" % (name, err)


def bgc_args(synth, syn, indices=None):
    tr = synth.bgc_tracer_array(syn, indices)
    return (tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"],
            syn["fesed"], syn["dust"], syn["sw"])


def compare_bgc(a, b):
    assert_same("tend", b["tend"], a["tend"])
    assert_same("ph_prev", b["ph_prev"], a["ph_prev"])
    assert_same("ph_prev_alt", b["ph_prev_alt"], a["ph_prev_alt"])
    for grp in ("kc", "kca", "ca", "c"):
        for nm in a[grp]:
            assert_same(grp + "." + nm, b[grp][nm], a[grp][nm])


@pytest.mark.parametrize("nz,nc,col0", [(60, 2500, 0), (100, 700, 4000), (7, 300, 100)])
def test_bgc_source_sink_cold_and_warm(ref, oracle, synth, nz, nc, col0):
    syn = synth.generate(nc, nz, col0=col0)
    args = bgc_args(synth, syn)
    z = np.zeros((nc, nz))
    a = ref.bgc_source_sink(*args, z, z, diag_init=-3.0)
    b = oracle.bgc_source_sink(*args, z, z, diag_init=-3.0)
    compare_bgc(a, b)
    tr2 = args[0] * (1.0 + 1e-3 * np.cos(np.arange(args[0].size).reshape(args[0].shape)))
    a2 = ref.bgc_source_sink(tr2, *args[1:], a["ph_prev"], a["ph_prev_alt"], alt_co2_use_eco=False)
    b2 = oracle.bgc_source_sink(tr2, *args[1:], b["ph_prev"], b["ph_prev_alt"], alt_co2_use_eco=False)
    compare_bgc(a2, b2)


def test_bgc_permuted_indices_and_parameters(synth):
    from oracle import orabind, refbind
    if not refbind.available():
        pytest.skip("reference library not built")
    rng = np.random.RandomState(7)
    perm = rng.permutation(30) + 1
    R = refbind.Ref(bgc_indices=perm, T0_Kelvin=273.16)
    O = orabind.Oracle(bgc_indices=perm, T0_Kelvin=273.16)
    p = R.bgc_get_params()
    assert np.array_equal(p["aint"], O.bgc_get_params()["aint"])   # BGC_init index derivation
    p["parms"] = p["parms"] * (1.0 + 0.1 * rng.rand(p["parms"].size))
    p["areal"] = p["areal"] * (1.0 + 0.05 * rng.rand(*p["areal"].shape))
    p["scalelen_vals"] = p["scalelen_vals"] * 1.1
    R.bgc_set_params(p)
    O.bgc_set_params(p)
    nc, nz = 600, 60
    syn = synth.generate(nc, nz, col0=12345)
    indices = {nm: int(perm[i]) for i, nm in enumerate(F.BGC_TRACERS)}
    args = bgc_args(synth, syn, indices)
    z = np.zeros((nc, nz))
    compare_bgc(R.bgc_source_sink(*args, z, z), O.bgc_source_sink(*args, z, z))
    f = syn["forcing"]
    flux = [syn[nm] for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")]
    s1, s2 = R.bgc_surface_fluxes(args[0], f, *flux), O.bgc_surface_fluxes(args[0], f, *flux)
    for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux", "netFlux"):
        assert_same(nm, s2[nm], s1[nm])
    # restore defaults for other tests sharing the (stateful) reference module variables
    refbind.Ref()


FLAG_VARIANTS = {
    "temperature functions swapped": {"temp_function": [2, 1, 2, 1]},
    "one grazee class": {"grazee_ind": [1, 1, 1, 1]},
    "all grazee classes distinct": {"grazee_ind": [1, 2, 3, 4]},
    "two N fixers, no calcifier": {"Nfixer": [0, 1, 1, 0], "imp_calcifier": [0, 0, 0, 0]},
}


def with_flags(p0, changes):
    p = {k: np.array(v, copy=True) for k, v in p0.items()}
    for name, vals in changes.items():
        p["aint"][:, F.AUTOTROPH_INTS.index(name)] = vals
    return p


@pytest.mark.parametrize("variant", sorted(FLAG_VARIANTS))
def test_bgc_autotroph_flag_variants(synth, variant):
    """A host may re-configure the autotroph table (which group fixes N2, calcifies, which share a grazee class, which
    temperature function they follow): the restatement must follow the reference bit for bit through those branches too."""
    from oracle import orabind, refbind
    if not refbind.available():
        pytest.skip("reference library not built")
    R, O = refbind.Ref(), orabind.Oracle()
    p = with_flags(R.bgc_get_params(), FLAG_VARIANTS[variant])
    R.bgc_set_params(p)
    O.bgc_set_params(p)
    nc, nz = 300, 60
    syn = synth.generate(nc, nz, col0=4242)
    args = bgc_args(synth, syn)
    z = np.zeros((nc, nz))
    a, b = R.bgc_source_sink(*args, z, z), O.bgc_source_sink(*args, z, z)
    compare_bgc(a, b)
    default = orabind.Oracle().bgc_source_sink(*args, z, z)
    assert not np.array_equal(default["tend"], b["tend"])        # the variant really takes another path
    refbind.Ref()                                               # restore the reference's module variables


def test_surface_dms_macros(ref, oracle, synth):
    nc, nz = 3000, 60
    syn = synth.generate(nc, nz, col0=777)
    tr = synth.bgc_tracer_array(syn)
    f = syn["forcing"]
    flux = [syn[nm] for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")]
    for kw in (dict(), dict(lcalc_o2=False), dict(lcalc_co2=False)):
        s1 = ref.bgc_surface_fluxes(tr, f, *flux, diag_init=-2.0, **kw)
        s2 = oracle.bgc_surface_fluxes(tr, f, *flux, diag_init=-2.0, **kw)
        for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux", "netFlux"):
            assert_same(nm, s2[nm], s1[nm])
        for nm in s1["forcing"]:
            assert_same("forcing." + nm, s2["forcing"][nm], s1["forcing"][nm])
        for nm in s1["fdiag"]:
            assert_same("fdiag." + nm, s2["fdiag"][nm], s1["fdiag"][nm])
    dtr = synth.dms_tracer_array(syn)
    d1 = ref.dms_source_sink(dtr, syn["dz"], syn["kmax"], f["SST"], syn["sw"], diag_init=-2.0)
    d2 = oracle.dms_source_sink(dtr, syn["dz"], syn["kmax"], f["SST"], syn["sw"], diag_init=-2.0)
    assert_same("dms.tend", d2["tend"], d1["tend"])
    for nm in d1["kc"]:
        assert_same("dms." + nm, d2["kc"][nm], d1["kc"][nm])
    e1, e2 = ref.dms_surface_fluxes(dtr, f, diag_init=-2.0), oracle.dms_surface_fluxes(dtr, f, diag_init=-2.0)
    assert_same("dms.net", e2["netFlux"], e1["netFlux"])
    for nm in e1["fdiag"]:
        assert_same("dms.fdiag." + nm, e2["fdiag"][nm], e1["fdiag"][nm])
    mtr = synth.macros_tracer_array(syn)
    m1, m2 = ref.macros_source_sink(mtr, syn["kmax"], diag_init=-2.0), oracle.macros_source_sink(mtr, syn["kmax"], diag_init=-2.0)
    assert_same("macros.tend", m2["tend"], m1["tend"])
    for nm in m1["kc"]:
        assert_same("macros." + nm, m2["kc"][nm], m1["kc"][nm])


def test_init_names_and_units(ref):
    names = ref.bgc_names()
    assert names[0] == ("PO4", "Dissolved Inorganic Phosphate", "mmol/m^3")
    assert names[8][2] == "meq/m^3" and names[17][2] == "mg/m^3"      # BGC_mod.F90:323-328
    assert names[23][0] == "diatSi" and names[19][0] == "spCaCO3"     # BGC_mod.F90:303-320
    assert ref.dms_names()[1][0] == "DMSP" and ref.macros_names()[2][0] == "LIP"


def dms_macros_case(synth, rng):
    """permuted DMS / MACROS tracer slots, parameters off their defaults by up to 10 %"""
    dperm, mperm = rng.permutation(14) + 1, rng.permutation(8) + 1
    nc, nz = 700, 60
    syn = synth.generate(nc, nz, col0=31337)
    dtr = synth.dms_tracer_array(syn, {nm: int(dperm[i]) for i, nm in enumerate(F.DMS_TRACERS)})
    mtr = synth.macros_tracer_array(syn, {nm: int(mperm[i]) for i, nm in enumerate(F.MACROS_TRACERS)})
    return dperm, mperm, syn, dtr, mtr


def test_dms_macros_permuted_slots_and_parameters(synth):
    from oracle import orabind, refbind
    if not refbind.available():
        pytest.skip("reference library not built")
    rng = np.random.RandomState(23)
    dperm, mperm, syn, dtr, mtr = dms_macros_case(synth, rng)
    R = refbind.Ref(dms_indices=dperm, macros_indices=mperm)
    O = orabind.Oracle(dms_indices=dperm, macros_indices=mperm)
    dp = R.dms_get_parms() * (1.0 + 0.1 * rng.rand(len(F.DMS_PARMS)))
    mp = R.macros_get_parms() * (1.0 + 0.1 * rng.rand(len(F.MACROS_PARMS)))
    R.dms_set_parms(dp); O.dms_set_parms(dp)
    R.macros_set_parms(mp); O.macros_set_parms(mp)
    f = syn["forcing"]
    d1 = R.dms_source_sink(dtr, syn["dz"], syn["kmax"], f["SST"], syn["sw"], diag_init=-2.0)
    d2 = O.dms_source_sink(dtr, syn["dz"], syn["kmax"], f["SST"], syn["sw"], diag_init=-2.0)
    assert_same("dms.tend", d2["tend"], d1["tend"])
    for nm in d1["kc"]:
        assert_same("dms." + nm, d2["kc"][nm], d1["kc"][nm])
    e1, e2 = R.dms_surface_fluxes(dtr, f, diag_init=-2.0), O.dms_surface_fluxes(dtr, f, diag_init=-2.0)
    assert_same("dms.net", e2["netFlux"], e1["netFlux"])
    m1, m2 = R.macros_source_sink(mtr, syn["kmax"], diag_init=-2.0), O.macros_source_sink(mtr, syn["kmax"], diag_init=-2.0)
    assert_same("macros.tend", m2["tend"], m1["tend"])
    for nm in m1["kc"]:
        assert_same("macros." + nm, m2["kc"][nm], m1["kc"][nm])
    default = orabind.Oracle(dms_indices=dperm, macros_indices=mperm)
    assert not np.array_equal(default.dms_source_sink(dtr, syn["dz"], syn["kmax"], f["SST"], syn["sw"])["tend"], d2["tend"])
    R2 = refbind.Ref()                                          # restore the reference's module variables ...
    R2.dms_set_parms(orabind.Oracle().dms_get_parms())          # ... including the parameters *_parms_init does not reset
    R2.macros_set_parms(orabind.Oracle().macros_get_parms())
