"""Parity bars shared by the GPU tests, smoke() and bench.py.

Per output array X:  e_norm = max|X - Xref| / max|Xref|   and
                     e_pt   = max over |Xref| > 1e-6 max|Xref| of |X - Xref| / |Xref|.

  tendencies (BGC, DMS, MACROS)              e_norm <= 1e-12, e_pt <= 1e-9
  PH_PREV_*, surface_pH*                     e_norm <= 1e-12
  every other fp diagnostic / flux           e_norm <= 1e-10
  sentinel-like diagnostics                  exact (zsat* == -1/0 pattern, O2_ZMIN_DEPTH)
  conservation integrals Jint_*              |diff| <= 1e-12 * (scale of the integrand): they
                                             are ~0 by construction (sources - sinks), so only an
                                             absolute bar is meaningful

The bars sit 3 orders above the reference's own FMA-vs-no-FMA spread (measured by building the reference with and without contraction).
"""
import numpy as np

TOL_TEND_NORM = 1e-12
TOL_TEND_PT = 1e-9
TOL_PH = 1e-12
TOL_DIAG = 1e-10
EXACT_C = ("O2_ZMIN_DEPTH",)
JINT = ("Jint_Ctot", "Jint_100m_Ctot", "Jint_Ntot", "Jint_100m_Ntot", "Jint_Ptot", "Jint_100m_Ptot", "Jint_Sitot",
        "Jint_100m_Sitot")


def e_norm(x, xref):
    xref = np.asarray(xref)
    m = np.abs(xref).max() if xref.size else 0.0
    d = np.abs(np.asarray(x) - xref).max() if xref.size else 0.0
    return d / m if m > 0 else d


def e_pt(x, xref, floor=1e-6):
    xref = np.asarray(xref)
    if xref.size == 0:
        return 0.0
    m = np.abs(xref).max()
    if m == 0:
        return 0.0
    sel = np.abs(xref) > floor * m
    if not sel.any():
        return 0.0
    return (np.abs(np.asarray(x)[sel] - xref[sel]) / np.abs(xref[sel])).max()


class Report:
    def __init__(self):
        self.worst = {}
        self.fail = []

    def note(self, cls, name, val, tol):
        w = self.worst.get(cls, (0.0, ""))
        if val >= w[0]:
            self.worst[cls] = (float(val), name)
        if not (val <= tol):
            self.fail.append("%s: %s = %.3e > %.1e" % (cls, name, val, tol))

    def ok(self):
        return not self.fail

    def summary(self):
        return {k: {"max": v[0], "at": v[1]} for k, v in self.worst.items()}


def compare_tend(rep, name, got, want):
    assert np.isfinite(got).all() or not np.isfinite(want).all(), name + " produced non-finite values"
    for i in range(want.shape[0]):
        rep.note("tend_e_norm", "%s[%d]" % (name, i), e_norm(got[i], want[i]), TOL_TEND_NORM)
        rep.note("tend_e_pt", "%s[%d]" % (name, i), e_pt(got[i], want[i]), TOL_TEND_PT)


def compare_bgc(got, want, rep=None, diagnostics=True, dz=None):
    rep = rep or Report()
    compare_tend(rep, "bgc.tend", got["tend"], want["tend"])
    rep.note("ph", "ph_prev", e_norm(got["ph_prev"], want["ph_prev"]), TOL_PH)
    rep.note("ph", "ph_prev_alt", e_norm(got["ph_prev_alt"], want["ph_prev_alt"]), TOL_PH)
    if not diagnostics:
        return rep
    for grp in ("kc", "kca", "ca"):
        for nm, w in want[grp].items():
            rep.note("diag", grp + "." + nm, e_norm(got[grp][nm], w), TOL_DIAG)
    scale = np.abs(want["c"]["photoC_TOT_zint"]).max() + np.abs(want["c"]["tot_CaCO3_form_zint"]).max() + 1e-300
    if dz is not None:
        # the integrand itself: sum over levels of dz * |tendencies| (needed when the carbon turnover of a column is
        # not tied to its photosynthesis, e.g. a dark bloom in tests/crafted.py)
        scale = max(scale, (np.abs(want["tend"]).sum(axis=0) * np.asarray(dz)).sum(axis=1).max())
    for nm, w in want["c"].items():
        g = got["c"][nm]
        if nm in EXACT_C:
            rep.note("exact", "c." + nm, float(not np.array_equal(g, w)), 0.0)
        elif nm in ("zsatcalc", "zsatarag"):
            # -1 / 0 sentinels must agree exactly; interpolated depths to the diagnostic bar
            sent = (w == -1.0) | (w == 0.0)
            rep.note("exact", "c." + nm + ".sentinel", float(not np.array_equal(g[sent], w[sent])), 0.0)
            if (~sent).any():
                rep.note("diag", "c." + nm, e_norm(g[~sent], w[~sent]), TOL_DIAG)
        elif nm in JINT:
            rep.note("jint_abs", "c." + nm, np.abs(g - w).max() / scale, 1e-12)
        else:
            rep.note("diag", "c." + nm, e_norm(g, w), TOL_DIAG)
    return rep


def compare_surface(got, want, rep=None):
    rep = rep or Report()
    for nm in ("surface_pH", "surface_pH_alt_co2"):
        rep.note("ph", nm, e_norm(got["forcing"][nm], want["forcing"][nm]), TOL_PH)
    rep.note("exact", "iceFraction", float(not np.array_equal(got["forcing"]["iceFraction"], want["forcing"]["iceFraction"])), 0.0)
    for nm in ("depositionFlux", "riverFlux", "seaIceFlux"):
        rep.note("exact", nm, float(not np.array_equal(got[nm], want[nm])), 0.0)
    for nm in ("gasFlux", "netFlux"):
        for i in range(want[nm].shape[0]):
            rep.note("diag", "%s[%d]" % (nm, i), e_norm(got[nm][i], want[nm][i]), TOL_DIAG)
    for nm, w in want["fdiag"].items():
        rep.note("diag", "fdiag." + nm, e_norm(got["fdiag"][nm], w), TOL_DIAG)
    return rep


def compare_simple(got, want, prefix, rep=None):
    """DMS / MACROS source-sink: tend + (k,c) diagnostics"""
    rep = rep or Report()
    compare_tend(rep, prefix + ".tend", got["tend"], want["tend"])
    for nm, w in want["kc"].items():
        rep.note("diag", "%s.kc.%s" % (prefix, nm), e_norm(got["kc"][nm], w), TOL_DIAG)
    return rep


def golden_bgc(g, prefix):
    """Re-assemble a result dictionary from a golden .npz"""
    out = dict(tend=g[prefix + ".tend"], ph_prev=g[prefix + ".ph_prev"], ph_prev_alt=g[prefix + ".ph_prev_alt"],
               kc={}, kca={}, ca={}, c={})
    for key in g:
        if key.startswith(prefix + "."):
            parts = key[len(prefix) + 1:].split(".", 1)
            if parts[0] in ("kc", "kca", "ca", "c") and len(parts) == 2:
                out[parts[0]][parts[1]] = g[key]
    return out
