"""Diagnostics selected by name (obgc_diag_set_*): the names come from a parameter file's [diagnostics] section, the library
builds the pointer structs and the mean buffers itself, three steps are averaged on the device, and the means, downloaded
and written to a file, are numpy's mean of the ORACLE's fields of the same three steps.  Members that were not selected
are NULL in the structs the kernels get: never allocated, never written."""
import os

import numpy as np
import pytest

import parity as P
from conftest import has_gpu

pytestmark = pytest.mark.gpu

SELECTION = """
[bgc]
parm_Fe_bioavail = 0x1p-1
[diagnostics]
bgc = CO3, diag_pH_3D, photoC, NITRIF      # (k,column) and (k,column,autotroph) members
bgc = photoC_zint, Jint_Ctot, zsatcalc, O2_ZMIN_DEPTH, photoC_TOT_zint
dms = DMS_R_TOTAL
macros = PROT_S_TOTAL
"""


def _steps(synth, nc, nz, col0):
    syn = synth.generate(nc, nz, col0=col0)
    tr0 = synth.bgc_tracer_array(syn)
    out = []
    for i in range(3):
        tr = tr0 * (1.0 + 0.02 * i)
        s = dict(syn)
        out.append((s, tr, _dms(synth, s, tr), _macros(synth, s, tr)))
    return out


def _dms(synth, syn, tr):
    d = synth.dms_tracer_array(syn)
    from oracle import fields as F
    for j, nm in enumerate(F.DMS_TRACERS[2:], start=2):
        d[j] = tr[F.BGC_TRACERS.index(nm)]
    return d


def _macros(synth, syn, tr):
    m = synth.macros_tracer_array(syn)
    from oracle import fields as F
    for j, nm in enumerate(F.MACROS_TRACERS[3:], start=3):
        m[j] = tr[F.BGC_TRACERS.index(nm)]
    return m


def _oracle_means(oracle, steps):
    nc, nz = steps[0][0]["temp"].shape
    z = np.zeros((nc, nz))
    acc, last = {}, None
    for syn, tr, dtr, mtr in steps:
        b = oracle.bgc_source_sink(tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"],
                                   syn["dust"], syn["sw"], z, z)
        d = oracle.dms_source_sink(dtr, syn["dz"], syn["kmax"], syn["forcing"]["SST"], syn["sw"])
        m = oracle.macros_source_sink(mtr, syn["kmax"])
        cur = {}
        for grp in ("kc", "kca", "ca", "c"):
            for nm, v in b[grp].items():
                cur["bgc." + nm] = v
        cur.update({"dms." + nm: v for nm, v in d["kc"].items()})
        cur.update({"macros." + nm: v for nm, v in m["kc"].items()})
        for nm, v in cur.items():
            acc[nm] = acc.get(nm, 0.0) + v / 3.0
        last = cur
    return acc, last


@pytest.mark.parametrize("layout", ["colfast", "tiled"])
def test_selected_means_over_three_steps(pkg, oracle, synth, tmp_path, layout):
    if not has_gpu():
        pytest.fail("needs a GPU")
    import torch
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine, TileArena, DiagSet, diag_names_from_file, diag_file_read
    pfile = str(tmp_path / "params.txt")
    open(pfile, "w").write(SELECTION)
    names = diag_names_from_file(pfile)
    assert names[:4] == ["bgc.CO3", "bgc.pH_3D", "bgc.photoC", "bgc.NITRIF"] and "dms.DMS_R_TOTAL" in names and len(names) == 11
    rows, vecs = DiagSet.rows(names)
    assert (rows, vecs) == (1 + 1 + 4 + 1 + 1 + 1, 4 + 1 + 1 + 1 + 1)
    nc, nz = 333, 60
    steps = _steps(synth, nc, nz, 12_000)
    want_mean, want_last = _oracle_means(oracle, steps)
    eng = Engine(0)
    tiled = layout == "tiled"
    f64 = dict(dtype=torch.float64, device="cuda")
    if tiled:
        # the production profiles the column integrals read must be among the diagnostics in this layout
        names = names + ["bgc.photoC_NO3", "bgc.CaCO3_form", "bgc.bSi_form"]
        rows, _ = DiagSet.rows(names)
        A = TileArena(torch, "cuda:0", nc, nz, 30 + 6 + 5 + 30 + 2 + 5 + 2 * rows)
        KC = lambda n=1: A.take(n, (nc, nz) if n == 1 else None)
        g = dict(ld=A.ld, plane=64)
        LAY = L.OBGC_LAYOUT_TILED
    else:
        KC = lambda n=1: torch.zeros((nz, nc) if n == 1 else (n, nz, nc), **f64)
        g = {}
        LAY = L.OBGC_LAYOUT_COLFAST
    tr_d, own = KC(30), KC(5)
    fld = {nm: KC() for nm in ("temp", "salt", "zmid", "dz", "zbot", "fesed")}
    tend, own_tend, php, pha = KC(30), KC(5), KC(), KC()
    inst_rows = KC(rows) if tiled else None
    mean_rows = KC(rows) if tiled else None
    S = eng.diag_set(nz, nc, names, LAY, inst_rows=inst_rows, mean_rows=mean_rows, **g)
    assert len(S) == len(names)
    # unselected members are NULL in the structs the kernels see
    sel = {n.split(".", 1)[1] for n in names if n.startswith("bgc.")}
    for nm in L.BGC_DIAG_KC + L.BGC_DIAG_KCA + L.BGC_DIAG_CA + L.BGC_DIAG_C:
        assert bool(getattr(S.bgc.contents, nm)) == (nm in sel), nm
    assert [nm for nm in L.DMS_DIAG_KC if getattr(S.dms.contents, nm)] == ["DMS_R_TOTAL"]
    assert [nm for nm in L.MACROS_DIAG_KC if getattr(S.macros.contents, nm)] == ["PROT_S_TOTAL"]

    def put(dst, x, planes=None):
        if tiled:
            dst.put(x)
        else:
            dst.copy_(torch.from_numpy(np.ascontiguousarray(np.swapaxes(x, -1, -2))).cuda())

    def sub(x, j):
        return x.sub(j) if tiled else x[j]

    free0 = None
    for i, (syn, tr, dtr, mtr) in enumerate(steps):
        put(tr_d, tr)
        put(own, np.stack([dtr[0], dtr[1], mtr[0], mtr[1], mtr[2]]))
        for nm in fld:
            put(fld[nm], syn[nm])
        (php.zero_(), pha.zero_())
        cv = {nm: torch.from_numpy(np.ascontiguousarray(syn[nm])).cuda() for nm in ("lat", "dust", "sw")}
        kmax = torch.from_numpy(np.asarray(syn["kmax"], dtype=np.int32)).cuda()
        sst = torch.from_numpy(np.ascontiguousarray(syn["forcing"]["SST"])).cuda()
        torch.cuda.synchronize()
        if i == 1:
            free0 = torch.cuda.mem_get_info()[0]
        eng.fused_step(nz, nc, LAY, tr_d, fld["temp"], fld["salt"], fld["zmid"], fld["dz"], fld["zbot"], cv["lat"], kmax, fld["fesed"],
                       cv["dust"], cv["sw"], tend, php, pha, S.bgc, True, dms_diag=S.dms, macros_diag=S.macros, dms=sub(own, 0),
                       dmsp=sub(own, 1), SST=sst, dms_tend=sub(own_tend, 0), dmsp_tend=sub(own_tend, 1), prot=sub(own, 2), poly=sub(own, 3),
                       lip=sub(own, 4), prot_tend=sub(own_tend, 2), poly_tend=sub(own_tend, 3), lip_tend=sub(own_tend, 4), **g)
        S.accumulate(1.0 / 3.0, reset=(i == 0))
        if i == 1:
            eng.sync()
            assert torch.cuda.mem_get_info()[0] == free0      # no device memory for what was not selected, none per step
    rep = P.Report()
    base = [n for n in names if n not in ("bgc.photoC_NO3", "bgc.CaCO3_form", "bgc.bSi_form")]
    got = {}
    for n in base:
        got[n] = S.download(n)
        inst = S.download(n, "inst")
        assert got[n].shape == want_mean[n].shape, n
        if n in ("bgc.zsatcalc", "bgc.O2_ZMIN_DEPTH"):
            rep.note("diag", "inst " + n, P.e_norm(inst, want_last[n]), P.TOL_DIAG)
            continue                                         # sentinels / depths: the mean of -1 flags means nothing; the field does
        if n == "bgc.Jint_Ctot":
            scale = np.abs(want_mean["bgc.photoC_TOT_zint"]).max()
            rep.note("jint_abs", n, np.abs(got[n] - want_mean[n]).max() / scale, 1e-12)
            continue
        rep.note("diag", "mean " + n, P.e_norm(got[n], want_mean[n]), P.TOL_DIAG)
        rep.note("diag", "inst " + n, P.e_norm(inst, want_last[n]), P.TOL_DIAG)
    assert rep.ok(), "\n".join(rep.fail[:10])
    # the file: every selected member, the bits of the download
    path = str(tmp_path / "means.obgcdia")
    S.write(path)
    hdr, rec = diag_file_read(path)
    assert hdr == dict(nlev=nz, ncol=nc, nrec=len(names), which=0) and list(rec) == names
    for n in base:
        assert np.array_equal(rec[n], got[n]), n
    with pytest.raises(Exception, match="not in the selection"):
        S.download("bgc.HCO3")
    S.close()

