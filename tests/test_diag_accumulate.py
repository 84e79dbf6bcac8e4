"""Running time means of the diagnostics kept on the device (obgc_bgc_diag_accumulate).

No reference counterpart (the host model averages its diagnostics), so parity is unpinned by construction; the
statement checked is  mean <- (reset ? 0 : mean) + w * inst  per member, bit for bit against numpy, for every
member kind ((k,c), (k,c,a), (c,a), (c)), both layouts, a partial selection, and three real BGC_SourceSink steps.
"""
import numpy as np
import pytest

from conftest import has_gpu


@pytest.fixture(scope="module")
def engines(pkg):
    if not has_gpu():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False (no CPU fallback exists)")
    from ocean_bgc_amd.modules import ColumnEngine
    return {p: ColumnEngine(0, path=p) for p in ("device_colfast", "device_kfast")}


def _shapes(L, nc, nz):
    sh = {}
    sh.update({nm: ((nc, nz), "kc") for nm in L.BGC_DIAG_KC})
    sh.update({nm: ((4, nc, nz), "kc") for nm in L.BGC_DIAG_KCA})
    sh.update({nm: ((4, nc), "c") for nm in L.BGC_DIAG_CA})
    sh.update({nm: ((nc,), "c") for nm in L.BGC_DIAG_C})
    return sh


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["device_colfast", "device_kfast"])
def test_running_means_bit_exact(engines, pkg, path):
    from ocean_bgc_amd import _lib as L
    E = engines[path]
    nc, nz = 1000, 60
    rng = np.random.default_rng(11)
    sh = _shapes(L, nc, nz)
    skip = {"POC_ACCUM", "photoC", "Jint_Ctot", "photoC_zint"}            # a partial selection: these stay untouched
    mean_np = {nm: rng.standard_normal(s) for nm, (s, _) in sh.items()}
    mean_dev = {nm: (None if nm in skip else E._up(mean_np[nm], k)) for nm, (s, k) in sh.items()}
    for step, (w, reset) in enumerate(((0.25, True), (0.5, False), (0.125, False))):
        inst_np = {nm: rng.standard_normal(s) for nm, (s, _) in sh.items()}
        inst_dev = {nm: E._up(inst_np[nm], k) for nm, (s, k) in sh.items()}
        E.eng.bgc_diag_accumulate(nz, nc, E._layout, inst_dev, mean_dev, w, reset)
        for nm in sh:
            if nm not in skip:
                mean_np[nm] = (0.0 if reset else mean_np[nm]) + w * inst_np[nm]
    E.eng.sync()
    for nm, (s, k) in sh.items():
        if nm in skip:
            continue
        assert np.array_equal(E._down(mean_dev[nm], k), mean_np[nm]), (path, nm)


@pytest.mark.gpu
def test_means_of_real_steps(engines, pkg, synth):
    """Three BGC_SourceSink calls with the diagnostics left on the device, averaged there, downloaded once."""
    import torch
    from ocean_bgc_amd import _lib as L
    E = engines["device_colfast"]
    nc, nz = 700, 60
    syn = synth.generate(nc, nz, col0=3)
    tr0 = synth.bgc_tracer_array(syn)
    sh = _shapes(L, nc, nz)
    up = E._up
    common = [up(syn[k], "kc") for k in ("temp", "salt", "zmid", "dz", "zbot")]
    lat, kmax = up(syn["lat"], "c"), up(syn["kmax"].astype(np.int32), "c")
    fesed, dust, sw = up(syn["fesed"], "kc"), up(syn["dust"], "c"), up(syn["sw"], "c")
    inst = {nm: up(np.zeros(s), k) for nm, (s, k) in sh.items()}
    mean = {nm: up(np.zeros(s), k) for nm, (s, k) in sh.items()}
    tend = up(np.zeros((30, nc, nz)), "kc")
    php, pha = up(np.zeros((nc, nz)), "kc"), up(np.zeros((nc, nz)), "kc")
    want = None
    for step in range(3):
        tr = up(tr0 * (1.0 + 0.01 * step), "kc")
        E.eng.bgc_source_sink(nz, nc, E._layout, tr, *common, lat, kmax, fesed, dust, sw, tend, php, pha, inst, True)
        E.eng.bgc_diag_accumulate(nz, nc, E._layout, inst, mean, 1.0 / 3.0, step == 0)
        snap = {nm: E._down(inst[nm], k) for nm, (s, k) in sh.items()}
        want = {nm: (0.0 if step == 0 else want[nm]) + (1.0 / 3.0) * snap[nm] for nm in sh}
    torch.cuda.synchronize()
    for nm, (s, k) in sh.items():
        assert np.array_equal(E._down(mean[nm], k), want[nm]), nm
    assert np.abs(want["photoC_TOT_zint"]).max() > 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["device_colfast", "device_kfast"])
def test_dms_macros_and_free_arrays_bit_exact(engines, pkg, path):
    """obgc_dms_diag_accumulate, obgc_macros_diag_accumulate and obgc_accumulate_arrays (more pointers than one
    launch table holds) against numpy."""
    from ocean_bgc_amd import _lib as L
    E = engines[path]
    nc, nz = 333, 17
    rng = np.random.default_rng(5)
    for names, fn in ((L.DMS_DIAG_KC, E.eng.dms_diag_accumulate), (L.MACROS_DIAG_KC, E.eng.macros_diag_accumulate)):
        skip = {names[1]}
        mean_np = {nm: rng.standard_normal((nc, nz)) for nm in names}
        mean_dev = {nm: (None if nm in skip else E._up(mean_np[nm], "kc")) for nm in names}
        for w, reset in ((0.5, True), (0.25, False)):
            inst_np = {nm: rng.standard_normal((nc, nz)) for nm in names}
            fn(nz, nc, E._layout, {nm: E._up(inst_np[nm], "kc") for nm in names}, mean_dev, w, reset)
            for nm in names:
                mean_np[nm] = (0.0 if reset else mean_np[nm]) + w * inst_np[nm]
        E.eng.sync()
        for nm in names:
            if nm not in skip:
                assert np.array_equal(E._down(mean_dev[nm], "kc"), mean_np[nm]), (path, nm)
    # free-form lists: 200 planes + 30 vectors = more than one table of 176 pointers
    npl, nv = 200, 30
    pi = [rng.standard_normal((nc, nz)) for _ in range(npl)]
    pm = [rng.standard_normal((nc, nz)) for _ in range(npl)]
    vi = [rng.standard_normal(nc) for _ in range(nv)]
    vm = [rng.standard_normal(nc) for _ in range(nv)]
    planes = [(E._up(a, "kc"), E._up(b, "kc")) for a, b in zip(pi, pm)]
    vecs = [(E._up(a, "c"), E._up(b, "c")) for a, b in zip(vi, vm)]
    E.eng.accumulate_arrays(nz, nc, E._layout, planes, vecs, weight=0.375, reset=False)
    E.eng.accumulate_arrays(nz, nc, E._layout, planes[:3], (), weight=2.0, reset=True)
    E.eng.sync()
    for i in range(npl):
        want = 2.0 * pi[i] if i < 3 else pm[i] + 0.375 * pi[i]
        assert np.array_equal(E._down(planes[i][1], "kc"), want), (path, i)
    for i in range(nv):
        assert np.array_equal(E._down(vecs[i][1], "c"), vm[i] + 0.375 * vi[i]), (path, i)
    # nothing to do / empty batch are no-ops
    E.eng.accumulate_arrays(nz, nc, E._layout, (), (), weight=1.0)
    E.eng.accumulate_arrays(nz, 0, E._layout, (), (), weight=1.0)
