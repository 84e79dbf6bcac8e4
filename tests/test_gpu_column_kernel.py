"""The column-block kernel (attenuation, PAR scan, cell terms and particle sweep of the native layout in ONE launch, light
and hand-off in LDS) against the separate kernels it replaces (OBGC_COLUMN_KERNEL=0, the form the reference layout still
takes): same statements in the same order, so EVERY output must be bit-identical -- with and without diagnostics, with
the DMS / MACROS terms of the fused step, with the carbonate solve inside or beside it, for column counts that do not
fill the last 64-column block and level counts that do not fill the last 4-level chunk."""
import numpy as np
import pytest

from conftest import has_gpu

pytestmark = pytest.mark.gpu


def _engines(monkeypatch, sched=None):
    from ocean_bgc_amd.modules import ColumnEngine
    if sched is not None:
        monkeypatch.setenv("OBGC_SCHED", str(sched))
    monkeypatch.setenv("OBGC_COLUMN_KERNEL", "0")
    old = ColumnEngine(0, path="device_colfast")
    monkeypatch.setenv("OBGC_COLUMN_KERNEL", "1")
    new = ColumnEngine(0, path="device_colfast")
    return old, new


def _same(a, b, what):
    for key in ("tend", "ph_prev", "ph_prev_alt"):
        if key in a:
            assert np.array_equal(a[key], b[key]), "%s %s" % (what, key)
    for grp in ("kc", "kca", "ca", "c"):
        for nm, v in a.get(grp, {}).items():
            assert np.array_equal(v, b[grp][nm]), "%s %s.%s" % (what, grp, nm)


@pytest.mark.parametrize("nc,nz,diag,sched", [(20_000, 60, True, None), (5_003, 80, False, None), (777, 62, True, 0), (4_096, 100, True, 2),
                                              (130, 61, False, 1), (1, 60, True, None)])
def test_bgc_source_sink_bit_identical(pkg, synth, monkeypatch, nc, nz, diag, sched):
    if not has_gpu():
        pytest.fail("needs a GPU")
    old, new = _engines(monkeypatch, sched)
    syn = synth.generate(nc, nz, col0=424_242)
    tr = synth.bgc_tracer_array(syn)
    args = (tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"], syn["dust"],
            syn["sw"])
    z = np.zeros((nc, nz))
    a = old.BGC_SourceSink(*args, z, z, diag_init=-5.0, diagnostics=diag)
    b = new.BGC_SourceSink(*args, z, z, diag_init=-5.0, diagnostics=diag)
    assert old.eng.last_schedule() in (0, 1, 2) and new.eng.last_schedule() in (3, 4)
    _same(a, b, "cold")
    assert np.isfinite(b["tend"]).all() and (b["tend"] != -5.0).all()
    a2 = old.BGC_SourceSink(tr * 1.003, *args[1:], a["ph_prev"], a["ph_prev_alt"], alt_co2_use_eco=False, diag_init=-5.0, diagnostics=diag)
    b2 = new.BGC_SourceSink(tr * 1.003, *args[1:], a["ph_prev"], a["ph_prev_alt"], alt_co2_use_eco=False, diag_init=-5.0, diagnostics=diag)
    _same(a2, b2, "warm")


@pytest.mark.parametrize("nc,nz,diag", [(10_000, 80, True), (3_001, 60, False)])
def test_fused_step_bit_identical(pkg, synth, monkeypatch, nc, nz, diag):
    if not has_gpu():
        pytest.fail("needs a GPU")
    old, new = _engines(monkeypatch)
    syn = synth.generate(nc, nz, col0=77)
    t = syn["tracers"]
    rng = np.random.RandomState(9)
    t["spFe"] = np.where(rng.rand(nc, nz) < 0.002, 0.0, t["spFe"])      # columns where DMS cannot share BGC's light
    t["spFe"][5::10, 1] = 0.0
    tr, dtr, mtr = synth.bgc_tracer_array(syn), synth.dms_tracer_array(syn), synth.macros_tracer_array(syn)
    args = (tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"], syn["dust"],
            syn["sw"])
    z = np.zeros((nc, nz))
    sst = syn["forcing"]["SST"]
    a = old.Fused_SourceSink(*args, z, z, sst, dtr, mtr, diag_init=-5.0, diagnostics=diag)
    b = new.Fused_SourceSink(*args, z, z, sst, dtr, mtr, diag_init=-5.0, diagnostics=diag)
    for part in ("bgc", "dms", "macros"):
        _same(a[part], b[part], part)
    # and against the stand-alone DMS / MACROS entry points
    d = new.DMS_SourceSink(dtr, syn["dz"], syn["kmax"], sst, syn["sw"], diag_init=-5.0)
    m = new.MACROS_SourceSink(mtr, syn["kmax"], diag_init=-5.0)
    assert np.array_equal(d["tend"], b["dms"]["tend"]) and np.array_equal(m["tend"], b["macros"]["tend"])
    if diag:
        _same(d, b["dms"], "dms alone")
        _same(m, b["macros"], "macros alone")


def test_crafted_columns_bit_identical(pkg, synth, monkeypatch):
    """bracket growth, surviving dust deficit, empty column, iron clamp (tests/crafted.py)"""
    if not has_gpu():
        pytest.fail("needs a GPU")
    import crafted
    old, new = _engines(monkeypatch)
    syn, php = crafted.crafted_columns(synth)
    args = (synth.bgc_tracer_array(syn), syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"],
            syn["fesed"], syn["dust"], syn["sw"])
    _same(old.BGC_SourceSink(*args, php, php.copy(), diag_init=-5.0), new.BGC_SourceSink(*args, php, php.copy(), diag_init=-5.0), "crafted")


@pytest.mark.parametrize("knob", ["default", "OBGC_KFAST_HYBRID=0", "OBGC_KFAST_DIRECT=1"])
def test_reference_layout_forms_match_native(pkg, synth, monkeypatch, knob):
    """Every way the reference (level-fastest) layout is served on the device -- cell kernels on the caller's arrays with
    LDS transposes for the column kernels (default), a whole native workspace per column tile (OBGC_KFAST_HYBRID=0, which
    runs the column-block kernel on the workspace) and generic strides (OBGC_KFAST_DIRECT=1) -- gives the native layout's
    bits, over several column tiles with a short last one."""
    if not has_gpu():
        pytest.fail("needs a GPU")
    from ocean_bgc_amd.modules import ColumnEngine
    nc, nz = 2_500, 61
    syn = synth.generate(nc, nz, col0=31_337)
    tr = synth.bgc_tracer_array(syn)
    args = (tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"], syn["dust"],
            syn["sw"])
    z = np.zeros((nc, nz))
    native = ColumnEngine(0, path="device_colfast").BGC_SourceSink(*args, z, z, diag_init=-5.0)
    monkeypatch.setenv("OBGC_KFAST_TILE", "1024")
    if knob != "default":
        k, v = knob.split("=")
        monkeypatch.setenv(k, v)
    got = ColumnEngine(0, path="device_kfast").BGC_SourceSink(*args, z, z, diag_init=-5.0)
    _same(native, got, knob)


@pytest.mark.parametrize("nz", [1, 2, 3, 5, 9])
def test_tiny_columns_and_partial_blocks(pkg, synth, monkeypatch, nz):
    """Level counts below, at and just above one 4-level chunk; column counts around one 64-column block; land columns
    (kmax = 0) in between: the chunk loop, the rotating walk and the masked lanes against the separate kernels, bit for bit,
    BGC alone and with the DMS / MACROS terms."""
    if not has_gpu():
        pytest.fail("needs a GPU")
    old, new = _engines(monkeypatch)
    for nc in (1, 63, 65, 130):
        syn = synth.generate(nc, nz, col0=900 + nc)
        syn["kmax"][::7] = 0                                  # land
        tr, dtr, mtr = synth.bgc_tracer_array(syn), synth.dms_tracer_array(syn), synth.macros_tracer_array(syn)
        args = (tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"], syn["dust"],
                syn["sw"])
        z = np.zeros((nc, nz))
        _same(old.BGC_SourceSink(*args, z, z, diag_init=-5.0), new.BGC_SourceSink(*args, z, z, diag_init=-5.0), "bgc %d x %d" % (nc, nz))
        sst = syn["forcing"]["SST"]
        a = old.Fused_SourceSink(*args, z, z, sst, dtr, mtr, diag_init=-5.0)
        b = new.Fused_SourceSink(*args, z, z, sst, dtr, mtr, diag_init=-5.0)
        for part in ("bgc", "dms", "macros"):
            _same(a[part], b[part], "%s %d x %d" % (part, nc, nz))


@pytest.mark.parametrize("column_kernel", ["1", "0"])
def test_garbage_below_the_sea_floor_changes_nothing(pkg, synth, monkeypatch, column_kernel):
    """Cells below the sea floor run along in the wavefronts that straddle it (one store pass, zeros for them): whatever a
    host leaves in its arrays there -- NaN, infinities, zero thickness, negative concentrations -- must not reach any
    output.  The reference never reads those cells (its loops stop at number_of_active_levels, BGC_mod.F90:733)."""
    if not has_gpu():
        pytest.fail("needs a GPU")
    from ocean_bgc_amd.modules import ColumnEngine
    monkeypatch.setenv("OBGC_COLUMN_KERNEL", column_kernel)
    E = ColumnEngine(0, path="device_colfast")
    nc, nz = 1_500, 62
    syn = synth.generate(nc, nz, col0=4_711)
    syn["kmax"][::9] = 0
    syn["kmax"][1::9] = np.minimum(syn["kmax"][1::9], 7)
    below = np.arange(nz)[None, :] >= syn["kmax"][:, None]
    rng = np.random.RandomState(3)
    junk = np.array([np.nan, np.inf, -np.inf, 0.0, -1.0e300, 1.0e300])

    def run(s):
        tr, dtr, mtr = synth.bgc_tracer_array(s), synth.dms_tracer_array(s), synth.macros_tracer_array(s)
        z = np.zeros((nc, nz))
        return E.Fused_SourceSink(tr, s["temp"], s["salt"], s["zmid"], s["dz"], s["zbot"], s["lat"], s["kmax"], s["fesed"], s["dust"],
                                  s["sw"], z, z, s["forcing"]["SST"], dtr, mtr, diag_init=-5.0)

    clean = run(syn)
    dirty = dict(syn)
    dirty["tracers"] = {k: np.where(below, junk[rng.randint(0, junk.size, below.shape)], v) for k, v in syn["tracers"].items()}
    for nm in ("temp", "salt", "zmid", "dz", "zbot", "fesed", "dms", "dmsp", "prot", "poly", "lip"):
        dirty[nm] = np.where(below, junk[rng.randint(0, junk.size, below.shape)], syn[nm])
    got = run(dirty)
    for part in ("bgc", "dms", "macros"):
        _same(clean[part], got[part], "%s (column kernel %s)" % (part, column_kernel))
    assert np.isfinite(got["bgc"]["tend"]).all() and np.isfinite(got["dms"]["tend"]).all() and np.isfinite(got["macros"]["tend"]).all()


def test_small_batches_take_the_separate_kernels_by_default(pkg, synth, monkeypatch):
    """Without OBGC_COLUMN_MIN_CELLS (this suite sets it to 0) a native-layout call below 700 000 cells runs on the separate
    kernels, which spread a small batch over the whole device, and a larger one on the column-block kernel; either way the
    same bits, for BGC_SourceSink and for the fused step."""
    if not has_gpu():
        pytest.fail("needs a GPU")
    from ocean_bgc_amd.modules import ColumnEngine
    from oracle import fields as F
    monkeypatch.delenv("OBGC_COLUMN_MIN_CELLS", raising=False)
    monkeypatch.delenv("OBGC_COLUMN_KERNEL", raising=False)
    auto = ColumnEngine(0, path="device_colfast")
    monkeypatch.setenv("OBGC_COLUMN_KERNEL", "2")
    forced = ColumnEngine(0, path="device_colfast")
    for nc, nz, small in ((2_000, 60, True), (11_000, 64, False)):
        syn = synth.generate(nc, nz, col0=9_000)
        tr, dtr, mtr = synth.bgc_tracer_array(syn), synth.dms_tracer_array(syn), synth.macros_tracer_array(syn)
        args = (tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"], syn["dust"],
                syn["sw"])
        z = np.zeros((nc, nz))
        a = auto.BGC_SourceSink(*args, z, z, diag_init=-5.0)
        assert (auto.eng.last_schedule() in (0, 1, 2)) == small and (auto.eng.last_schedule() in (3, 4)) == (not small)
        b = forced.BGC_SourceSink(*args, z, z, diag_init=-5.0)
        assert forced.eng.last_schedule() in (3, 4)
        _same(a, b, "bgc %d" % nc)
        sst = syn["forcing"]["SST"]
        fa = auto.Fused_SourceSink(*args, z, z, sst, dtr, mtr, diag_init=-5.0)
        assert (auto.eng.last_schedule() in (0, 1, 2)) == small
        fb = forced.Fused_SourceSink(*args, z, z, sst, dtr, mtr, diag_init=-5.0)
        for part in ("bgc", "dms", "macros"):
            _same(fa[part], fb[part], "fused %s %d" % (part, nc))


@pytest.mark.parametrize("diagnostics", [True, False])
def test_small_batch_sweep_in_three_launches_same_bits(pkg, synth, monkeypatch, diagnostics):
    """Small native-layout batches run the particle sweep as decay factors per cell, the recurrence per column and the rest
    of the level per cell (bgc_walk_kernel): the bits of the one-kernel sweep and of the column-block kernel, for
    BGC_SourceSink and the fused step, cold and warm, on a ragged column count with empty columns in it."""
    if not has_gpu():
        pytest.fail("needs a GPU")
    from ocean_bgc_amd.modules import ColumnEngine
    monkeypatch.setenv("OBGC_COLUMN_KERNEL", "0")
    monkeypatch.setenv("OBGC_SPLIT_SWEEP", "0")
    whole = ColumnEngine(0, path="device_colfast")
    monkeypatch.setenv("OBGC_SPLIT_SWEEP", "2")
    split = ColumnEngine(0, path="device_colfast")
    monkeypatch.setenv("OBGC_COLUMN_KERNEL", "2")
    column = ColumnEngine(0, path="device_colfast")
    for nc, nz in ((777, 60), (5_001, 80)):
        syn = synth.generate(nc, nz, col0=31_000)
        syn["kmax"][5::97] = 0                       # land columns
        syn["kmax"][11::89] = 1                      # one active level: the surface is the sea floor
        tr, dtr, mtr = synth.bgc_tracer_array(syn), synth.dms_tracer_array(syn), synth.macros_tracer_array(syn)
        args = (tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"], syn["dust"],
                syn["sw"])
        z = np.zeros((nc, nz))
        a = whole.BGC_SourceSink(*args, z, z, diag_init=-5.0, diagnostics=diagnostics)
        b = split.BGC_SourceSink(*args, z, z, diag_init=-5.0, diagnostics=diagnostics)
        c = column.BGC_SourceSink(*args, z, z, diag_init=-5.0, diagnostics=diagnostics)
        assert whole.eng.last_schedule() in (0, 1, 2) and split.eng.last_schedule() in (0, 1, 2) and column.eng.last_schedule() in (3, 4)
        _same(a, b, "bgc %d" % nc)
        _same(c, b, "bgc %d against the column-block kernel" % nc)
        assert np.isfinite(b["tend"]).all() and np.abs(b["tend"]).max() > 0
        # warm second step on moved tracers
        a2 = whole.BGC_SourceSink(tr * 1.003, *args[1:], a["ph_prev"], a["ph_prev_alt"], diag_init=-5.0, diagnostics=diagnostics)
        b2 = split.BGC_SourceSink(tr * 1.003, *args[1:], b["ph_prev"], b["ph_prev_alt"], diag_init=-5.0, diagnostics=diagnostics)
        _same(a2, b2, "bgc warm %d" % nc)
        sst = syn["forcing"]["SST"]
        fa = whole.Fused_SourceSink(*args, z, z, sst, dtr, mtr, diag_init=-5.0, diagnostics=diagnostics)
        fb = split.Fused_SourceSink(*args, z, z, sst, dtr, mtr, diag_init=-5.0, diagnostics=diagnostics)
        for part in ("bgc", "dms", "macros"):
            if diagnostics:
                _same(fa[part], fb[part], "fused %s %d" % (part, nc))
            else:
                assert np.array_equal(fa[part]["tend"], fb[part]["tend"]), (part, nc)
        _same(fb["bgc"], b, "fused bgc against the BGC call %d" % nc)
