"""obgc_dms_source_sink_noalloc (fixed library scratch, no device allocation) against obgc_dms_source_sink through
the C ABI: same kernels, so every output must be bit-identical -- on two calls in a row, the second with a smaller
input (the scratch array keeps the first call's contents), and after the size limit has been probed."""
import numpy as np
import pytest

from conftest import has_gpu

pytestmark = pytest.mark.gpu


def _batch(torch, L, eng, nz, nc, col0, ld):
    f64 = dict(dtype=torch.float64, device="cuda")
    tr = torch.empty((30, nz, nc), **f64)
    dz = torch.empty((nz, nc), **f64)
    temp = torch.empty((nz, nc), **f64)
    sw = torch.empty((nc,), **f64)
    kmax = torch.empty((nc,), dtype=torch.int32, device="cuda")
    eng.synth_fill(nz, nc, col0, tracers=tr, dz=dz, temp=temp, sw_flux=sw, kmax=kmax)
    own = torch.empty((2, nz, nc), **f64)
    eng.synth_fill_extra(nz, nc, col0, dms=own[0], dmsp=own[1])
    # DMS tracer array with a padded leading dimension `ld` >= nc: [14][nz][ld]
    dtr = torch.full((14, nz, ld), float("nan"), **f64)
    dtr[0, :, :nc] = own[0]
    dtr[1, :, :nc] = own[1]
    for j, nm in enumerate(L.DMS_TRACERS[2:], start=2):
        dtr[j, :, :nc] = tr[L.BGC_TRACERS.index(nm)]
    dzp = torch.full((nz, ld), float("nan"), **f64)
    dzp[:, :nc] = dz
    torch.cuda.synchronize()
    return dict(nz=nz, nc=nc, ld=ld, dtr=dtr, dz=dzp, kmax=kmax, sst=temp[0].contiguous(), sw=sw)


def _run(torch, L, eng, b, noalloc, diag):
    f64 = dict(dtype=torch.float64, device="cuda")
    tend = torch.full((14, b["nz"], b["ld"]), -7.0, **f64)
    d = {nm: torch.full((b["nz"], b["ld"]), -7.0, **f64) for nm in L.DMS_DIAG_KC} if diag else None
    torch.cuda.synchronize()
    eng.dms_source_sink(b["nz"], b["nc"], L.OBGC_LAYOUT_COLFAST, b["dtr"], b["dz"], b["kmax"], b["sst"], b["sw"], tend, d,
                        ld=b["ld"], plane=b["nz"] * b["ld"], noalloc=noalloc)
    eng.sync()
    return tend, d


def test_noalloc_entry_point_matches_the_allocating_one(pkg):
    if not has_gpu():
        pytest.fail("needs a GPU")
    import torch
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine, ObgcError
    eng = Engine(0)
    assert L.load().obgc_noalloc_max_cells() == 1 << 20
    # first call close to the limit (16 000 x 64 + padding = 1 026 0xx elements <= 2^20), second one much smaller
    for (nz, nc, ld, col0, diag) in ((64, 16_000, 16_064, 0, True), (60, 1_000, 1_000, 5_000, True), (60, 777, 832, 123, False)):
        b = _batch(torch, L, eng, nz, nc, col0, ld)
        want_t, want_d = _run(torch, L, eng, b, False, diag)
        got_t, got_d = _run(torch, L, eng, b, True, diag)
        live = torch.zeros((nz, ld), dtype=torch.bool, device="cuda")
        live[:, :nc] = True
        assert torch.isfinite(want_t[:, live]).all()
        assert torch.equal(got_t, want_t), (nz, nc)
        # something was computed: the DMS tendency is not the fill value and not all zero
        assert (want_t[0][live] != -7.0).all() and (want_t[0][live] != 0).any()
        if diag:
            for nm in L.DMS_DIAG_KC:
                assert torch.equal(got_d[nm], want_d[nm]), nm
    # beyond the fixed scratch: a clear error, nothing launched
    b = _batch(torch, L, eng, 60, 20_000, 0, 20_000)
    with pytest.raises(ObgcError, match="exceed the fixed scratch"):
        _run(torch, L, eng, b, True, False)
    # the reference layout is not served by this entry point
    with pytest.raises(ObgcError, match="native column-fastest layout only"):
        tend = torch.zeros((14, 100, 60), dtype=torch.float64, device="cuda")
        eng.dms_source_sink(60, 100, L.OBGC_LAYOUT_KFAST, torch.zeros((14, 100, 60), dtype=torch.float64, device="cuda"),
                            torch.ones((100, 60), dtype=torch.float64, device="cuda"), b["kmax"][:100].contiguous(),
                            b["sst"][:100].contiguous(), b["sw"][:100].contiguous(), tend, None, noalloc=True)
    # and a call after the rejected ones still works
    b = _batch(torch, L, eng, 60, 640, 9, 640)
    want_t, _ = _run(torch, L, eng, b, False, False)
    got_t, _ = _run(torch, L, eng, b, True, False)
    assert torch.equal(got_t, want_t)


def test_first_call_of_a_fresh_handle_allocates_nothing(pkg):
    """A handle whose FIRST call is the no-alloc entry point: the arena of the allocating entry point was never grown, the
    device's free memory does not move across the call, and the bits are those of the allocating entry point on another
    handle."""
    if not has_gpu():
        pytest.fail("needs a GPU")
    import torch
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine
    helper = Engine(0)                      # only generates the inputs (obgc_synth_fill needs a handle)
    b = _batch(torch, L, helper, 60, 4_000, 77, 4_032)
    f64 = dict(dtype=torch.float64, device="cuda")
    tend = torch.full((14, b["nz"], b["ld"]), -7.0, **f64)
    d = {nm: torch.full((b["nz"], b["ld"]), -7.0, **f64) for nm in L.DMS_DIAG_KC}
    # the HIP runtime uploads a kernel's code object on its first launch in a process: let ANOTHER handle pay for that
    small = _batch(torch, L, helper, 60, 64, 0, 64)
    _run(torch, L, helper, small, False, True)
    fresh = Engine(0)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    fresh.dms_source_sink(b["nz"], b["nc"], L.OBGC_LAYOUT_COLFAST, b["dtr"], b["dz"], b["kmax"], b["sst"], b["sw"], tend, d,
                          ld=b["ld"], plane=b["nz"] * b["ld"], noalloc=True)
    fresh.sync()
    assert torch.cuda.mem_get_info()[0] == free0, "the no-alloc entry point changed the device's free memory"
    want_t, want_d = _run(torch, L, Engine(0), b, False, True)     # the allocating entry point, on a third handle, afterwards
    assert torch.equal(tend, want_t)
    for nm in L.DMS_DIAG_KC:
        assert torch.equal(d[nm], want_d[nm]), nm
