"""The fused step under HIP graph capture, through the C ABI: a host that steps a small block many times records the
call once (an OBGC_ASYNC handle, capture begun on the engine's stream) and replays the graph.  Every replay must give the
bits of the eager call, with the program's own eager work and synchronisations between the replays, and the recorded
graph must hold kernels, event records and waits only -- no memset node.  Both kernel forms a native-layout call can take:
the separate kernels of a small batch and the column-block kernel."""
import ctypes as C

import pytest

from conftest import has_gpu

pytestmark = pytest.mark.gpu

HIP_GRAPH_NODE_TYPE_KERNEL, HIP_GRAPH_NODE_TYPE_MEMSET = 0, 2
HIP_STREAM_CAPTURE_MODE_THREAD_LOCAL = 1


def _hip():
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
    hip.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    hip.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    hip.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipGraphGetNodes.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    hip.hipGraphNodeGetType.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    hip.hipGraphExecDestroy.argtypes = [C.c_void_p]
    hip.hipGraphDestroy.argtypes = [C.c_void_p]
    return hip


@pytest.mark.parametrize("form", ["separate", "column"])
def test_fused_step_replays_from_a_hip_graph(pkg, monkeypatch, form):
    if not has_gpu():
        pytest.fail("needs a GPU")
    import torch
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine
    monkeypatch.setenv("OBGC_COLUMN_MIN_CELLS", "700000" if form == "separate" else "0")
    eng = Engine(0, flags=L.OBGC_ASYNC)
    nz, nc = 60, 2048
    f64 = dict(dtype=torch.float64, device="cuda")
    tr = torch.empty((30, nz, nc), **f64)
    kc = {nm: torch.empty((nz, nc), **f64) for nm in ("temp", "salt", "zmid", "dz", "zbot", "fesedflux")}
    cc = {nm: torch.empty((nc,), **f64) for nm in ("lat", "dust_flux_in", "sw_flux")}
    kmax = torch.empty((nc,), dtype=torch.int32, device="cuda")
    own = torch.empty((5, nz, nc), **f64)
    torch.cuda.synchronize()
    eng.synth_fill(nz, nc, 4242, tracers=tr, kmax=kmax, **kc, **cc)
    eng.synth_fill_extra(nz, nc, 4242, dms=own[0], dmsp=own[1], prot=own[2], poly=own[3], lip=own[4])
    eng.sync()
    outs = dict(tend=torch.empty((30, nz, nc), **f64), own_tend=torch.empty((5, nz, nc), **f64),
                php=torch.empty((nz, nc), **f64), pha=torch.empty((nz, nc), **f64))
    diag = {}
    for nm in L.BGC_DIAG_KC:
        diag[nm] = torch.empty((nz, nc), **f64)
    for nm in L.BGC_DIAG_KCA:
        diag[nm] = torch.empty((4, nz, nc), **f64)
    for nm in L.BGC_DIAG_CA:
        diag[nm] = torch.empty((4, nc), **f64)
    for nm in L.BGC_DIAG_C:
        diag[nm] = torch.empty((nc,), **f64)
    ddiag = {nm: torch.empty((nz, nc), **f64) for nm in L.DMS_DIAG_KC}
    mdiag = {nm: torch.empty((nz, nc), **f64) for nm in L.MACROS_DIAG_KC}
    everything = list(outs.values()) + list(diag.values()) + list(ddiag.values()) + list(mdiag.values())
    ot = outs["own_tend"]
    prepared = eng.fused_step_prepare(nz, nc, L.OBGC_LAYOUT_COLFAST, tr, kc["temp"], kc["salt"], kc["zmid"], kc["dz"], kc["zbot"],
                                      cc["lat"], kmax, kc["fesedflux"], cc["dust_flux_in"], cc["sw_flux"], outs["tend"], outs["php"],
                                      outs["pha"], diag, True, dms=own[0], dmsp=own[1], SST=kc["temp"][0].contiguous(),
                                      dms_tend=ot[0], dmsp_tend=ot[1], dms_diag=ddiag, prot=own[2], poly=own[3], lip=own[4],
                                      prot_tend=ot[2], poly_tend=ot[3], lip_tend=ot[4], macros_diag=mdiag)

    def reset():
        for x in everything:
            x.fill_(-7.0)
        outs["php"].zero_(); outs["pha"].zero_()               # a cold carbonate solve every time
        torch.cuda.synchronize()

    # eager: the answer, and the call that reserves the scratch arena (nothing may allocate under capture)
    reset()
    eng.fused_step_run(prepared)
    eng.sync()
    assert eng.last_schedule() == (1 if form == "separate" else 4)
    want = [x.clone() for x in everything]
    assert torch.isfinite(outs["tend"]).all() and (outs["tend"] != -7.0).all()

    hip = _hip()
    stream = C.c_void_p(eng.lib.obgc_stream(eng.h))
    reset()
    assert hip.hipStreamBeginCapture(stream, HIP_STREAM_CAPTURE_MODE_THREAD_LOCAL) == 0
    try:
        eng.fused_step_run(prepared)
    finally:
        graph = C.c_void_p()
        rc = hip.hipStreamEndCapture(stream, C.byref(graph))
    assert rc == 0 and graph.value
    n = C.c_size_t(0)
    assert hip.hipGraphGetNodes(graph, None, C.byref(n)) == 0 and n.value > 0
    nodes = (C.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(graph, nodes, C.byref(n)) == 0
    types = []
    for nd in nodes:
        t = C.c_int(-1)
        assert hip.hipGraphNodeGetType(nd, C.byref(t)) == 0
        types.append(t.value)
    assert HIP_GRAPH_NODE_TYPE_MEMSET not in types
    assert types.count(HIP_GRAPH_NODE_TYPE_KERNEL) >= (5 if form == "separate" else 3)
    # nothing ran while recording
    torch.cuda.synchronize()
    assert (outs["tend"] == -7.0).all()
    gexec = C.c_void_p()
    assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
    for replay in range(3):
        reset()
        busy = (tr * 1.5).sum()                                # the program's own eager work between replays
        torch.cuda.synchronize()
        assert hip.hipGraphLaunch(gexec, stream) == 0
        eng.sync()
        for got, ref in zip(everything, want):
            assert torch.equal(got, ref), (form, replay)
        assert torch.isfinite(busy)
    assert hip.hipGraphExecDestroy(gexec) == 0 and hip.hipGraphDestroy(graph) == 0
    # the handle goes on working eagerly after the capture
    reset()
    eng.fused_step_run(prepared)
    eng.sync()
    for got, ref in zip(everything, want):
        assert torch.equal(got, ref)
