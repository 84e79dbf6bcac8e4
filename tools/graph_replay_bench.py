# A small block stepped many times: the fused step enqueued eagerly (OBGC_ASYNC handle) against replays of a HIP graph that
# recorded one call.  python tools/graph_replay_bench.py [ncol [nlev]]
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from __graft_entry__ import load_pkg
load_pkg()
import torch
from ocean_bgc_amd import _lib as L
from ocean_bgc_amd.engine import Engine
from test_gpu_graph_capture import _hip, HIP_STREAM_CAPTURE_MODE_THREAD_LOCAL

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 60
hip = _hip()
f64 = dict(dtype=torch.float64, device="cuda")
for diag_on in (True, False):
    for form, min_cells in (("separate kernels", "1000000000"), ("column-block kernel", "0")):
        os.environ["OBGC_COLUMN_MIN_CELLS"] = min_cells
        eng = Engine(0, flags=L.OBGC_ASYNC)
        tr = torch.empty((30, nz, nc), **f64)
        kc = {nm: torch.empty((nz, nc), **f64) for nm in ("temp", "salt", "zmid", "dz", "zbot", "fesedflux")}
        cc = {nm: torch.empty((nc,), **f64) for nm in ("lat", "dust_flux_in", "sw_flux")}
        kmax = torch.empty((nc,), dtype=torch.int32, device="cuda")
        own = torch.empty((5, nz, nc), **f64)
        torch.cuda.synchronize()
        eng.synth_fill(nz, nc, 4242, tracers=tr, kmax=kmax, **kc, **cc)
        eng.synth_fill_extra(nz, nc, 4242, dms=own[0], dmsp=own[1], prot=own[2], poly=own[3], lip=own[4])
        tend, ot = torch.zeros((30, nz, nc), **f64), torch.zeros((5, nz, nc), **f64)
        php, pha = torch.zeros((nz, nc), **f64), torch.zeros((nz, nc), **f64)
        diag = ddiag = mdiag = None
        if diag_on:
            diag = {nm: torch.zeros((nz, nc), **f64) for nm in L.BGC_DIAG_KC}
            diag.update({nm: torch.zeros((4, nz, nc), **f64) for nm in L.BGC_DIAG_KCA})
            diag.update({nm: torch.zeros((4, nc), **f64) for nm in L.BGC_DIAG_CA})
            diag.update({nm: torch.zeros((nc,), **f64) for nm in L.BGC_DIAG_C})
            ddiag = {nm: torch.zeros((nz, nc), **f64) for nm in L.DMS_DIAG_KC}
            mdiag = {nm: torch.zeros((nz, nc), **f64) for nm in L.MACROS_DIAG_KC}
        torch.cuda.synchronize()
        p = eng.fused_step_prepare(nz, nc, L.OBGC_LAYOUT_COLFAST, tr, kc["temp"], kc["salt"], kc["zmid"], kc["dz"], kc["zbot"], cc["lat"], kmax,
                                   kc["fesedflux"], cc["dust_flux_in"], cc["sw_flux"], tend, php, pha, diag, True, dms=own[0], dmsp=own[1],
                                   SST=kc["temp"][0].contiguous(), dms_tend=ot[0], dmsp_tend=ot[1], dms_diag=ddiag, prot=own[2], poly=own[3],
                                   lip=own[4], prot_tend=ot[2], poly_tend=ot[3], lip_tend=ot[4], macros_diag=mdiag)
        for _ in range(5):
            eng.fused_step_run(p)
        eng.sync()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            eng.fused_step_run(p)
        eng.sync()
        eager = (time.perf_counter() - t0) / n
        stream = C.c_void_p(eng.lib.obgc_stream(eng.h))
        assert hip.hipStreamBeginCapture(stream, HIP_STREAM_CAPTURE_MODE_THREAD_LOCAL) == 0
        eng.fused_step_run(p)
        graph, gexec = C.c_void_p(), C.c_void_p()
        assert hip.hipStreamEndCapture(stream, C.byref(graph)) == 0
        assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
        for _ in range(5):
            assert hip.hipGraphLaunch(gexec, stream) == 0
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            hip.hipGraphLaunch(gexec, stream)
        eng.sync()
        replay = (time.perf_counter() - t0) / n
        print("%d columns x %d, %s, %s: eager %.3f ms per step, graph replay %.3f ms"
              % (nc, nz, "every diagnostic" if diag_on else "tendencies only", form, 1e3 * eager, 1e3 * replay), flush=True)
        hip.hipGraphExecDestroy(gexec); hip.hipGraphDestroy(graph)
        eng.close()
