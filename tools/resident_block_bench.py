# One step of a device-resident mesh block (obgc_block_step) at cfg2 size, PCIe-inclusive:
# only the model's arrays cross the link.  python tools/resident_block_bench.py
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from __graft_entry__ import load_pkg
load_pkg()
from ocean_bgc_amd import _lib as L, synth
from ocean_bgc_amd.engine import Engine
from test_mpas_adapter import mpas_block, block_args
nc, nz = 100000, 60
m, pk = mpas_block(synth, nc, nz, col0=0)
E = Engine(0)
blk = C.c_void_p()
assert E.lib.obgc_block_create(E.h, nc, nz, C.byref(blk)) == 0
tend = np.zeros((nc, nz, 30))
cells = int(m["kmax"].clip(0, nz).sum())
for acc in (False, True):
    ts = []
    a, keep, sflux = block_args(L, m, nc, nz, tend, acc)      # the model's arrays as they are: nothing is converted per step
    for i in range(5):
        t0 = time.perf_counter()
        assert E.lib.obgc_block_step(blk, C.byref(a)) == 0
        ts.append(time.perf_counter() - t0)
    print("resident block step, accumulate=%s: best %.1f ms -> %.3e updates/s (PCIe-inclusive)" % (acc, 1e3 * min(ts[1:]), 30 * cells / min(ts[1:])), flush=True)

# the same block with DMS and MACROS in the step (own tracers 2 + 3 planes up, their tendencies down, one fused step)
syn = synth.generate(nc, nz, col0=0)
dtr, mtr = synth.dms_tracer_array(syn), synth.macros_tracer_array(syn)
own_d = np.ascontiguousarray(dtr[0:2].transpose(1, 2, 0))
own_m = np.ascontiguousarray(mtr[0:3].transpose(1, 2, 0))
dtend, mtend, dsf = np.zeros((nc, nz, 2)), np.zeros((nc, nz, 3)), np.zeros((nc, 2))
a, keep, sflux = block_args(L, m, nc, nz, tend, False)
a.dmsTracers = own_d.ctypes.data_as(C.c_void_p); a.dmsTracersTend = dtend.ctypes.data_as(C.c_void_p)
a.dmsSurfaceFlux = dsf.ctypes.data_as(C.c_void_p)
a.macrosTracers = own_m.ctypes.data_as(C.c_void_p); a.macrosTracersTend = mtend.ctypes.data_as(C.c_void_p)
ts = []
for i in range(5):
    t0 = time.perf_counter()
    assert E.lib.obgc_block_step(blk, C.byref(a)) == 0
    ts.append(time.perf_counter() - t0)
print("resident block step with DMS + MACROS (fused): best %.1f ms -> %.3e updates/s (PCIe-inclusive)"
      % (1e3 * min(ts[1:]), 35 * cells / min(ts[1:])), flush=True)

# the same steps for a model whose arrays already live on the GPU (obgc_block_step_device): no link in the step at all
import torch
dev = torch.device("cuda:0")
host = {"ecosysTracersTend": tend, "ecosysSurfaceFlux": sflux, "dmsTracers": own_d, "dmsTracersTend": dtend, "dmsSurfaceFlux": dsf,
        "macrosTracers": own_m, "macrosTracersTend": mtend}
by_addr = {x.ctypes.data: x for x in keep}
own_names = ("dmsTracers", "dmsTracersTend", "dmsSurfaceFlux", "macrosTracers", "macrosTracersTend")
for with_own in (False, True):
    b2 = C.c_void_p()
    assert E.lib.obgc_block_create(E.h, nc, nz, C.byref(b2)) == 0
    t = {}
    d = L.BlockStepArgs()
    for n, _t in L.BlockStepArgs._fields_:
        if n == "accumulate" or (n in own_names and not with_own):
            continue
        src = host[n] if n in host else by_addr[getattr(a, n)]
        t[n] = torch.from_numpy(src).to(dev)
        setattr(d, n, C.c_void_p(t[n].data_ptr()))
    for acc in (False, True):
        d.accumulate = 1 if acc else 0
        ts = []
        for i in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            assert E.lib.obgc_block_step_device(b2, C.byref(d)) == 0
            ts.append(time.perf_counter() - t0)
        print("device-pointer block step%s, accumulate=%s: best %.2f ms -> %.3e updates/s"
              % (" with DMS + MACROS (fused)" if with_own else "", acc, 1e3 * min(ts[1:]), (35 if with_own else 30) * cells / min(ts[1:])),
              flush=True)
    assert E.lib.obgc_block_destroy(b2) == 0
