# Device-resident reference (level-fastest) layout at cfg2 size -- hybrid vs all-transposes vs strides.
# Run from the repository root on an MI355X:  python tools/reference_layout_bench.py
# device-resident REFERENCE layout (Fortran k-fastest) at cfg2 size: tiled-transpose path vs generic strides
import os, sys, time, torch
sys.path.insert(0,'.'); from __graft_entry__ import load_pkg; load_pkg()
from ocean_bgc_amd import _lib as L
from ocean_bgc_amd.engine import Engine
nz, nc = 60, 100000
def run(tag, env):
    for k,v in env.items(): os.environ[k]=v
    eng = Engine(0)
    f64 = dict(dtype=torch.float64, device='cuda')
    trn = torch.empty((30,nz,nc), **f64); kc = {nm: torch.empty((nz,nc), **f64) for nm in ("temp","salt","zmid","dz","zbot","fesedflux")}
    cc = {nm: torch.empty((nc,), **f64) for nm in ("lat","dust_flux_in","sw_flux")}; kmax = torch.empty((nc,), dtype=torch.int32, device='cuda')
    eng.synth_fill(nz, nc, 0, tracers=trn, kmax=kmax, **kc, **cc)
    tr = trn.transpose(1,2).contiguous(); kk = {k: v.t().contiguous() for k,v in kc.items()}
    tend = torch.empty((30,nc,nz), **f64); ph = torch.zeros((2,nc,nz), **f64)
    diag = {}
    for nm in L.BGC_DIAG_KC:
        if nm not in L.BGC_DIAG_KC_UNTOUCHED: diag[nm] = torch.empty((nc,nz), **f64)
    for nm in L.BGC_DIAG_KCA: diag[nm] = torch.empty((4,nc,nz), **f64)
    for nm in L.BGC_DIAG_CA: diag[nm] = torch.empty((4,nc), **f64)
    for nm in L.BGC_DIAG_C: diag[nm] = torch.empty((nc,), **f64)
    torch.cuda.synchronize()
    for dg, name in ((diag, "full"), (None, "none")):
        ts=[]
        for i in range(5):
            ph.zero_(); torch.cuda.synchronize(); t0=time.perf_counter()
            eng.bgc_source_sink(nz, nc, L.OBGC_LAYOUT_KFAST, tr, kk["temp"], kk["salt"], kk["zmid"], kk["dz"], kk["zbot"], cc["lat"], kmax, kk["fesedflux"], cc["dust_flux_in"], cc["sw_flux"], tend, ph[0], ph[1], dg, True)
            ts.append(time.perf_counter()-t0)
        print(tag, name, "ms/step %.2f" % (1e3*min(ts[1:])), flush=True)
    eng.close()
run("kfast hybrid         ", {"OBGC_KFAST_DIRECT":"0", "OBGC_KFAST_HYBRID":"1"})
if os.environ.get("OBGC_BENCH_HYBRID_ONLY", "0") != "1":     # set to 1 to time the shipped scheme alone
    run("kfast all-transposes ", {"OBGC_KFAST_DIRECT":"0", "OBGC_KFAST_HYBRID":"0"})
    run("kfast generic strides", {"OBGC_KFAST_DIRECT":"1"})
