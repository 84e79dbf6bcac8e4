# PCIe-inclusive host-memory path at cfg2 size.  python tools/host_path_bench.py
# PCIe-inclusive rate: host (numpy) arrays in the reference layout through the C ABI, cfg2 size
import sys, time, numpy as np
sys.path.insert(0,'.'); from __graft_entry__ import load_pkg; load_pkg()
from ocean_bgc_amd import _lib as L, synth
from ocean_bgc_amd.engine import Engine
nz, nc = 60, 100000
eng = Engine(0)
syn = synth.generate(nc, nz)
tr = synth.bgc_tracer_array(syn)
tend = np.zeros_like(tr); ph = np.zeros((2, nc, nz))
diag = {}
for nm in L.BGC_DIAG_KC:
    if nm not in L.BGC_DIAG_KC_UNTOUCHED: diag[nm] = np.zeros((nc, nz))
for nm in L.BGC_DIAG_KCA: diag[nm] = np.zeros((4, nc, nz))
for nm in L.BGC_DIAG_CA: diag[nm] = np.zeros((4, nc))
for nm in L.BGC_DIAG_C: diag[nm] = np.zeros((nc,))
cells = int(syn["kmax"].sum())
# a host that reads a handful of members (what obgc_bgc_select_diagnostics of the Fortran drop-in passes on): six planes and
# the column integrals instead of 133 planes
few = {nm: diag[nm] for nm in ("CO3", "pH_3D", "O2_CONSUMPTION", "POC_FLUX_IN", "photoC_TOT", "NITRIF", "photoC_TOT_zint",
                               "Jint_Ctot", "zsatcalc", "O2_ZMIN")}
for dg, name in ((diag, "full"), (few, "ten selected members"), (None, "none")):
    ts = []
    for i in range(4):
        ph[:] = 0; t0 = time.perf_counter()
        eng.bgc_source_sink(nz, nc, L.OBGC_LAYOUT_KFAST, tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"], syn["dust"], syn["sw"], tend, ph[0], ph[1], dg, True)
        ts.append(time.perf_counter() - t0)
    print("host path", name, "best %.1f ms -> %.3e updates/s (PCIe-inclusive, pageable host memory)" % (1e3 * min(ts), 30 * cells / min(ts)), flush=True)

# BGC_SurfaceFluxes through host memory: only the surface rows of six tracers travel, not the (k,column,n) array
f = {k: v.copy() for k, v in syn["forcing"].items()}
fl = [syn[nm].copy() for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")]
net = np.zeros((30, nc))
fd = {nm: np.zeros(nc) for nm in L.BGC_FLUX_DIAG}
ts = []
for i in range(4):
    t0 = time.perf_counter()
    eng.bgc_surface_fluxes(nz, nc, L.OBGC_LAYOUT_KFAST, tr, f, *fl, net, fd)
    ts.append(time.perf_counter() - t0)
print("host path BGC_SurfaceFluxes best %.1f ms (%d columns)" % (1e3 * min(ts), nc), flush=True)
