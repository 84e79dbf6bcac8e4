# Bandwidth of obgc_apply_tendencies and obgc_bgc_diag_accumulate.  python tools/resident_kernels_bench.py [ncol]
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
load_pkg()
import numpy as np, torch
from ocean_bgc_amd.engine import Engine
from ocean_bgc_amd import _lib as L
nc, nz = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 60
E = Engine(0)
dev = torch.device("cuda:0")
x = torch.rand((30, nz, nc), dtype=torch.float64, device=dev)
t = torch.rand((30, nz, nc), dtype=torch.float64, device=dev) * 1e-6
dz = torch.full((nz, nc), 1000.0, dtype=torch.float64, device=dev)
nf = torch.rand((30, nc), dtype=torch.float64, device=dev)
kmax = torch.tensor((nz - (np.arange(nc) % 7)).astype(np.int32), device=dev)
active = int((nz - (np.arange(nc) % 7)).sum())
torch.cuda.synchronize()
ms = []
for i in range(12):
    E.apply_tendencies(nz, nc, L.OBGC_LAYOUT_COLFAST, x, t, kmax, 900.0, nf, dz, True)
    E.sync()
    ms.append(E.lib.obgc_last_kernel_ms(E.h, 0))
ms = sorted(ms[2:])
m = ms[len(ms) // 2]
byt = active * 30 * 24
print("apply_tendencies %d x %d: %.3f ms, %.1f GB algorithmic -> %.0f GB/s (%.1f%% of 8 TB/s)" % (nc, nz, m, byt / 1e9, byt / m / 1e6, byt / m / 1e6 / 80))
# ---- running means of all (k,c) + (k,c,a) diagnostics
names_kc, names_kca = list(L.BGC_DIAG_KC), list(L.BGC_DIAG_KCA)
inst = {nm: torch.rand((nz, nc), dtype=torch.float64, device=dev) for nm in names_kc}
inst.update({nm: torch.rand((4, nz, nc), dtype=torch.float64, device=dev) for nm in names_kca})
mean = {nm: torch.zeros_like(v) for nm, v in inst.items()}
torch.cuda.synchronize()
ms = []
for i in range(8):
    E.bgc_diag_accumulate(nz, nc, L.OBGC_LAYOUT_COLFAST, inst, mean, 0.1, i == 0)
    E.sync()
    ms.append(E.lib.obgc_last_kernel_ms(E.h, 0))
m = sorted(ms[2:])[len(ms[2:]) // 2]
nplane = len(names_kc) + 4 * len(names_kca)
byt = nplane * nz * nc * 24
print("diag_accumulate %d planes %d x %d: %.3f ms, %.1f GB -> %.0f GB/s (%.1f%% of 8 TB/s)" % (nplane, nc, nz, m, byt / 1e9, byt / m / 1e6, byt / m / 1e6 / 80))
