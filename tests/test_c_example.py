"""examples/resident_step.c: a C host on the C ABI alone (no Python, no PyTorch in that process).  The same sequence
through the Python host must leave the same bits in the tracers."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, has_gpu

EXE = os.path.join(ROOT, "examples", "build", "resident_step")


@pytest.mark.gpu
def test_c_host_matches_python_host(tmp_path, pkg):
    if not has_gpu():
        pytest.fail("needs a GPU")
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    nc, nz, nsteps, dt = 3000, 60, 3, 900.0
    out = str(tmp_path / "tracers.bin")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    res = subprocess.run([EXE, str(nc), str(nz), out], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "resident_step columns=%d" % nc in res.stdout, res.stdout + res.stderr
    got = np.fromfile(out, dtype=np.float64).reshape(30, nz, nc)

    import torch
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine
    E = Engine(0)
    f64 = dict(dtype=torch.float64, device="cuda:0")
    tr = torch.empty((30, nz, nc), **f64)
    kc = {k: torch.empty((nz, nc), **f64) for k in ("temp", "salt", "zmid", "dz", "zbot", "fesedflux")}
    cc = {k: torch.empty((nc,), **f64) for k in ("lat", "dust_flux_in", "sw_flux")}
    kmax = torch.empty((nc,), dtype=torch.int32, device="cuda:0")
    E.synth_fill(nz, nc, 0, tracers=tr, kmax=kmax, **kc, **cc)
    const = dict(surfacePressure=1.0, iceFraction=0.1, windSpeedSquared10m=6.4e5, atmCO2=400.0, atmCO2_ALT_CO2=280.0,
                 surface_pH=0.0, surface_pH_alt_co2=0.0, surfaceDepth=5.0, SSS=35.0)
    forcing = {k: torch.full((nc,), v, **f64) for k, v in const.items()}
    forcing["SST"] = kc["temp"][0].clone()
    flux = [torch.zeros((30, nc), **f64) for _ in range(4)]
    net = torch.zeros((30, nc), **f64)
    tend = torch.zeros((30, nz, nc), **f64)
    ph = torch.zeros((2, nz, nc), **f64)
    torch.cuda.synchronize()
    CF = L.OBGC_LAYOUT_COLFAST
    for _ in range(nsteps):
        E.bgc_surface_fluxes(nz, nc, CF, tr, forcing, *flux, net, None)
        E.bgc_source_sink(nz, nc, CF, tr, kc["temp"], kc["salt"], kc["zmid"], kc["dz"], kc["zbot"], cc["lat"], kmax, kc["fesedflux"],
                          cc["dust_flux_in"], cc["sw_flux"], tend, ph[0], ph[1], None, True)
        E.apply_tendencies(nz, nc, CF, tr, tend, kmax, dt, net, kc["dz"], True)
    E.sync()
    assert np.array_equal(got, tr.cpu().numpy())
    assert np.isfinite(got).all()
