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


@pytest.mark.gpu
def test_c_model_on_the_resident_block_matches_python_host(tmp_path, pkg, synth):
    """examples/mesh_block_device.c: device arrays in the model's layout through obgc_block_step_device, a diagnostic set on
    the block, the state for a restart -- against the Python host stepping the same block from HOST arrays."""
    import ctypes as C
    if not has_gpu():
        pytest.fail("needs a GPU")
    exe = os.path.join(ROOT, "examples", "build", "mesh_block_device")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    nc, nz, nsteps = 1500, 60, 3
    out = str(tmp_path / "block.bin")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    res = subprocess.run([exe, str(nc), str(nz), out], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "mesh_block_device cells=%d" % nc in res.stdout, res.stdout + res.stderr
    raw = np.fromfile(out, dtype=np.float64)
    kc = nc * nz
    sizes = [("tend", (nc, nz, 30)), ("sflux", (nc, 30)), ("co3", (nc, nz)), ("pcz", (nc,)), ("php", (nc, nz)), ("sph", (nc,))]
    got, pos = {}, 0
    for nm, shape in sizes:
        n = int(np.prod(shape))
        got[nm] = raw[pos:pos + n].reshape(shape)
        pos += n
    assert pos == raw.size

    from test_mpas_adapter import block_args, mpas_block
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine
    m, _ = mpas_block(synth, nc, nz, col0=0)
    E = Engine(0)
    blk = C.c_void_p()
    assert E.lib.obgc_block_create(E.h, nc, nz, C.byref(blk)) == 0
    ds = E.diag_set(nz, nc, ["CO3", "pH_3D", "photoC_TOT_zint"])
    assert E.lib.obgc_block_set_diagnostics(blk, C.cast(ds.bgc, C.c_void_p), None, None) == 0
    tend = np.zeros((nc, nz, 30))
    for s in range(nsteps):
        a, keep, sflux = block_args(L, m, nc, nz, tend, False)
        assert E.lib.obgc_block_step(blk, C.byref(a)) == 0
        ds.accumulate(1.0 / nsteps, reset=(s == 0))
    st = [np.empty((nc, nz)), np.empty((nc, nz)), np.empty(nc), np.empty(nc)]
    assert E.lib.obgc_block_get_state(blk, *[x.ctypes.data_as(C.c_void_p) for x in st]) == 0
    assert np.array_equal(got["tend"], tend) and np.array_equal(got["sflux"], sflux)
    assert np.array_equal(got["co3"], ds.download("CO3")) and np.array_equal(got["pcz"], ds.download("photoC_TOT_zint"))
    assert np.array_equal(got["php"], st[0]) and np.array_equal(got["sph"], st[2])
    assert np.isfinite(got["tend"]).all() and np.abs(got["tend"]).max() > 0 and (got["co3"] > 0).any()
    ds.close()
    assert E.lib.obgc_block_destroy(blk) == 0
