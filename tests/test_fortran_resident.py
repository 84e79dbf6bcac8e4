"""examples/resident_step.F90: a Fortran host whose tracers live on the GPU (device pointers through the bind(C) layer,
module obgc_device).  SurfaceFluxes -> SourceSink -> explicit update over several steps, no array crossing PCIe between
set-up and read-back:
  * BGC alone: the same bits as the C host (examples/resident_step.c) and the oracle's multi-step sequence inside the bars;
  * BGC + DMS + MACROS through obgc_fused_step: the same bits as the Python host driving the same entry points;
  * wall time per step at 100 000 columns x 60 levels (printed; the shim on host arrays needs 159 ms per call there).
The executables are built beforehand (make -C examples, run by __graft_entry__.build())."""
import os
import re
import subprocess

import numpy as np
import pytest

import parity as P
from conftest import ROOT, has_gpu

pytestmark = pytest.mark.gpu

EXE_F = os.path.join(ROOT, "examples", "build", "resident_step_f")
EXE_C = os.path.join(ROOT, "examples", "build", "resident_step")


def _run(exe, *args):
    if not has_gpu():
        pytest.fail("needs a GPU")
    if not os.path.exists(exe):
        pytest.fail("%s is not built (make -C examples)" % exe)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:/opt/rocm/lib/llvm/lib:" + env.get("LD_LIBRARY_PATH", "")
    res = subprocess.run([exe] + [str(a) for a in args], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    return res.stdout


UNIFORM = dict(surfacePressure=1.0, iceFraction=0.1, windSpeedSquared10m=6.4e5, atmCO2=400.0, atmCO2_ALT_CO2=280.0,
               surface_pH=0.0, surface_pH_alt_co2=0.0, surfaceDepth=5.0, SSS=35.0)


def test_fortran_host_equals_c_host_and_oracle_sequence(tmp_path, pkg, oracle, synth):
    nc, nz, nsteps, dt = 3000, 60, 3, 900.0
    out_f, out_c = str(tmp_path / "f.bin"), str(tmp_path / "c.bin")
    txt = _run(EXE_F, nc, nz, out_f, 0, nsteps)
    assert "resident_step_f columns=%d" % nc in txt
    _run(EXE_C, nc, nz, out_c)
    got = np.fromfile(out_f, dtype=np.float64).reshape(30, nz, nc)
    assert np.array_equal(got, np.fromfile(out_c, dtype=np.float64).reshape(30, nz, nc))
    assert np.isfinite(got).all()
    # the oracle's own three-step sequence on a sample of the columns (states drift apart by ulps: relative 1e-12)
    sel = np.arange(0, nc, 12)
    syn = synth.generate(nc, nz, col0=0)
    x = synth.bgc_tracer_array(syn)[:, sel, :]
    cut = lambda a: a[sel]
    forcing = {k: np.full(sel.size, v) for k, v in UNIFORM.items()}
    forcing["SST"] = syn["temp"][sel, 0].copy()
    flux = [np.zeros((30, sel.size)) for _ in range(4)]
    ph = [np.zeros((sel.size, nz)), np.zeros((sel.size, nz))]
    a = tuple(cut(syn[k]) for k in ("temp", "salt", "zmid", "dz", "zbot", "lat", "kmax", "fesed", "dust", "sw"))
    for _ in range(nsteps):
        so = oracle.bgc_surface_fluxes(x, forcing, *flux)
        wo = oracle.bgc_source_sink(x, *a, *ph, diagnostics=False)
        x = oracle.apply_tendencies(x, wo["tend"], a[6], dt, so["netFlux"], a[3], True)
        ph = [wo["ph_prev"], wo["ph_prev_alt"]]
        for nm in ("surface_pH", "surface_pH_alt_co2"):
            forcing[nm] = so["forcing"][nm]
    assert P.e_norm(got[:, :, sel].transpose(0, 2, 1), x) < 1e-12


def test_fortran_fused_step_equals_python_host(tmp_path, pkg):
    nc, nz, nsteps, dt = 2000, 80, 3, 900.0
    out_f = str(tmp_path / "f.bin")
    txt = _run(EXE_F, nc, nz, out_f, 1, nsteps)
    assert "fused=1" in txt and re.search(r"schedule of the last source-sink call: [34]", txt), txt
    got = np.fromfile(out_f, dtype=np.float64).reshape(35, nz, nc)

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
    forcing = {k: torch.empty((nc,), **f64) for k in ("surfacePressure", "iceFraction", "windSpeedSquared10m", "atmCO2", "atmCO2_ALT_CO2",
                                                       "surfaceDepth", "SST", "SSS", "surface_pH", "surface_pH_alt_co2")}
    flux = {k: torch.empty((30, nc), **f64) for k in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")}
    own = torch.empty((5, nz, nc), **f64)
    E.synth_fill_extra(nz, nc, 0, dms=own[0], dmsp=own[1], prot=own[2], poly=own[3], lip=own[4], **forcing, **flux)
    net = torch.zeros((30, nc), **f64)
    tend, otend = torch.zeros((30, nz, nc), **f64), torch.zeros((5, nz, nc), **f64)
    ph = torch.zeros((2, nz, nc), **f64)
    torch.cuda.synchronize()
    CF = L.OBGC_LAYOUT_COLFAST
    for _ in range(nsteps):
        E.bgc_surface_fluxes(nz, nc, CF, tr, forcing, *[flux[k] for k in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")], net, None)
        E.fused_step(nz, nc, CF, tr, kc["temp"], kc["salt"], kc["zmid"], kc["dz"], kc["zbot"], cc["lat"], kmax, kc["fesedflux"],
                     cc["dust_flux_in"], cc["sw_flux"], tend, ph[0], ph[1], None, True, dms=own[0], dmsp=own[1], SST=forcing["SST"],
                     dms_tend=otend[0], dmsp_tend=otend[1], prot=own[2], poly=own[3], lip=own[4], prot_tend=otend[2],
                     poly_tend=otend[3], lip_tend=otend[4])
        E.apply_tendencies(nz, nc, CF, tr, tend, kmax, dt, net, kc["dz"], True)
        E.apply_tendencies(nz, nc, CF, own, otend, kmax, dt, None, kc["dz"], True)
    E.sync()
    assert np.array_equal(got[:30], tr.cpu().numpy())
    assert np.array_equal(got[30:], own.cpu().numpy())
    assert np.isfinite(got).all() and (got[30:] >= 0).all()


def test_fortran_resident_step_wall_time():
    """100 000 columns x 60 levels, tracers resident: a step (SurfaceFluxes + SourceSink + update, one host sync) must stay
    within a few kernel times -- two orders of magnitude below the 159 ms of one shim call on host arrays."""
    for fused in (0, 1):
        txt = _run(EXE_F, 100_000, 60, "-", fused, 6)
        m = re.search(r"first\s+([0-9.]+) ms, later steps\s+([0-9.]+) ms", txt)
        assert m, txt
        print("\nFortran resident loop, 100000 x 60, fused=%d: first step %s ms, later steps %s ms" % (fused, m.group(1), m.group(2)))
        assert float(m.group(2)) < 15.0, txt


@pytest.mark.parametrize("fused,nc,nz", [(0, 3_000, 60), (1, 2_029, 80)])
def test_fortran_host_in_the_column_block_layout(tmp_path, pkg, fused, nc, nz):
    """The same Fortran host with its level-dependent arrays as rows of one arena (OBGC_LAYOUT_TILED, module obgc_device with
    tiled = .true.): three steps give the bits of the native layout, for BGC alone and for the fused step, also when the
    column count is not a multiple of the 64-column block."""
    out_n, out_t = str(tmp_path / "native.bin"), str(tmp_path / "tiled.bin")
    _run(EXE_F, nc, nz, out_n, fused, 3, 0)
    txt = _run(EXE_F, nc, nz, out_t, fused, 3, 1)
    assert "tiled=1" in txt and re.search(r"schedule of the last source-sink call: [34]", txt), txt
    a, b = np.fromfile(out_n, dtype=np.float64), np.fromfile(out_t, dtype=np.float64)
    assert a.size == (35 if fused else 30) * nz * nc and np.isfinite(a).all()
    assert np.array_equal(a, b)
