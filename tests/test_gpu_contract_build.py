"""The reference-order build of the library (make -C ocean-bgc_amd/csrc contract: -DOBGC_TALK_REFORDER
-DOBGC_PRESSURE_FACTOR_SEPARATE, i.e. talk_row term by term as co2calc.F90:1001-1092 and K(T,S) * exp(lnKfac) as
co2calc.F90:480-490) on the GPU against the oracle.  The shipped library re-associates both for speed; this keeps the
contract form compiling and inside the same bars.  Runs in a child process: one process binds one libobgc."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, has_gpu

pytestmark = pytest.mark.gpu

CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.environ["OBGC_ROOT"]); sys.path.insert(0, os.path.join(os.environ["OBGC_ROOT"], "tests"))
from conftest import load_pkg
load_pkg()
import parity as P
from ocean_bgc_amd import _lib, synth
assert _lib.LIB_PATH.endswith("libobgc_reforder.so"), _lib.LIB_PATH
from ocean_bgc_amd.modules import ColumnEngine
from oracle import orabind
nc, nz = 3000, 80
syn = synth.generate(nc, nz, col0=1234)
tr = synth.bgc_tracer_array(syn)
args = (tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"], syn["dust"], syn["sw"])
z = np.zeros((nc, nz))
E, O = ColumnEngine(0, path="device_colfast"), orabind.Oracle()
cold_e, cold_o = E.BGC_SourceSink(*args, z, z), O.bgc_source_sink(*args, z, z)
rep = P.compare_bgc(cold_e, cold_o)
tr2 = tr * 1.001
args2 = (tr2,) + args[1:]
P.compare_bgc(E.BGC_SourceSink(*args2, cold_o["ph_prev"], cold_o["ph_prev_alt"]), O.bgc_source_sink(*args2, cold_o["ph_prev"], cold_o["ph_prev_alt"]), rep)
f = syn["forcing"]
flux = [syn[nm] for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")]
P.compare_surface(E.BGC_SurfaceFluxes(tr, f, *flux), O.bgc_surface_fluxes(tr, f, *flux), rep)
print("RESULT " + json.dumps({"ok": rep.ok(), "fail": rep.fail[:8], "summary": rep.summary()}))
"""


def test_reference_order_build_meets_the_bars():
    if not has_gpu():
        pytest.fail("needs a GPU")
    lib = os.path.join(ROOT, "ocean-bgc_amd", "libobgc_reforder.so")
    assert os.path.exists(lib), "build it first: make -C ocean-bgc_amd/csrc contract (__graft_entry__.build() does)"
    env = dict(os.environ, OBGC_LIB=lib, OBGC_ROOT=ROOT)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert res["ok"], res["fail"]
    print("reference-order build:", res["summary"])
