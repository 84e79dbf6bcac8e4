# Worst-case differences GPU vs oracle, 20 000 columns x 60, all outputs of BGC_SourceSink.
# python tools/parity_summary.py   (on an MI355X; the oracle is test infrastructure and only checks here)
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg
load_pkg()
import parity as P
from ocean_bgc_amd import synth
from ocean_bgc_amd.modules import ColumnEngine
from oracle import orabind
nc, nz = 20000, 60
syn = synth.generate(nc, nz, col0=0)
args = (synth.bgc_tracer_array(syn), syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"],
        syn["fesed"], syn["dust"], syn["sw"], np.zeros((nc, nz)), np.zeros((nc, nz)))
rep = P.compare_bgc(ColumnEngine(0, path="device_colfast").BGC_SourceSink(*args), orabind.Oracle().bgc_source_sink(*args), dz=syn["dz"])
print("kernel rev", open(os.path.join(ROOT, "ocean-bgc_amd", "KERNEL_REV")).read().strip(), "ok" if rep.ok() else "FAIL")
print("parity summary", rep.summary() if hasattr(rep, "summary") else rep.worst)
