# A wider net than the test-suite casts: GPU vs oracle over many blocks of synthetic columns (cold call, then a warm call on
# perturbed tracers with the previous pH), 60, 80 and 100 levels, worst case per class.  The cold call is the fused
# BGC + DMS + MACROS step (the headline path), the warm call BGC_SourceSink alone.  ~2 minutes for 8 blocks, mostly oracle time.
# python tools/parity_soak.py [blocks [path [columns per block]]]   (on an MI355X; the oracle only checks here; path = device_colfast (default:
# blocks below 700 000 cells then run on the separate kernels, OBGC_COLUMN_KERNEL=2 forces the column-block kernel) or device_tiled)
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
nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
# (blocks of at most 3 000 columns stay under 300 000 cells: the native layout then runs the sweep as recurrence-only walk + cell-parallel kernels)
NCOLS = int(sys.argv[3]) if len(sys.argv) > 3 else 8000
E, O = ColumnEngine(0, path=sys.argv[2] if len(sys.argv) > 2 else "device_colfast"), orabind.Oracle()
worst, bad = {}, 0
for b in range(nblocks):
    nz = (60, 80, 100)[b % 3]
    nc, col0 = NCOLS, 1_000_003 * b + 17
    syn = synth.generate(nc, nz, col0=col0)
    tr = synth.bgc_tracer_array(syn)
    rest = (syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"], syn["dust"], syn["sw"])
    z = np.zeros((nc, nz))
    dtr, mtr, sst = synth.dms_tracer_array(syn), synth.macros_tracer_array(syn), syn["forcing"]["SST"]
    fused = E.Fused_SourceSink(tr, *rest, z, z, sst, dtr, mtr)
    g1, o1 = fused["bgc"], O.bgc_source_sink(tr, *rest, z, z)
    rep_dm = P.compare_simple(fused["dms"], O.dms_source_sink(dtr, syn["dz"], syn["kmax"], sst, syn["sw"]), "dms")
    P.compare_simple(fused["macros"], O.macros_source_sink(mtr, syn["kmax"]), "macros", rep_dm)
    bad += 0 if rep_dm.ok() else 1
    for k, v in rep_dm.summary().items():
        if v["max"] >= worst.get("dm_" + k, (-1.0, ""))[0]:
            worst["dm_" + k] = (v["max"], "%s block %d (fused, col0=%d, nz=%d)" % (v["at"], b, col0, nz))
    tr2 = tr * (1.0 + 1e-3 * np.cos(np.arange(tr.size).reshape(tr.shape)))
    # warm call: both sides start from the ORACLE's previous pH, so the brackets are identical
    g2 = E.BGC_SourceSink(tr2, *rest, o1["ph_prev"], o1["ph_prev_alt"], alt_co2_use_eco=False)
    o2 = O.bgc_source_sink(tr2, *rest, o1["ph_prev"], o1["ph_prev_alt"], alt_co2_use_eco=False)
    for tag, g, o in (("cold", g1, o1), ("warm", g2, o2)):
        rep = P.compare_bgc(g, o, dz=syn["dz"])
        bad += 0 if rep.ok() else 1
        for k, v in rep.summary().items():
            if v["max"] >= worst.get(k, (-1.0, ""))[0]:
                worst[k] = (v["max"], "%s block %d (%s, col0=%d, nz=%d)" % (v["at"], b, tag, col0, nz))
    print("block %d nz=%d done, failures so far %d" % (b, nz, bad), flush=True)
print("kernel rev", open(os.path.join(ROOT, "ocean-bgc_amd", "KERNEL_REV")).read().strip(), "blocks", nblocks, "FAILURES", bad)
for k, v in worst.items():
    print("  %-12s %.3e  %s" % (k, v[0], v[1]))
