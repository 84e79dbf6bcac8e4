#!/bin/bash
# Measurement script (fp64 VALU utilisation per kernel); run from the repository root on an MI355X (e.g. through gpurun).

cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/valu
rm -rf $D; mkdir -p $D
for mode in full none; do
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $D/$mode -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --diag $mode > $D/$mode.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
out = {"kernel_rev": open('ocean-bgc_amd/KERNEL_REV').read().strip(), "unit": "fraction of the fp64 vector issue slots of 1024 SIMDs (SQ_ACTIVE_INST_VALU quad-cycles x 4 / (GRBM_GUI_ACTIVE/8 x 1024))", "runs": {}}
for mode in ("full", "none"):
    f = glob.glob('gpurun_out/valu/%s/*/*_counter_collection.csv' % mode)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        for key in ('bgc_cell_kernel', 'bgc_carbonate_kernel', 'bgc_sweep_kernel', 'bgc_att_kernel'):
            if key in n: acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
    run = {}
    for k, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        cyc = m['GRBM_GUI_ACTIVE'] / 8.0
        run[k] = {"valu_busy": m['SQ_ACTIVE_INST_VALU'] * 4.0 / (cyc * 1024.0), "valu_insts_per_wave": m['SQ_INSTS_VALU'] / m['SQ_WAVES'],
                  "kernel_cycles": cyc}
    out["runs"]["cfg2_" + mode] = run
json.dump(out, open('gpurun_out/valu/valu.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
