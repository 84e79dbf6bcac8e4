#!/bin/bash
# rocprofv3 evidence for the current kernel revision (ocean-bgc_amd/KERNEL_REV); run from the repository root on an
# MI355X (e.g. through gpurun).  Everything lands in gpurun_out/prof_<rev>/; copy the summaries into profiles/.
#   1. kernel stats (--kernel-trace --stats) of the bench command for the headline (cfg4, fused step, every diagnostic),
#      cfg4 without diagnostics, cfg3 and cfg2 with / without diagnostics
#   2. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (pmc_summary.py applies the guide's corrections)
#   3. fp64 VALU utilisation of the kernels (one --pmc pass)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
REV=$(cat ocean-bgc_amd/KERNEL_REV)
D=gpurun_out/prof_$REV
rm -rf $D; mkdir -p $D
B="--no-cpu-baseline --no-extras"
run() { # tag, rocprof args..., -- bench args
  local t=$1; shift
  timeout -s KILL 300 rocprofv3 "$@" > $D/$t.log 2>&1 || { echo "FAILED $t"; tail -5 $D/$t.log; exit 1; }
}
# (bench.py's default layout is the column-block one; *_colfast = the same workload in the plane-per-field native layout)
declare -A CFG=( [cfg4_full]="--config cfg4 --diag full" [cfg4_none]="--config cfg4 --diag none" [cfg3_full]="--config cfg3 --diag full" \
                 [cfg2_full]="--config cfg2 --diag full" [cfg2_none]="--config cfg2 --diag none" [cfg5_full]="--config cfg5 --diag full" \
                 [cfg4_full_colfast]="--config cfg4 --diag full --layout colfast" [cfg2_full_colfast]="--config cfg2 --diag full --layout colfast" )
for tag in cfg4_full cfg4_none cfg3_full cfg2_full cfg2_none cfg5_full cfg4_full_colfast cfg2_full_colfast; do
  run stats_$tag --kernel-trace --stats --output-format csv -d $D/stats_$tag -- python3 bench.py $B ${CFG[$tag]} --steps 8 --warmup 2
  cp $D/stats_$tag/*/*_kernel_stats.csv $D/${REV}_kernel_stats_$tag.csv
  grep '^{' $D/stats_$tag.log | tail -1 > $D/${REV}_bench_under_rocprof_$tag.json
done
SPEC=""
for tag in cfg4_full cfg4_none cfg2_full cfg2_none cfg4_full_colfast; do
  run fetch_$tag --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch_$tag -- python3 bench.py $B ${CFG[$tag]} --steps 3 --warmup 1
  run write_$tag --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write_$tag -- python3 bench.py $B ${CFG[$tag]} --steps 3 --warmup 1
  SPEC="$SPEC $tag=$D/fetch_$tag,$D/write_$tag"
done
python3 tools/pmc_summary.py $D/pmc_traffic.json $SPEC > $D/pmc_summary.log 2>&1 || { echo "pmc_summary failed"; tail -5 $D/pmc_summary.log; exit 1; }
for tag in cfg4_full cfg4_none cfg2_full cfg2_none cfg4_full_colfast; do
  run valu_$tag --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $D/valu_$tag -- python3 bench.py $B ${CFG[$tag]} --steps 3 --warmup 1
done
python3 - "$D" <<'PY'
import csv, glob, collections, json, re, sys
D = sys.argv[1]
out = {"kernel_rev": open('ocean-bgc_amd/KERNEL_REV').read().strip(),
       "unit": "valu_busy: fraction of the vector issue slots of 1024 SIMDs (SQ_ACTIVE_INST_VALU quad-cycles x 4 / (GRBM_GUI_ACTIVE / 8 x 1024)); "
               "wait_inst_any: share of the wavefront cycles spent waiting for any instruction to issue", "runs": {}}
for tag in ("cfg4_full", "cfg4_none", "cfg2_full", "cfg2_none", "cfg4_full_colfast"):
    f = glob.glob('%s/valu_%s/*/*_counter_collection.csv' % (D, tag))[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)\b", r['Kernel_Name'].replace("(anonymous namespace)::", ""))
        if m and m.group(1).startswith(("bgc_", "fused_", "dms_", "macros_")) and m.group(1) != "bgc_recip_kernel":
            acc[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
    run = {}
    for k, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        cyc = m['GRBM_GUI_ACTIVE'] / 8.0
        run[k] = {"valu_busy": m['SQ_ACTIVE_INST_VALU'] * 4.0 / (cyc * 1024.0), "valu_insts_per_wave": m['SQ_INSTS_VALU'] / m['SQ_WAVES'],
                  "wait_inst_any": m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'], "kernel_cycles": cyc}
    out["runs"][tag] = run
json.dump(out, open(D + '/valu.json', 'w'), indent=1, sort_keys=True)
PY
echo "== headline kernel stats"; cut -d, -f1-4 $D/${REV}_kernel_stats_cfg4_full.csv | head -8 | cut -c1-170
echo "== traffic (cfg4_full)"; python3 -c "
import json; t=json.load(open('$D/pmc_traffic.json'))
for k,v in sorted(t['runs']['cfg4_full'].items(), key=lambda kv:-kv[1]['total'])[:6]: print('  %-24s read %.2f GB write %.2f GB' % (k, v['read']/1e9, v['write']/1e9))"
