#!/bin/bash
# Scalar-memory counters of the kernels of one bench.py run:  tools/pmc_smem.sh <tag> [bench.py args ...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/pmcs_$tag
rm -rf $D; mkdir -p $D
P="SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAVE_CYCLES"
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $D/p1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > $D/p1.log 2>&1 || { echo "pass failed"; tail -5 $D/p1.log; exit 1; }
python3 - "$D" <<'PY'
import csv, glob, collections, sys
D = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(D + '/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('obgc::', '').split('(')[0]
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    if m.get('SQ_WAVE_CYCLES', 0) < 1e8: continue
    print(k[:80]); print("   ", {c: "%.4g" % v for c, v in sorted(m.items())})
    if m.get('SQ_INSTS_SMEM'):
        print("    smem: mean cycles in flight per instr %.0f ; level*4/wave_cycles %.3f ; lds mean %.0f" % (
            4 * m['SQ_INST_LEVEL_SMEM'] / m['SQ_INSTS_SMEM'], m['SQ_INST_LEVEL_SMEM'] / m['SQ_WAVE_CYCLES'],
            4 * m['SQ_INST_LEVEL_LDS'] / max(1, m['SQ_INSTS_LDS'])))
PY
