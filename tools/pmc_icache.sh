#!/bin/bash
# Instruction-cache counters of the kernels of one bench.py run:  tools/pmc_icache.sh <tag> [bench.py args ...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/pmci_$tag
rm -rf $D; mkdir -p $D
P="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_STALL SQ_IFETCH_LEVEL SQ_IFETCH"
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
    if m.get('SQC_ICACHE_REQ', 0) < 1e6: continue
    print(k[:80]); print("   ", {c: "%.3g" % v for c, v in sorted(m.items())})
PY
