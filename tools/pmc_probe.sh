#!/bin/bash
# Wavefront-level counters of the kernels of one bench.py run (separate passes, <= 8 SQ counters each):
#   tools/pmc_probe.sh <tag> [bench.py args ...]        (run on the GPU box from the repository root; env passes through)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/pmc_$tag
rm -rf $D; mkdir -p $D
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"
P3="SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_IFETCH"
P4="GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $D/p$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > $D/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $D/p$i.log; exit 1; }
done
python3 - "$D" <<'PY'
import csv, glob, collections, sys, json
D = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(D + '/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('obgc::', '')
        n = n.split('(')[0]
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    if m.get('SQ_WAVES', 0) < 64: continue
    out[k] = m
json.dump(out, open(D + '/summary.json', 'w'), indent=1, sort_keys=True)
for k, m in sorted(out.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0))[:8]:
    w = m['SQ_WAVES']
    cyc = m.get('GRBM_GUI_ACTIVE', 0) / 8.0      # the counter is summed over 8 XCDs
    print("\n%s\n  waves %d  kernel cycles %.0f (%.3f ms @2.4GHz)  wave_cycles/wave %.0f" % (k[:110], w, cyc, cyc / 2.4e6, m['SQ_WAVE_CYCLES'] / w))
    wc = m['SQ_WAVE_CYCLES']
    print("  of wave-cycles: wait_any %.2f  wait_inst_any %.2f  wait_inst_lds %.3f  | active_inst_any %.2f valu %.2f lds %.3f vmem %.3f sca %.3f" % (
        m['SQ_WAIT_ANY'] / wc, m['SQ_WAIT_INST_ANY'] / wc, m.get('SQ_WAIT_INST_LDS', 0) / wc, m['SQ_ACTIVE_INST_ANY'] * 4 / wc, m['SQ_ACTIVE_INST_VALU'] * 4 / wc,
        m.get('SQ_ACTIVE_INST_LDS', 0) * 4 / wc, m.get('SQ_ACTIVE_INST_VMEM', 0) * 4 / wc, m.get('SQ_ACTIVE_INST_SCA', 0) * 4 / wc))
    print("  per wave: valu %.0f salu %.0f smem %.0f lds %.0f (bank conflict cycles %.0f) vmem rd %.0f wr %.0f | vmem rd cycles %.0f wr cycles %.0f" % (
        m['SQ_INSTS_VALU'] / w, m.get('SQ_INSTS_SALU', 0) / w, m.get('SQ_INSTS_SMEM', 0) / w, m.get('SQ_INSTS_LDS', 0) / w, m.get('SQ_LDS_BANK_CONFLICT', 0) / w,
        m.get('SQ_INSTS_VMEM_RD', 0) / w, m.get('SQ_INSTS_VMEM_WR', 0) / w, m.get('SQ_INST_CYCLES_VMEM_RD', 0) / w, m.get('SQ_INST_CYCLES_VMEM_WR', 0) / w))
    if cyc:
        print("  chip: valu busy %.2f of 1024 SIMDs   wr data fifo full %.3f  addr fifo full %.3f (of busy cycles)" % (
            m['SQ_ACTIVE_INST_VALU'] * 4 / (cyc * 1024), m.get('SQ_VMEM_WR_TA_DATA_FIFO_FULL', 0) / max(1, m['SQ_BUSY_CYCLES']), m.get('SQ_VMEM_TA_ADDR_FIFO_FULL', 0) / max(1, m['SQ_BUSY_CYCLES'])))
PY
