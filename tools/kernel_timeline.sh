#!/bin/bash
# Measurement script (start / end of every kernel of a step); run from the repository root on an MI355X (e.g. through gpurun).

cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/tl
rm -rf $D; mkdir -p $D
for mode in none full; do
OBGC_SCHED=${SCHED:-1} rocprofv3 --kernel-trace --output-format csv -d $D/$mode -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --diag $mode > $D/$mode.log 2>&1
python3 - $D/$mode <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'obgc' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step: find last att kernel
idx = [i for i, r in enumerate(rows) if 'bgc_att_kernel' in r['Kernel_Name']][-1]
t0 = int(rows[idx]['Start_Timestamp'])
print(sys.argv[1])
for r in rows[idx:]:
    n = r['Kernel_Name'].split('bgc_')[-1][:20]
    print('  %-28s start %7.3f  end %7.3f  dur %6.3f ms' % (n, (int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6))
PY
done
