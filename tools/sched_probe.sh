# cfg2 / cfg3 / cfg4 step time under each arrangement of the carbonate solve (OBGC_SCHED), both layouts
set -e
for cfg in cfg2 cfg3; do
for lay in tiled colfast; do
for sched in default 0 2; do
  if [ $sched = default ]; then unset OBGC_SCHED; else export OBGC_SCHED=$sched; fi
  echo -n "$cfg $lay OBGC_SCHED=$sched: "
  timeout -k 10 120 python -u bench.py --config $cfg --layout $lay --no-cpu-baseline --no-extras --steps 40 --warmup 5 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms_per_step %.3f'%d['ms_per_step'])"
done; done; done
