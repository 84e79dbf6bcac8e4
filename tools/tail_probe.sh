set -e
for lay in tiled colfast; do
for n in 65536 98304 100000 114688 131072 163840; do
  echo "== $lay $n"
  timeout -k 10 120 python -u bench.py --config cfg2 --columns $n --layout $lay --no-cpu-baseline --no-extras --steps 60 --warmup 5 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms_per_step %.3f  per Mcol %.3f'%(d['ms_per_step'], d['ms_per_step']/($n/1e6)))"
done; done
