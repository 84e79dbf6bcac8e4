# A/B of two builds of the library on one GPU box:  bash tools/ab_lib.sh <variant>
#   ocean-bgc_amd/libobgc.so against ocean-bgc_amd/libobgc_<variant>.so (make BUILD=build_<variant> OUT=.../libobgc_<variant>.so EXTRA=...):
#   first every output of the test call set, bit for bit (tools/output_hashes.py), then interleaved timings of cfg4 / cfg2 with
#   and without diagnostics.
V=$1
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
P=$PWD/ocean-bgc_amd
timeout -k 10 300 python tools/output_hashes.py gpurun_out/bitsum_base.json || exit 1
OBGC_LIB=$P/libobgc_$V.so timeout -k 10 300 python tools/output_hashes.py gpurun_out/bitsum_$V.json || exit 1
python3 - $V <<'PY'
import json,sys
a=json.load(open('gpurun_out/bitsum_base.json')); b=json.load(open('gpurun_out/bitsum_%s.json'%sys.argv[1]))
diff=[k for k in a if a[k]!=b.get(k)]
print(len(a), 'arrays;', len(diff), 'differ')
for k in diff[:40]: print('  ', k, a[k], b.get(k))
PY
run() { name=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras "$@" > gpurun_out/ab2_$name.json 2>gpurun_out/ab2_$name.err || { echo FAIL $name; tail -3 gpurun_out/ab2_$name.err; exit 1; }
  python3 - gpurun_out/ab2_$name.json $name <<'PY'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
r=j['roofline']
print(sys.argv[2], 'ms/step %.3f'%j['ms_per_step'], 'sched', r.get('schedule'), 'carb %.3f col %.3f call %.3f'%(r['kernel_ms']['bgc_carbonate_kernel'], r['kernel_ms']['bgc_cell_kernel'], r['kernel_ms']['bgc_call_device_ms']), flush=True)
PY
}
for v in base $V base $V; do
  L=$P/libobgc_$v.so; [ $v = base ] && L=$P/libobgc.so
  run c4f_$v OBGC_LIB=$L -- --steps 10 || exit 1
  run c4n_$v OBGC_LIB=$L -- --diag none --steps 10 || exit 1
  run c2f_$v OBGC_LIB=$L -- --config cfg2 --steps 40 --warmup 5 || exit 1
  run c2n_$v OBGC_LIB=$L -- --config cfg2 --diag none --steps 40 --warmup 5 || exit 1
done
