mkdir -p gpurun_out
run() { # name env... -- args
  name=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras "$@" > gpurun_out/ab_$name.json 2>gpurun_out/ab_$name.err || { echo FAIL $name; tail -3 gpurun_out/ab_$name.err; exit 1; }
}
