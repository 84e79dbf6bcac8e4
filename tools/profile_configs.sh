#!/bin/bash
# rocprofv3 kernel stats of the larger configurations (parity-test sizes, not bench lines) and of the reference-layout
# device path; run from the repository root on an MI355X (e.g. through gpurun).  Summaries land in gpurun_out/prof_cfgs.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
REV=$(cat ocean-bgc_amd/KERNEL_REV)
D=gpurun_out/prof_cfgs
rm -rf $D; mkdir -p $D
for cfg in cfg3 cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/$cfg -- python3 bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline > $D/$cfg.log 2>&1 || exit 1
  cp $D/$cfg/*/*_kernel_stats.csv $D/${REV}_kernel_stats_${cfg}_diag_full.csv
  echo "== $cfg" && head -14 $D/${REV}_kernel_stats_${cfg}_diag_full.csv | cut -c1-150
done
export OBGC_BENCH_HYBRID_ONLY=1
rocprofv3 --kernel-trace --stats --output-format csv -d $D/kfast -- python3 tools/reference_layout_bench.py > $D/kfast.log 2>&1 || exit 1
cp $D/kfast/*/*_kernel_stats.csv $D/${REV}_kernel_stats_cfg2_reference_layout.csv
echo "== reference layout" && head -16 $D/${REV}_kernel_stats_cfg2_reference_layout.csv | cut -c1-150
