#!/bin/bash
# Measurement script (kernel stats and HBM traffic of the 100 000-column workload); run from the repository root on an MI355X (e.g. through gpurun).

# kernel stats + HBM traffic (separate PMC passes) for the current kernel revision
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
REV=$(cat ocean-bgc_amd/KERNEL_REV)
D=gpurun_out/prof_$REV
rm -rf $D; mkdir -p $D
for mode in full none; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats_$mode -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --diag $mode > $D/stats_$mode.log 2>&1
  cp $D/stats_$mode/*/*_kernel_stats.csv $D/kernel_stats_cfg2_diag_$mode.csv
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch_$mode -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --diag $mode > $D/fetch_$mode.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write_$mode -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --diag $mode > $D/write_$mode.log 2>&1
done
python3 tools/pmc_summary.py $D/pmc_traffic.json cfg2_full=$D/fetch_full,$D/write_full cfg2_none=$D/fetch_none,$D/write_none > $D/pmc_summary.log 2>&1
tail -5 $D/stats_full.log
head -12 $D/kernel_stats_cfg2_diag_full.csv | cut -c1-160
