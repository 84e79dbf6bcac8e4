# the separate kernels' sweep in one launch (OBGC_SPLIT_SWEEP=0) against three (=2) for small native-layout batches
set -e
for n in 256 1024 2048 4096; do
for sp in 0 2; do
  echo "== OBGC_SPLIT_SWEEP=$sp"
  OBGC_SPLIT_SWEEP=$sp timeout -k 10 200 python -u tools/graph_replay_bench.py $n | grep -v amdgpu
done; done
