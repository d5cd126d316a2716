#!/bin/bash
# Forward-sweep throughput over the synthetic mesh sizes with the pair-compact Jacobi launch on (default) and
# off (FEMFCT_T4_PAIR=0: the 1024-thread walking launch): where the two cross.
#   gpurun --timeout 900 -- 'bash tools/sizes_pair_crossover.sh r04'   ->  gpurun_out/prof/r04_sizes.txt
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd $REPO
F=$OUT/${TAG}_sizes.txt
SIZES="41 81 257 513 769 1025 1537 2049 3073 4097"
{
  echo "# tools/bench_sizes.py $SIZES (forward sweep, one trajectory; build $(python3 -c 'import bench; print(bench.source_sha16())'))"
  echo "## default (pair-compact walking Jacobi launch where the bandwidth regime walks)"
  timeout -k 10 400 python3 tools/bench_sizes.py $SIZES
  echo "## FEMFCT_T4_PAIR=0 (1024-thread walking launch)"
  FEMFCT_T4_PAIR=0 timeout -k 10 400 python3 tools/bench_sizes.py $SIZES
  echo "## examples/c5_beta_trial_batch.py (64 units, one GPU)"
  (cd examples && timeout -k 10 200 python3 c5_beta_trial_batch.py)
} > $F 2>&1
tail -n 40 $F
