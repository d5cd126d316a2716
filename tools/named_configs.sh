#!/bin/bash
# End-to-end record of BASELINE.json's named configs through the examples/ drivers at full size, at the current HEAD:
#   gpurun --timeout 1100 -- 'bash tools/named_configs.sh r03'   ->  gpurun_out/prof/r03_named_configs.txt  (copy into profiles/)
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
F=$OUT/${TAG}_named_configs.txt
cd $REPO
: > $F
run() { echo "== $1: python $2" >> $F; timeout -k 10 ${3:-300} python $2 >> $F 2>&1 || echo "   (exit code $?)" >> $F; echo "[named] $1 done"; }
run "C2" "examples/c2_solidbody_pdeco_finaltime.py --iters 20"
run "C3 as the named script sets it up (time-dependent wind, all-time misfit)" "examples/c3_c4_systems_pdeco.py schnak --iters 10 --named-c3"
run "C3" "examples/c3_c4_systems_pdeco.py schnak --iters 10 --optim alltime"
run "C3 (HEAD driver, final-time)" "examples/c3_c4_systems_pdeco.py schnak --iters 10"
run "C4" "examples/c3_c4_systems_pdeco.py chtxs --iters 10"
run "nonlinear" "examples/c3_c4_systems_pdeco.py nonlinear --iters 10"
run "C5 (one GPU)" "examples/c5_beta_sweep.py --iters 5"
run "C1" "examples/c1_forward_solidbody.py --steps 500"
echo "commit $(cat $REPO/.git_head 2>/dev/null) source_sha16 $(python -c 'import bench; print(bench.source_sha16())')" >> $F
