#!/bin/bash
# One pass on ONE MI355X box that regenerates everything under profiles/ from the library as built from the
# current sources, so that kernel stats, PMC traffic and the bench line describe the same binary:
#   gpurun --timeout 1190 -- 'bash tools/refresh_profiles.sh r04'
# Outputs land in gpurun_out/prof/ (merged back by gpurun); copy them into profiles/ with
#   cp gpurun_out/prof/r04_* gpurun_out/prof/traffic.json profiles/
# Order: PMC passes first (traffic.json, stamped with source_sha16), then the plain bench run (whose
# roofline.traffic echoes that stamp-checked file), then the kernel trace.
set -e
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
N_ROOF=4198401          # 2049^2, the roofline mesh of bench.py
PMC_ARGS="--cpu-sample 0 --pgd-iters 0 --batched= --steps 1 --warmup 1 --systems 0 --tolerance-table 0"
echo "[refresh] PMC FETCH_SIZE pass"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pf -- python3 $REPO/bench.py $PMC_ARGS > /dev/null 2> $OUT/pf.err
echo "[refresh] PMC WRITE_SIZE pass"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pw -- python3 $REPO/bench.py $PMC_ARGS > /dev/null 2> $OUT/pw.err
cp $(find $OUT/pf -name "*counter_collection.csv" | head -1) $OUT/fetch.csv
cp $(find $OUT/pw -name "*counter_collection.csv" | head -1) $OUT/write.csv
cp $REPO/profiles/traffic.json $OUT/traffic.json 2>/dev/null || true
python3 $REPO/tools/pmc_traffic.py $OUT/fetch.csv $OUT/write.csv $N_ROOF $OUT/${TAG} > $OUT/${TAG}_pmc_summary.txt
echo "[refresh] PMC passes at 4097 x 4097"
N_ROOF2=16785409
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pf2 -- python3 $REPO/bench.py $PMC_ARGS --roofline-cells 4096 > /dev/null 2> $OUT/pf2.err
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pw2 -- python3 $REPO/bench.py $PMC_ARGS --roofline-cells 4096 > /dev/null 2> $OUT/pw2.err
cp $(find $OUT/pf2 -name "*counter_collection.csv" | head -1) $OUT/fetch2.csv
cp $(find $OUT/pw2 -name "*counter_collection.csv" | head -1) $OUT/write2.csv
python3 $REPO/tools/pmc_traffic.py $OUT/fetch2.csv $OUT/write2.csv $N_ROOF2 $OUT/${TAG}_4097 > $OUT/${TAG}_pmc_summary_4097.txt
cp $OUT/traffic.json $REPO/profiles/traffic.json      # on the box: the bench runs below read it
echo "[refresh] bench run"
cd $REPO
timeout -k 10 400 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
echo "[refresh] kernel trace"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --cpu-sample 0 --batched= --systems 0 --tolerance-table 0 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/kt.err
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
echo "[refresh] kernel trace of the BATCHED regime (64 C2 trajectories per launch: the lever at the config sizes)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kb -- python3 $REPO/bench.py --batch 64 --steps 3 --warmup 1 --roofline-cells 0 --cpu-sample 0 --pgd-iters 0 --batched= --systems 0 --tolerance-table 0 > $OUT/${TAG}_bench_batch64_under_rocprof.json 2> $OUT/kb.err
cp $(find $OUT/kb -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_batch64_kernel_stats.csv
echo "[refresh] the other systems' sweeps under the same profiler command that used to fault (DESIGN.md section 9)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -- python3 $REPO/tools/bench_systems.py > $OUT/${TAG}_systems_under_rocprof.txt 2>&1 && echo "exit code 0" >> $OUT/${TAG}_systems_under_rocprof.txt || echo "exit code $? (non-zero)" >> $OUT/${TAG}_systems_under_rocprof.txt
cp $(find $OUT/ks -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_systems_kernel_stats.csv 2>/dev/null || true
echo "[refresh] the same sweeps without the profiler (graphs replayed)"
timeout -k 10 300 python3 $REPO/tools/bench_systems.py > $OUT/${TAG}_systems.txt 2>&1
echo "[refresh] kernel trace at 41 x 41 (configs 3 / 4): one workgroup per trajectory vs the tile path, 1 / 64 / 256 trajectories per launch"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/km -- python3 $REPO/tools/mesh_step_check.py 40 50 1,64,256 > $OUT/${TAG}_mesh41_under_rocprof.txt 2> $OUT/km.err
cp $(find $OUT/km -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_mesh41_kernel_stats.csv
echo "[refresh] roofline entry at 4097 x 4097"
cd $REPO
timeout -k 10 400 python3 bench.py --roofline-cells 4096 --steps 2 --warmup 1 --cpu-sample 0 --pgd-iters 0 --batched= --systems 0 --tolerance-table 0 > $OUT/${TAG}_bench_roofline4097.json 2> $OUT/b4097.err
rm -rf $OUT/kt $OUT/kb $OUT/ks $OUT/km $OUT/pf $OUT/pw $OUT/pf2 $OUT/pw2 $OUT/fetch.csv $OUT/write.csv $OUT/fetch2.csv $OUT/write2.csv
ls -la $OUT
