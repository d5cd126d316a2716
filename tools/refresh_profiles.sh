set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
timeout -k 10 400 python bench.py > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/kt -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --batched "" > $GRAFT_REPO_ROOT/gpurun_out/prof/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof/kt.err
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/pf -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --pgd-iters 0 --batched "" --steps 1 --warmup 1 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/pf.err
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/pw -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --pgd-iters 0 --batched "" --steps 1 --warmup 1 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/pw.err
cd $GRAFT_REPO_ROOT/gpurun_out/prof
cp $(find kt -name "*kernel_stats.csv" | head -1) kernel_stats.csv
cp $(find pf -name "*counter_collection.csv" | head -1) fetch.csv
cp $(find pw -name "*counter_collection.csv" | head -1) write.csv
rm -rf kt pf pw
ls -la
