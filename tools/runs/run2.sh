set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r2b/pytest.log
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2b/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 50 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2b/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2b/bench_prof.err; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
find gpurun_out/r2b/prof -name "*kernel_stats*" | head; cat $(find gpurun_out/r2b/prof -name "*kernel_stats.csv" | head -1)
