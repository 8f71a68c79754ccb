cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2h
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
cd /tmp
for dbg in 0 7 15 23 39 63; do
  CRBM_STATS_DEBUG=$dbg CRBM_STATS=split timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dbg$dbg -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg2 30 > $O/dbg$dbg.log 2>&1
  echo "debug=$dbg"; grep "stats_mfma" $O/dbg$dbg/p_kernel_stats.csv | cut -d, -f1-4
done
