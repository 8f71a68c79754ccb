cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2f
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for dbg in 0 1 2 3 4 7; do
  CRBM_STATS_DEBUG=$dbg CRBM_STATS=split timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dbg$dbg -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg2 30 > $O/dbg$dbg.log 2>&1
  echo "debug=$dbg"; grep "stats_mfma" $O/dbg$dbg/p_kernel_stats.csv | cut -d, -f1-4
done
for thr in 256 1024; do
  CRBM_STATS_THREADS=$thr CRBM_STATS=split timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/thr$thr -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg2 30 > $O/thr$thr.log 2>&1
  echo "threads=$thr"; grep "stats_mfma" $O/thr$thr/p_kernel_stats.csv | cut -d, -f1-4
done
