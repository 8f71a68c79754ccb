cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2m
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
cd /tmp
for mode in one two split; do
  if [ $mode = one ]; then unset CRBM_STATS; else export CRBM_STATS=$mode; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/m_$mode -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg2 100 > $O/m_$mode.log 2>&1
  echo "mode=$mode"; grep "us/train" $O/m_$mode.log; cut -d, -f1-4 $O/m_$mode/p_kernel_stats.csv | head -6
done
unset CRBM_STATS
for mode in one two; do
  if [ $mode = one ]; then unset CRBM_STATS; else export CRBM_STATS=$mode; fi
  python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg2 200 2>&1 | tail -1
  python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg2 200 2>&1 | tail -1
done
