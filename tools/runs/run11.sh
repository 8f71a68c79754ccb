cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2k
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for mt in 32 16; do
  CRBM_JIT_DEFINES="-DCRBM_STATS_MAX_TILES=$mt" CRBM_STATS_MAX_TILES=$mt timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mt$mt -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg4 8 > $O/mt$mt.log 2>&1
  echo "max_tiles=$mt"; tail -1 $O/mt$mt.log; cut -d, -f1-4 $O/mt$mt/p_kernel_stats.csv | head -6
done
for mt in 32 16; do
  CRBM_JIT_DEFINES="-DCRBM_STATS_MAX_TILES=$mt" CRBM_STATS_MAX_TILES=$mt timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5mt$mt -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg5 20 > $O/c5mt$mt.log 2>&1
  echo "cfg5 max_tiles=$mt"; tail -1 $O/c5mt$mt.log; cut -d, -f1-4 $O/c5mt$mt/p_kernel_stats.csv | head -6
done
