# quick numbers on one box: training step and chain launch of the bench configs.  usage: bash tools/runs/quick_perf.sh <tag> [cfgs]
TAG=${1:-quick}; shift; CFGS=${@:-cfg2 cfg5 cfg4}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O
for cfg in $CFGS; do
  n=400; [ $cfg = cfg2 ] || n=100
  for rep in 1 2; do
    echo "$cfg train: $(timeout -k 10 120 python tools/prof_train.py $cfg $n 2>&1 | tail -1)   gibbs: $(timeout -k 10 120 python tools/prof_gibbs.py $cfg $((n*4)) 2>&1 | tail -1)"
  done
done | tee $O/quick_perf.txt
