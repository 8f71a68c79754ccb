# gather-table group size A/B of one bench config (chain launch and training step).  usage: bash tools/runs/ab_group.sh <tag> <cfg> "<G ...>"
TAG=$1; CFG=$2; GS=${3:-"2 3"}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do
  for g in $GS; do
    echo "$CFG G=$g  gibbs: $(CRBM_GROUP=$g timeout -k 10 120 python tools/prof_gibbs.py $CFG 400 2>&1 | tail -1)   train: $(CRBM_GROUP=$g timeout -k 10 120 python tools/prof_train.py $CFG 100 2>&1 | tail -1)"
  done
done | tee $O/ab_group_$CFG.txt
