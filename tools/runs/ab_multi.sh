# several variants of the training step on one box, interleaved.  usage: bash tools/runs/ab_multi.sh <tag> <cfg> "<VAR=v ...>" "<VAR=v ...>" ...
TAG=$1; CFG=$2; shift 2
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2 3; do
  for kv in "$@"; do
    echo "[$kv]  $(env $kv timeout -k 10 120 python tools/prof_train.py $CFG ${STEPS:-400} 2>&1 | tail -1)"
  done
done | tee $O/ab_multi_$CFG.txt
