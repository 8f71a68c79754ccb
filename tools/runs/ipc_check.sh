# the mapped-buffer all-reduce on one MI355X: distributed GPU tests, then 2- and 6-rank rehearsals of bench.py
# sharing the card (RCCL cannot form a communicator there; the ipc_all_reduce object of the line is what is checked)
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -q > $O/ipc_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/ipc_pytest.log
[ $rc = 0 ] || exit $rc
for n in 2 6; do
  CRBM_BENCH_SHARE_GPU=1 timeout -k 10 300 python3 bench.py --gpus $n --steps 20 --warmup 5 > $O/ipc_bench_${n}rank.json 2> $O/ipc_bench_${n}rank.err; rc=$?; echo "$n-rank rehearsal rc=$rc"
  [ $rc = 0 ] || exit $rc
done
python - <<PY
import json
for n in (2, 6):
    d = json.loads(open("$O/ipc_bench_%drank.json" % n).read().strip().splitlines()[-1])
    print(n, "ranks: train", {k: d["train"].get(k) for k in ("ms_per_train_step", "all_reduce_us", "error")}, "ipc", d["train"].get("ipc_all_reduce"))
PY
