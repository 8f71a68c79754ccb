cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2n
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; tail -4 $O/bench_driver.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2n/bench_driver.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","wall_ms_per_step","n_gpus","steps")}, d["train"], d["cpu_baseline"])
print(d["roofline"]["traffic"], d["roofline"]["traffic_source"])
PY
CRBM_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_2rank_rehearsal.json 2> $O/bench_2rank.err; echo "2-rank rehearsal rc=$?"; cat $O/bench_2rank_rehearsal.json | cut -c1-400
