cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2o
mkdir -p $O
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_driver.json 2> $O/bench_driver.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2o/bench_driver.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","wall_ms_per_step")}, d["train"]["ms_per_train_step"], d["roofline"]["measured_copy_gbs"])
PY
python tools/bench_fit.py cfg2 40 2>&1 | tee $O/fit_cfg2.txt
timeout -k 10 1000 python tools/soak_parity.py 120 11 2>&1 | tee $O/soak.txt | grep -v ": ok" | tail -15
