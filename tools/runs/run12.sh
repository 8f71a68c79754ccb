cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2l
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
timeout -k 10 300 python bench.py --steps 400 --warmup 20 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2l/bench_cfg2.json").read().strip().splitlines()[-1])
print("value %.0f"%d["value"], "launch_us %.2f"%d["roofline"]["avg_launch_us"], "train", d["train"])
PY
