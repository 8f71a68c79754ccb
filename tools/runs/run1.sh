set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
./tools/mfma_probe > gpurun_out/r2a/mfma_probe.txt 2>&1; cat gpurun_out/r2a/mfma_probe.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r2a/pytest.log
timeout -k 10 200 python bench.py --steps 400 --warmup 50 --no-cpu-baseline > gpurun_out/r2a/bench_mfma.json 2> gpurun_out/r2a/bench_mfma.err; echo "bench rc=$?"
CRBM_STATS=walk timeout -k 10 200 python bench.py --steps 400 --warmup 50 --no-cpu-baseline > gpurun_out/r2a/bench_walk.json 2> gpurun_out/r2a/bench_walk.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ("bench_mfma","bench_walk"):
    try:
        d=json.loads(open("gpurun_out/r2a/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "value", d["value"], "launch_us", d["roofline"]["avg_launch_us"], "wall", d["wall_ms_per_step"], "train", d["train"], "copy", d["roofline"]["measured_copy_gbs"])
    except Exception as e: print(f, "ERR", e)
PY
