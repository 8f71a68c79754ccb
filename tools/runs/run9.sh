cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2i
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for cfg in cfg2 cfg5 cfg4; do
  st=400; [ $cfg = cfg2 ] || st=100
  timeout -k 10 300 python bench.py --config $cfg --steps $st --warmup 20 --no-cpu-baseline > $O/bench_${cfg}.json 2> $O/bench_${cfg}.err; echo "bench $cfg rc=$?"
done
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 50 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2i/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "value %.0f"%d["value"], "launch_us %.2f"%d["roofline"]["avg_launch_us"], "train", d["train"].get("ms_per_train_step"), d["train"].get("error"), "copy", d["roofline"]["measured_copy_gbs"])
    except Exception as e: print(f, "ERR", e)
for f in glob.glob("gpurun_out/r2i/prof/**/*kernel_stats.csv", recursive=True):
    print(open(f).read())
PY
