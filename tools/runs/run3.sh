set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2c
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
for mode in fused split walk; do
  if [ $mode = fused ]; then unset CRBM_STATS; else export CRBM_STATS=$mode; fi
  timeout -k 10 200 python bench.py --steps 400 --warmup 50 --no-cpu-baseline > $O/bench_cfg2_$mode.json 2> $O/bench_cfg2_$mode.err; echo "bench $mode rc=$?"
done
unset CRBM_STATS
for cfg in cfg5 cfg4; do
  for mode in fused split; do
    if [ $mode = fused ]; then unset CRBM_STATS; else export CRBM_STATS=$mode; fi
    timeout -k 10 300 python bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_${cfg}_$mode.json 2> $O/bench_${cfg}_$mode.err; echo "bench $cfg $mode rc=$?"
  done
done
unset CRBM_STATS
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 50 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2c/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "value %.0f"%d["value"], "launch_us %.2f"%d["roofline"]["avg_launch_us"], "train", d["train"].get("ms_per_train_step"), d["train"].get("error"), "copy", d["roofline"]["measured_copy_gbs"])
    except Exception as e: print(f, "ERR", e)
for f in glob.glob("gpurun_out/r2c/prof/**/*kernel_stats.csv", recursive=True):
    print(open(f).read())
PY
