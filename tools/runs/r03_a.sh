# Round 3, first GPU pass: the whole -m gpu suite, the bench with its in-run PMC children, the bench as the
# driver runs it, and the multi-rank rehearsals on ONE GPU (CRBM_BENCH_SHARE_GPU=1; the pool's process guard
# allows 6 GPU processes, so the "8-GPU" plumbing is rehearsed with 6 ranks).   usage: bash tools/runs/r03_a.sh <tag>
TAG=${1:-r03a}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
( time timeout -k 10 1000 python -m pytest tests -q -m gpu -x ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
( time timeout -k 10 400 python3 bench.py ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench rc=$?"; tail -3 $O/bench_cfg2.err
( time timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench driver-style rc=$?"
CRBM_BENCH_SHARE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 6 --steps 20 --warmup 5 > $O/bench_6rank_one_gpu_rehearsal.json 2> $O/bench_6rank.err; echo "6-rank rehearsal rc=$?"; tail -3 $O/bench_6rank.err
CRBM_BENCH_SHARE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 4 --config cfg5 --steps 20 --warmup 5 > $O/bench_cfg5_4rank_one_gpu_rehearsal.json 2> $O/bench_cfg5_4rank.err; echo "cfg5 4-rank rehearsal rc=$?"; tail -3 $O/bench_cfg5_4rank.err
python - <<PY
import json
for f in ("bench_cfg2", "bench_driver", "bench_6rank_one_gpu_rehearsal", "bench_cfg5_4rank_one_gpu_rehearsal"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, {k: d[k] for k in ("value", "value_device", "ms_per_step", "device_ms_per_step", "n_gpus", "steps")},
              {k: r.get(k) for k in ("bound", "frac", "algorithmic_frac", "hbm_actual_frac", "traffic", "valu_cycles_per_wave_inst", "counters_source")},
              d.get("train"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
