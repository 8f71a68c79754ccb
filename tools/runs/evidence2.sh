# Second half of a round's evidence on one MI355X: smoke(), the bench exactly as the driver runs it,
# a two-rank rehearsal on the one GPU, fit() wall time, and the randomised parity soak.
#   usage: bash tools/runs/evidence2.sh <tag>
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_2
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; tail -4 $O/bench_driver.err
CRBM_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_2rank_rehearsal.json 2> $O/bench_2rank.err; echo "2-rank rehearsal rc=$?"
python tools/bench_fit.py cfg2 40 > $O/fit_cfg2.txt 2>&1; echo "fit rc=$?"; tail -3 $O/fit_cfg2.txt
timeout -k 10 1000 python tools/soak_parity.py 120 11 > $O/soak.txt 2>&1; echo "soak rc=$?"; grep -v ": ok" $O/soak.txt | tail -8
python - <<PY
import json
d=json.loads(open("$O/bench_driver.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","wall_ms_per_step","n_gpus","steps")}, d["train"], d["cpu_baseline"])
d=json.loads(open("$O/bench_2rank_rehearsal.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","n_gpus")}, d["train"])
PY
