# Round-4 evidence on one MI355X: GPU tests, bench lines (default / driver style / cfg4 / cfg5), rocprofv3 kernel stats of the
# default bench command, PMC passes of the chain kernel and the training step, per-kernel traces of the training steps,
# the ramp probe.   usage: bash tools/runs/evidence_r4.sh <tag>
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
( time timeout -k 10 500 python bench.py ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench cfg2 rc=$?"
( time timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench driver-style rc=$?"
for cfg in cfg4 cfg5; do
  timeout -k 10 300 python bench.py --config $cfg --steps 1000 --warmup 100 --no-cpu-baseline > $O/bench_${cfg}.json 2> $O/bench_${cfg}.err; echo "bench $cfg rc=$?"
done
timeout -k 10 200 python3 tools/ramp_probe.py cfg2 > $O/ramp_probe.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 200 python tools/bench_fit.py cfg2 40 > $O/fit_cfg2.txt 2>&1; echo "fit rc=$?"; tail -5 $O/fit_cfg2.txt
timeout -k 10 200 python tools/bench_sweep.py > $O/sweep_1M_cfg2model.json 2>&1; echo "sweep rc=$?"
CRBM_GIBBS_TIMELINE=1 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-train --no-other-configs > $O/timeline_20.json 2> $O/timeline_20.txt
bash tools/runs/r4_rehearse.sh > $O/rehearse.txt 2>&1; cp gpurun_out/r4_rehearse/*.json $O/ 2>/dev/null
KS=0,1,2,4,16 timeout -k 10 120 python tools/gibbs_k_scan.py cfg2 > $O/gibbs_steps_per_launch_scan.txt 2>&1
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc --no-other-configs > $O/bench_cfg2_under_rocprof.json 2> $O/bench_cfg2_under_rocprof.err; echo "prof rc=$?"
A="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32"
for cfg in cfg2 cfg4 cfg5; do
  n=20; [ $cfg = cfg2 ] || n=6
  for pass in a b f w; do
    case $pass in a) C="$A";; b) C="$B";; f) C="FETCH_SIZE";; w) C="WRITE_SIZE";; esac
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_train_${cfg}_$pass -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py $cfg $n > $O/pmc_train_${cfg}_$pass.log 2>&1
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_gibbs_${cfg}_$pass -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_gibbs.py $cfg $n > $O/pmc_gibbs_${cfg}_$pass.log 2>&1
  done
  echo "pmc $cfg done"
done
cd $GRAFT_REPO_ROOT
for cfg in cfg2 cfg5 cfg4; do bash tools/runs/trace_train.sh $TAG $cfg 60; done > $O/trace_train.txt 2>&1
python tools/pmc_summary.py $O > $O/pmc_summary.txt 2>&1
tail -5 $O/pmc_summary.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split('/')[-1], "value %.0f"%d["value"], "device %.0f"%d["value_device"], "step_us %.2f"%r["avg_launch_us"], "kernel_avg", r.get("kernel_avg_us"), "bound", r["bound"], "frac", r["frac"], "alg_frac %.3f"%r["algorithmic_frac"], "hbm_frac", r.get("hbm_actual_frac"), "train", d["train"].get("ms_per_train_step"), d["train"].get("error"), "copy", r["measured_copy_gbs"])
    except Exception as e: print(f, "ERR", e)
PY
