# Collects the measurement evidence of a round on one MI355X: bench lines, rocprofv3 kernel stats of
# the same command, PMC passes (VALU / LDS / MFMA counters, FETCH_SIZE and WRITE_SIZE in passes of
# their own) for the Gibbs kernel and the statistics kernels.   usage: bash tools/runs/evidence.sh <tag>
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench cfg2 rc=$?"
for cfg in cfg4 cfg5; do
  timeout -k 10 300 python bench.py --config $cfg --steps 2000 --warmup 200 --no-cpu-baseline > $O/bench_${cfg}.json 2> $O/bench_${cfg}.err; echo "bench $cfg rc=$?"
done
KS=0,1,2,4,16 timeout -k 10 120 python tools/gibbs_k_scan.py cfg2 > $O/gibbs_steps_per_launch_scan.txt 2>&1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/bench_cfg2_under_rocprof.json 2> $O/bench_cfg2_under_rocprof.err; echo "prof rc=$?"
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
python tools/pmc_summary.py $O > $O/pmc_summary.txt 2>&1
tail -5 $O/pmc_summary.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "value %.0f"%d["value"], "launch_us %.2f"%d["roofline"]["avg_launch_us"], "frac %.3f"%d["roofline"]["frac"], "train", d["train"].get("ms_per_train_step"), d["train"].get("error"), "copy", d["roofline"]["measured_copy_gbs"])
    except Exception as e: print(f, "ERR", e)
PY
