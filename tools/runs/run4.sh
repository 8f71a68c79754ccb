set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2d
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_a -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg2 20 > $O/pmc_a.log 2>&1; echo "pmc_a rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/pmc_b -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg2 20 > $O/pmc_b.log 2>&1; echo "pmc_b rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg2 20 > $O/pmc_f.log 2>&1; echo "pmc_f rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg2 20 > $O/pmc_w.log 2>&1; echo "pmc_w rc=$?"
cd $GRAFT_REPO_ROOT
for k in crbm_stats_mfma_data crbm_gibbs_sparse_stats; do echo "== $k"; python tools/pmc_gibbs.py $O $k; done > $O/pmc_summary.txt 2>&1
tail -60 $O/pmc_summary.txt
for cfg in cfg5 cfg4; do
    timeout -k 10 300 python bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_${cfg}.json 2> $O/bench_${cfg}.err; echo "bench $cfg rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2d/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "value %.0f"%d["value"], "launch_us %.2f"%d["roofline"]["avg_launch_us"], "train", d["train"].get("ms_per_train_step"), d["train"].get("error"))
    except Exception as e: print(f, "ERR", e)
PY
