# Collects the measurement evidence of a round on one MI355X: bench lines, rocprofv3 kernel stats of
# the same command, PMC passes (VALU / LDS / MFMA counters, FETCH_SIZE and WRITE_SIZE in passes of
# their own) for the Gibbs kernel and the statistics kernels.   usage: bash tools/runs/evidence.sh <tag>
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
( time timeout -k 10 400 python bench.py ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench cfg2 rc=$?"
( time timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench driver-style rc=$?"
CRBM_BENCH_SHARE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 6 --steps 20 --warmup 5 > $O/bench_6rank_one_gpu_rehearsal.json 2> $O/bench_6rank.err; echo "6-rank rehearsal rc=$?"
CRBM_BENCH_SHARE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 4 --config cfg5 --steps 20 --warmup 5 > $O/bench_cfg5_4rank_one_gpu_rehearsal.json 2> $O/bench_cfg5_4rank.err; echo "cfg5 4-rank rehearsal rc=$?"
for cfg in cfg4 cfg5; do
  timeout -k 10 300 python bench.py --config $cfg --steps 2000 --warmup 200 --no-cpu-baseline > $O/bench_${cfg}.json 2> $O/bench_${cfg}.err; echo "bench $cfg rc=$?"
done
KS=0,1,2,4,16 timeout -k 10 120 python tools/gibbs_k_scan.py cfg2 > $O/gibbs_steps_per_launch_scan.txt 2>&1
# what a launch costs beyond its steps: the same scan with parts of the launch skipped (CRBM_GIBBS_DEBUG: 1 table copy,
# 2 state load, 4 state store -- with 4 alone the chains stay at h = 0 and the top-down walk has nothing to do)
for dbg in 0 1 2 6 7 4; do echo "== CRBM_GIBBS_DEBUG=$dbg"; CRBM_GIBBS_DEBUG=$dbg KS=0,1,2,16 timeout -k 10 120 python tools/gibbs_k_scan.py cfg2; done > $O/floor_scan.txt 2>&1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc > $O/bench_cfg2_under_rocprof.json 2> $O/bench_cfg2_under_rocprof.err; echo "prof rc=$?"
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
# which LDS access of the statistics kernel conflicts: the stand-alone data-half kernel (CRBM_STATS=split) with phases skipped
# (CRBM_STATS_DEBUG: 1 MFMA steps incl. B-fragment and LUT reads, 2 the h|v arithmetic incl. gathers and transposed stores,
#  4 the staging of the letter windows)
for dbg in 0 1 2 3 4; do
  CRBM_STATS=split CRBM_STATS_DEBUG=$dbg timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $O/pmc_ldsphase_$dbg -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg2 12 > $O/pmc_ldsphase_$dbg.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY > $O/lds_conflict_phases.txt
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import glob, bench
print("crbm_stats_mfma_data (config #2, data half stand-alone), mean per launch")
for dbg, what in ((0, "complete"), (1, "without the MFMA steps (B-fragment reads, LUT reads, MFMAs)"), (2, "without the h|v arithmetic (gathers, sigmoids, transposed stores)"),
                  (3, "without both"), (4, "without the window staging")):
    for f in glob.glob("$O/pmc_ldsphase_%d/**/*counter_collection.csv" % dbg, recursive=True):
        v = bench.pmc_of_csv(f, "crbm_stats_mfma_data")
        print("  debug=%d %-70s conflicts %10.0f of %10.0f LDS-array cycles (%.0f %%); LDS insts %9.0f VALU insts %10.0f; launch %7.0f cycles" % (
            dbg, what, v["SQ_LDS_BANK_CONFLICT"], v["SQ_LDS_IDX_ACTIVE"], 100 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"],
            v["SQ_INSTS_LDS"], v["SQ_INSTS_VALU"], v["SQ_BUSY_CYCLES"] / 32))
PY
for cfg in cfg2 cfg5 cfg4; do bash tools/runs/trace_train.sh $TAG $cfg 60; done > $O/trace_train.txt 2>&1
python tools/pmc_summary.py $O > $O/pmc_summary.txt 2>&1
tail -5 $O/pmc_summary.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split('/')[-1], "value %.0f"%d["value"], "device %.0f"%d["value_device"], "launch_us %.2f"%r["avg_launch_us"], "bound", r["bound"], "frac", r["frac"], "alg_frac %.3f"%r["algorithmic_frac"], "hbm_frac", r.get("hbm_actual_frac"), "train", d["train"].get("ms_per_train_step"), d["train"].get("error"), "copy", r["measured_copy_gbs"])
    except Exception as e: print(f, "ERR", e)
PY
