cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4_split_pmc; rm -rf $O; mkdir -p $O/mix $O/plain
{
echo "three launches (CRBM_STATS=split), mix:   $(CRBM_STATS=split timeout -k 10 120 python tools/prof_train.py cfg2 400 2>&1 | tail -1)"
echo "three launches (CRBM_STATS=split), plain: $(CRBM_STATS=split CRBM_JIT_DEFINES=-DCRBM_SPLIT_PLAIN timeout -k 10 120 python tools/prof_train.py cfg2 400 2>&1 | tail -1)"
} 2>&1 | tee $O/split3.txt
A="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32"
cd /tmp
for pass in a b; do
  case $pass in a) C="$A";; b) C="$B";; esac
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/mix/pmc_train_cfg2_$pass -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg2 20 > $O/mix/pmc_$pass.log 2>&1 &&
  CRBM_JIT_DEFINES=-DCRBM_SPLIT_PLAIN timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/plain/pmc_train_cfg2_$pass -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg2 20 > $O/plain/pmc_$pass.log 2>&1
  CRBM_STATS=split timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/mix/pmc_train_cfg5_$pass -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg2 20 > $O/mix/pmc3_$pass.log 2>&1 &&
  CRBM_STATS=split CRBM_JIT_DEFINES=-DCRBM_SPLIT_PLAIN timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/plain/pmc_train_cfg5_$pass -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg2 20 > $O/plain/pmc3_$pass.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/mix > $O/mix.txt 2>&1; python tools/pmc_summary.py $O/plain > $O/plain.txt 2>&1
find $O -name "*.csv" -size +2M -delete
