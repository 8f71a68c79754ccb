cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_fused; mkdir -p $O
{
for rep in 1 2; do
  echo "fused 256: $(timeout -k 10 200 python tools/prof_train.py cfg2 400 2>&1 | tail -1)"
  echo "fused 512: $(CRBM_FUSED_THREADS=512 CRBM_JIT_DEFINES=-DCRBM_FUSED_TB=512 timeout -k 10 300 python tools/prof_train.py cfg2 400 2>&1 | tail -1)"
  echo "fused 512 rows 320: $(CRBM_STATS_ROWS=320 CRBM_FUSED_THREADS=512 CRBM_JIT_DEFINES=-DCRBM_FUSED_TB=512 timeout -k 10 300 python tools/prof_train.py cfg2 400 2>&1 | tail -1)"
  echo "fused 512 rows 448: $(CRBM_STATS_ROWS=448 CRBM_FUSED_THREADS=512 CRBM_JIT_DEFINES=-DCRBM_FUSED_TB=512 timeout -k 10 300 python tools/prof_train.py cfg2 400 2>&1 | tail -1)"
done
} 2>&1 | tee $O/fused.txt
CRBM_FUSED_THREADS=512 CRBM_JIT_DEFINES=-DCRBM_FUSED_TB=512 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "train_step_trace or baseline_config or full_size or golden" 2>&1 | tail -3
