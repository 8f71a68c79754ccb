cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_split; mkdir -p $O
{
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "train or stat or step or baseline" 2>&1 | tail -3
for rep in 1 2; do
  echo "cfg2 mix split:   $(timeout -k 10 120 python tools/prof_train.py cfg2 400 2>&1 | tail -1)"
  echo "cfg2 plain split: $(CRBM_JIT_DEFINES=-DCRBM_SPLIT_PLAIN timeout -k 10 120 python tools/prof_train.py cfg2 400 2>&1 | tail -1)"
done
echo "cfg5 mix split:   $(timeout -k 10 120 python tools/prof_train.py cfg5 100 2>&1 | tail -1)"
echo "cfg5 plain split: $(CRBM_JIT_DEFINES=-DCRBM_SPLIT_PLAIN timeout -k 10 120 python tools/prof_train.py cfg5 100 2>&1 | tail -1)"
echo "cfg4 mix split:   $(timeout -k 10 120 python tools/prof_train.py cfg4 60 2>&1 | tail -1)"
echo "cfg4 plain split: $(CRBM_JIT_DEFINES=-DCRBM_SPLIT_PLAIN timeout -k 10 120 python tools/prof_train.py cfg4 60 2>&1 | tail -1)"
} 2>&1 | tee $O/split.txt
