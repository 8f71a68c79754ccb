cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_rows; mkdir -p $O
{
for rep in 1 2; do
  for r in 0 640 704 768 832 896 1024; do echo "cfg2 stats rows=$r: $(CRBM_STATS_ROWS=$r timeout -k 10 120 python tools/prof_train.py cfg2 400 2>&1 | tail -1)"; done
done
} 2>&1 | tee $O/rows.txt
