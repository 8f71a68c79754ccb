cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_p3; mkdir -p $O
{
for rep in 1 2; do
  for p in 2 3 4; do echo "cfg2 parts=$p: $(CRBM_CHAIN_PARTS=$p timeout -k 10 120 python tools/prof_gibbs.py cfg2 3000 2>&1 | tail -1)"; done
  for p in 2 3; do echo "cfg5 parts=$p: $(CRBM_CHAIN_PARTS=$p timeout -k 10 120 python tools/prof_gibbs.py cfg5 600 2>&1 | tail -1)"; done
done
} 2>&1 | tee $O/p3.txt
