cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then export CRBM_JIT_DEFINES="-DCRBM_AMB_BITS=1"; else unset CRBM_JIT_DEFINES; fi
  echo "variant $v"; KS=1,16 python tools/gibbs_k_scan.py cfg2 2>&1 | tail -2; python tools/prof_train.py cfg2 200 | tail -1
done
done
