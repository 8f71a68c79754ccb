# round 4, first A/B on one box: table group of the solo chain launch, Philox rounds (timing only: samples differ from the oracle's)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_ab1; mkdir -p $O
{
for rep in 1 2; do
  for cfg in cfg2 cfg5 cfg4; do
    n=1600; [ $cfg = cfg2 ] || n=300
    for g in 0 4; do
      echo "$cfg G=$g rounds=10 gibbs: $(CRBM_GROUP=$g timeout -k 10 120 python tools/prof_gibbs.py $cfg $n 2>&1 | tail -1)"
    done
    for r in 7 6; do
      echo "$cfg G=0 rounds=$r gibbs: $(CRBM_JIT_DEFINES=-DCRBM_PHILOX_ROUNDS=$r timeout -k 10 160 python tools/prof_gibbs.py $cfg $n 2>&1 | tail -1)   train: $(CRBM_JIT_DEFINES=-DCRBM_PHILOX_ROUNDS=$r timeout -k 10 160 python tools/prof_train.py $cfg $((n/4)) 2>&1 | tail -1)"
    done
    echo "$cfg G=0 rounds=10 train: $(timeout -k 10 160 python tools/prof_train.py $cfg $((n/4)) 2>&1 | tail -1)"
  done
  echo "cfg2 G=4 rounds=7 gibbs: $(CRBM_GROUP=4 CRBM_JIT_DEFINES=-DCRBM_PHILOX_ROUNDS=7 timeout -k 10 160 python tools/prof_gibbs.py cfg2 1600 2>&1 | tail -1)"
done
} 2>&1 | tee $O/ab1.txt
