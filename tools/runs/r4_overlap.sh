cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_overlap; mkdir -p $O
{
for cfg in cfg2 cfg5 cfg4; do
  n=400; [ $cfg = cfg2 ] || n=100; [ $cfg = cfg4 ] && n=40
  echo "$cfg default:                 $(timeout -k 10 120 python tools/prof_train.py $cfg $n 2>&1 | tail -1)"
  echo "$cfg overlap + two launches:  $(CRBM_OVERLAP=1 CRBM_STATS=two timeout -k 10 120 python tools/prof_train.py $cfg $n 2>&1 | tail -1)"
  echo "$cfg overlap + three:         $(CRBM_OVERLAP=1 CRBM_STATS=split timeout -k 10 120 python tools/prof_train.py $cfg $n 2>&1 | tail -1)"
  echo "$cfg overlap alone:           $(CRBM_OVERLAP=1 timeout -k 10 120 python tools/prof_train.py $cfg $n 2>&1 | tail -1)"
done
} 2>&1 | tee $O/overlap.txt
