cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_s; mkdir -p $O
{
for rep in 1 2; do
  echo "cfg2: $(timeout -k 10 120 python tools/prof_gibbs.py cfg2 3000 2>&1 | tail -1)"
  echo "cfg5: $(timeout -k 10 120 python tools/prof_gibbs.py cfg5 600 2>&1 | tail -1)   train: $(timeout -k 10 160 python tools/prof_train.py cfg5 100 2>&1 | tail -1)"
done
} 2>&1 | tee $O/s2.txt
( time timeout -k 10 900 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
