cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2s
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
KS=0,1,2,16 timeout -k 10 120 python tools/gibbs_k_scan.py cfg2
for cfg in cfg2 cfg5 cfg4; do python tools/prof_gibbs.py $cfg 300 | tail -1; done
for cfg in cfg2 cfg5 cfg4; do python tools/prof_train.py $cfg 200 | tail -1; done
