cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2p
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for cfg in cfg2 cfg5 cfg4; do python tools/prof_train.py $cfg 100 | tail -1; done
CRBM_STATS=two python tools/prof_train.py cfg2 200 | tail -1
CRBM_STATS=split python tools/prof_train.py cfg2 200 | tail -1
