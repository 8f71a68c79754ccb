# pytest -m gpu as the driver runs it, with its wall time
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/tests
mkdir -p $O
( time timeout -k 10 1100 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
