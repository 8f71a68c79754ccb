cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_big; mkdir -p $O
( time timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "edge_shapes or generic or beyond_the_lds or pooling" ) > $O/pytest_big.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_big.log
for shape in "300 10 0 256 200" "120 40 1 256 200" "8 100 1 256 400" "300 10 0 4096 200"; do timeout -k 10 200 python tools/prof_big.py $shape 10 2>&1 | tail -2; done
