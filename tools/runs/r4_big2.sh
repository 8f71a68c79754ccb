cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_big2; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "alphabet or generic or beyond or edge_shapes or pooling" 2>&1 | tail -4 | tee $O/tests.txt
SHAPES="300 10 0 4096 200;300 10 0 256 200;120 40 1 256 200" bash tools/runs/r4_profbig.sh 2>&1 | grep -v rocprofv3 | tee $O/prof.txt
