cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_big4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "alphabet or generic or beyond or edge_shapes or pooling" 2>&1 | tail -3 | tee $O/tests.txt
bash tools/runs/r4_big3.sh 2>&1 | tail -12
SHAPES="300 10 0 4096 200" bash tools/runs/r4_profbig.sh 2>&1 | grep "big_" | tee $O/prof.txt
