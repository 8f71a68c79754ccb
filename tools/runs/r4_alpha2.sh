cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_alpha; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "alphabet" 2>&1 | tail -12 | tee $O/tests2.txt
