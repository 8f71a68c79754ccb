cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_alpha; mkdir -p $O
{ timeout -k 10 200 python tools/prof_big.py 32 12 0 256 300 10 20 && timeout -k 10 200 python tools/prof_big.py 32 12 0 4096 300 5 20 && timeout -k 10 200 python tools/prof_big.py 10 15 1 256 200 10 3 && timeout -k 10 200 python tools/prof_big.py 10 15 1 256 200 10 4; } 2>&1 | tee $O/perf.txt
