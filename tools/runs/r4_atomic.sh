cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_atomic; mkdir -p $O
{ timeout -k 10 120 tools/atomic_probe 1792 1536 2000 && timeout -k 10 120 tools/atomic_probe 1792 900 2000 && timeout -k 10 120 tools/atomic_probe 448 1536 2000; } 2>&1 | tee $O/atomic.txt
