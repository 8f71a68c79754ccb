cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_stress; mkdir -p $O
timeout -k 10 900 python tools/stress_handles.py 40 3 2>&1 | tail -45 | tee $O/stress.txt
