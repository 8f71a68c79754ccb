cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_cfg1; mkdir -p $O
timeout -k 10 200 python tools/prof_fit_cfg1.py 2>&1 | tee $O/fit_profile.txt | head -60
