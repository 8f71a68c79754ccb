cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_eval; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.txt
timeout -k 10 200 python tools/prof_fit_cfg1.py 2>&1 | head -4 | tee $O/fit.txt
timeout -k 10 200 python tools/bench_fit.py cfg2 40 2>&1 | tee $O/fit_cfg2.txt
