cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_quick; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
