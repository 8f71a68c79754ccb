cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_big; mkdir -p $O
( time timeout -k 10 1000 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout -k 10 900 python tools/soak_parity.py 120 77 > $O/soak_120_77.txt 2>&1; echo "soak rc=$?"; grep -v ": ok" $O/soak_120_77.txt | tail -8; grep -c ": ok" $O/soak_120_77.txt
