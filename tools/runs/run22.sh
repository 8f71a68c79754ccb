cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2q
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for cfg in cfg2 cfg5 cfg4; do python tools/prof_train.py $cfg 200 | tail -1; done
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o t -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py cfg2 300 > $O/prof_train.log 2>&1; echo "prof rc=$?"
f=$(ls $O/prof_train/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cut -d, -f1-6 "$f"
