# round 4: the data-parallel GPU tests (mapped-buffer all-reduce: skew, device-side wait, dead rank)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_dist; mkdir -p $O
( time timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu ) > $O/pytest_dist.log 2>&1; echo "pytest dist rc=$?"; tail -15 $O/pytest_dist.log
