# round 4: full GPU test suite + quick numbers + the two-stream probe
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_full; mkdir -p $O
( time timeout -k 10 1000 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
{
for geo in "1024 32" "512 16" "256 8"; do set -- $geo
  echo "== half-batch geometry threads=$1 S=$2"; CRBM_GIBBS_THREADS=$1 CRBM_GIBBS_S=$2 timeout -k 10 200 python tools/two_stream_gibbs.py 3000
done
echo "== 4 parts, 256 x 8"; PARTS=4 CRBM_GIBBS_THREADS=256 CRBM_GIBBS_S=8 timeout -k 10 200 python tools/two_stream_gibbs.py 3000
echo "== default geometry"; timeout -k 10 200 python tools/two_stream_gibbs.py 3000
} > $O/two_stream.txt 2>&1; cat $O/two_stream.txt
bash tools/runs/quick_perf.sh r4_full cfg2
