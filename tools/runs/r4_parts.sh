# round 4: partitioned chain launches (CRBM_CHAIN_PARTS) A/B + full GPU tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_parts; mkdir -p $O
{
for rep in 1 2; do
  for p in 0 1 2 3 4; do echo "cfg2 parts=$p gibbs: $(CRBM_CHAIN_PARTS=$p timeout -k 10 120 python tools/prof_gibbs.py cfg2 3000 2>&1 | tail -1)"; done
  for p in 1 2; do echo "cfg5 parts=$p gibbs: $(CRBM_CHAIN_PARTS=$p timeout -k 10 120 python tools/prof_gibbs.py cfg5 600 2>&1 | tail -1)"; done
  for p in 1 2; do echo "cfg4 parts=$p gibbs: $(CRBM_CHAIN_PARTS=$p timeout -k 10 120 python tools/prof_gibbs.py cfg4 100 2>&1 | tail -1)"; done
done
echo "cfg2 parts=1 solo G=3: $(CRBM_CHAIN_PARTS=1 CRBM_GROUP_SOLO=3 timeout -k 10 120 python tools/prof_gibbs.py cfg2 3000 2>&1 | tail -1)"
echo "cfg2 parts=1 solo G=4: $(CRBM_CHAIN_PARTS=1 CRBM_GROUP_SOLO=4 timeout -k 10 120 python tools/prof_gibbs.py cfg2 3000 2>&1 | tail -1)"
} 2>&1 | tee $O/parts.txt
( time timeout -k 10 1000 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; cut -c1-600 $O/bench_driver.json
