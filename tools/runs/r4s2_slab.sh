# slabbed kernels of generic models: tests, then timing with and without (CRBM_SLAB_STATS=0: generic kernels alone)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4s2_slab; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=150 -k "slabbed or generic or beyond or edge_shapes or pooling" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for s in 0 1; do
  for shape in "300 10 0 256 200" "120 40 1 256 200" "300 10 0 4096 200" "256 4 1 4096 200" "120 40 1 2048 200" "8 100 1 256 400"; do
    echo "CRBM_SLAB_STATS=$s: $(CRBM_SLAB_STATS=$s timeout -k 10 200 python tools/prof_big.py $shape 10 2>&1 | tr '\n' ' ')"
  done
done | tee $O/timing.txt
