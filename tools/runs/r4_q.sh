cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_q; mkdir -p $O
bash tools/runs/quick_perf.sh r4_q cfg2 cfg5 cfg4
( time timeout -k 10 900 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
python - <<PY
import time, numpy as np, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from crbm_amd import CRBM
import bench
for K, M, ds, B, L in ((300, 10, False, 256, 200), (120, 40, True, 256, 200), (8, 100, True, 256, 400)):
    m = CRBM(K, M, doublestranded=ds, batchsize=B, cd_k=1, fantasy_hidden_len=L - M + 1, seed=1)
    D = bench.synthetic_onehot(B, L, seed=2)
    m._trainingFct(D); m.gibbsSteps(1)
    t = time.perf_counter(); m.gibbsSteps(10); g = (time.perf_counter() - t) / 10
    t = time.perf_counter()
    for _ in range(5): m._trainingFct(D)
    tr = (time.perf_counter() - t) / 5
    print("generic kernels %3d x %3d ds=%d batch %d x %d: %.2f ms per Gibbs step, %.2f ms per training step (host array)" % (K, M, ds, B, L, 1e3 * g, 1e3 * tr))
PY
