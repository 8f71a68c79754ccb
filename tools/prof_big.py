"""Development tool: training steps and Gibbs steps of a model on the generic kernels (target of rocprofv3 runs).
usage: python tools/prof_big.py K M ds batch L [steps] [input_dims]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from crbm_amd import CRBM  # noqa: E402
from crbm_amd._lib import fptr  # noqa: E402

if __name__ == "__main__":
    K, M, ds, B, L = (int(x) for x in sys.argv[1:6])
    steps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
    A = int(sys.argv[7]) if len(sys.argv) > 7 else 4
    import warnings
    import numpy as np
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        m = CRBM(K, M, doublestranded=bool(ds), batchsize=B, cd_k=1, fantasy_hidden_len=L - M + 1, seed=1, input_dims=A)
    if A == 4:
        D = bench.synthetic_onehot(B, L, seed=2)
    else:
        D = np.zeros((B, 1, A, L), dtype=np.float32)
        D[np.arange(B)[:, None], 0, np.random.default_rng(2).integers(0, A, size=(B, L)), np.arange(L)[None, :]] = 1
    m._call("crbm_dataset_upload", fptr(D), B, L)
    m._call("crbm_train_step_resident", 0, B)
    t = time.perf_counter()
    for _ in range(steps):
        m._call("crbm_train_step_resident", 0, B)
    print("%d x %d (%d letters) ds=%d batch %d x %d: %.3f ms per training step (resident data)" % (K, M, A, ds, B, L, 1e3 * (time.perf_counter() - t) / steps))
    t = time.perf_counter()
    m.gibbsSteps(steps)
    print("   %.3f ms per Gibbs step" % (1e3 * (time.perf_counter() - t) / steps))
