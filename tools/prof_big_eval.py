"""Development tool: one epoch of training steps against one per-epoch evaluation (convRBM.py:616-625) of a model on the generic
path, both on the resident data set.  usage: python tools/prof_big_eval.py K M ds n L batch"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from crbm_amd import CRBM  # noqa: E402
from crbm_amd._lib import fptr  # noqa: E402

if __name__ == "__main__":
    K, M, ds, n, L, B = (int(x) for x in sys.argv[1:7])
    m = CRBM(K, M, doublestranded=bool(ds), batchsize=B, cd_k=1, fantasy_hidden_len=L - M + 1, seed=1)
    D = bench.synthetic_onehot(n, L, seed=2)
    m._call("crbm_dataset_upload", fptr(D), n, L)
    a, b = ctypes.c_double(), ctypes.c_double()
    for rep in range(2):
        t = time.perf_counter()
        m._call("crbm_train_epoch_resident", B)
        m._call("crbm_sync")
        te = time.perf_counter() - t
        t = time.perf_counter()
        m._call("crbm_eval_epoch_resident", B, ctypes.byref(a), ctypes.byref(b))
        tv = time.perf_counter() - t
    print("%d x %d ds=%d, %d x %d bp in batches of %d: training epoch %.2f ms, evaluation %.2f ms (FE %.3f, NumH %.4f)" % (K, M, ds, n, L, B, 1e3 * te, 1e3 * tv, a.value, b.value))
