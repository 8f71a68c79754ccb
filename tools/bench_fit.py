"""Development tool: wall time of CRBM.fit() per training step at a bench configuration
(host loop + library, one synchronisation per epoch).  usage: python tools/bench_fit.py [cfg] [batches]"""
import contextlib
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    model = bench.build_model(cfg, 1, 0, 0)
    model.epochs = 3
    codes = np.random.default_rng(1).integers(0, 4, size=(cfg["chains"] * nb, cfg["L"]), dtype=np.uint8)
    test = codes[:cfg["chains"]]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        model.fit(codes[:cfg["chains"]], test)       # warm-up (JIT, allocations)
        t = time.perf_counter()
        model.fit(codes, test)
        dt = time.perf_counter() - t
    print("fit: %.1f us per training step (%d steps, incl. per-epoch evaluation and upload)" % (1e6 * dt / (3 * nb), 3 * nb))
    # the same call once more: the staging buffer and the resident set of this size exist now (the first call's hipFree +
    # hipMalloc of tens of MB are the one-off cost)
    with contextlib.redirect_stdout(buf):
        t = time.perf_counter()
        model.fit(codes, test)
        dt2 = time.perf_counter() - t
    print("fit, second call at the same size: %.1f us per training step; the first call paid %.2f ms more" % (1e6 * dt2 / (3 * nb), 1e3 * (dt - dt2)))
    # the batch loop alone
    model._upload(codes, 0)
    model._call("crbm_train_epoch_resident", cfg["chains"])
    t = time.perf_counter()
    for _ in range(3):
        model._call("crbm_train_epoch_resident", cfg["chains"])
    dt = time.perf_counter() - t
    print("crbm_train_epoch_resident: %.1f us per training step" % (1e6 * dt / (3 * nb)))
    t = time.perf_counter()
    model._upload(codes, 0)
    print("upload of %d sequences: %.2f ms" % (codes.shape[0], 1e3 * (time.perf_counter() - t)))
    t = time.perf_counter()
    for _ in range(5):
        model._call("crbm_dataset_select", 0)
        import ctypes
        mfe, nmh = ctypes.c_float(), ctypes.c_float()
        model._call("crbm_eval_data_resident", 0, cfg["chains"], ctypes.byref(mfe), ctypes.byref(nmh))
        model._evaluateParams()
    print("per-epoch evaluation: %.2f ms" % (1e3 * (time.perf_counter() - t) / 5))
