"""Development tool: host wall time vs device time of N training steps (is the host keeping up?).
usage: python tools/train_wall.py [cfg] [n ...]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from crbm_amd._lib import fptr  # noqa: E402

if __name__ == "__main__":
    cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
    ns = [int(x) for x in sys.argv[2:]] or [50, 200, 800]
    model = bench.build_model(cfg, 1, 0, 0)
    D = bench.synthetic_onehot(cfg["chains"], cfg["L"], seed=1234)
    model._call("crbm_dataset_upload", fptr(D), cfg["chains"], cfg["L"])
    ms = ctypes.c_float()
    model._call("crbm_time_train", 0, cfg["chains"], 5, ctypes.byref(ms))
    for n in ns:
        t0 = time.perf_counter()
        model._call("crbm_time_train", 0, cfg["chains"], n, ctypes.byref(ms))
        wall = time.perf_counter() - t0
        print("n=%d device us/step %.2f wall us/step %.2f" % (n, 1e3 * ms.value / n, 1e6 * wall / n), flush=True)
