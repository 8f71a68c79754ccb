"""Development tool: N full training steps of a bench config on resident data
(target of rocprofv3 runs)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from crbm_amd._lib import fptr  # noqa: E402

if __name__ == "__main__":
    cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    model = bench.build_model(cfg, 1, 0, 0)
    D = bench.synthetic_onehot(cfg["chains"], cfg["L"], seed=1234)
    model._call("crbm_dataset_upload", fptr(D), cfg["chains"], cfg["L"])
    ms = ctypes.c_float()
    model._call("crbm_time_train", 0, cfg["chains"], 3, ctypes.byref(ms))
    model._call("crbm_time_train", 0, cfg["chains"], n, ctypes.byref(ms))
    print("us/train step", 1e3 * ms.value / n)
