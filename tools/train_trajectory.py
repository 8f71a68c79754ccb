"""Development tool: device time per training step in consecutive chunks, from the state bench.py's
training section starts in (chains burnt in, parameters as initialised).  python tools/train_trajectory.py cfg2 10 40"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from crbm_amd._lib import fptr  # noqa: E402

if __name__ == "__main__":
    cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    nchunks = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    model = bench.build_model(cfg, 1, 0, 0)
    model._call("crbm_gibbs_steps", 500)
    D = bench.synthetic_onehot(cfg["chains"], cfg["L"], seed=1234)
    model._call("crbm_dataset_upload", fptr(D), cfg["chains"], cfg["L"])
    ms = ctypes.c_float()
    out = []
    for i in range(nchunks):
        model._call("crbm_time_train", 0, cfg["chains"], chunk, ctypes.byref(ms))
        h, _ = model.get_fantasy()
        out.append("%d-%d: %.1f us (activity %.4f)" % (i * chunk, (i + 1) * chunk, 1e3 * ms.value / chunk, float(h.mean())))
    print("\n".join(out))
