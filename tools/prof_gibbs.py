"""Development tool: N back-to-back Gibbs launches of a bench config, nothing
else -- the target of rocprofv3 runs (kernel trace / PMC passes)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    model = bench.build_model(cfg, 1, 0, 0)
    model._call("crbm_gibbs_steps", 10)
    ms = ctypes.c_float()
    model._call("crbm_time_gibbs", cfg["k"], n, ctypes.byref(ms))
    print("us/launch", 1e3 * ms.value / n)
