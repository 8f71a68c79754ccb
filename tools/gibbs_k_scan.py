"""Development tool: launch time of the Gibbs kernel as a function of steps per launch
(fixed per-launch cost vs per-step cost).  usage: python tools/gibbs_k_scan.py [cfg]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
    model = bench.build_model(cfg, 1, 0, 0)
    model._call("crbm_gibbs_steps", 10)
    ms = ctypes.c_float()
    for k in [int(x) for x in os.environ.get('KS', '0,1,2,4').split(',')]:
        model._call("crbm_time_gibbs", k, 20, ctypes.byref(ms))
        model._call("crbm_time_gibbs", k, 200, ctypes.byref(ms))
        print("k=%d us/launch %.2f us/step %.2f" % (k, 1e3 * ms.value / 200, 1e3 * ms.value / 200 / max(k, 1)), flush=True)
