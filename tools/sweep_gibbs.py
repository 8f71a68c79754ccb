"""Development tool: times the Gibbs kernel under different launch geometries
(env overrides read by crbm_create).  Usage: python tools/sweep_gibbs.py [cfg]"""
import ctypes
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def time_cfg(cfg, env, steps=300):
    for k, v in env.items():
        os.environ[k] = str(v)
    model = bench.build_model(cfg, 1, 0, 0)
    try:
        model._h()
    except Exception as e:
        for k in env:
            os.environ.pop(k, None)
        return None, str(e)
    model._call("crbm_gibbs_steps", 10)
    ms = ctypes.c_float()
    model._call("crbm_time_gibbs", cfg["k"], 30, ctypes.byref(ms))
    model._call("crbm_time_gibbs", cfg["k"], steps, ctypes.byref(ms))
    from crbm_amd import _lib
    info = _lib.CrbmLaunchInfo()
    model._lib.crbm_get_launch_info(model._handle, ctypes.byref(info))
    h, _ = model.get_fantasy()
    res = (1e3 * ms.value / steps, info.gibbs_grid, info.gibbs_block, info.gibbs_seqs_per_tile, info.gibbs_lds_bytes,
           info.group, float(h.mean()))
    for k in env:
        os.environ.pop(k, None)
    del model
    return res, None


if __name__ == "__main__":
    cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
    print("us/launch grid block S lds G activity")
    variants = (sys.argv[2].split(",") if len(sys.argv) > 2 else ["sparse", "dense"])
    Ss = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 3, 4, 5, 6, 8, 10, 16]
    for td, S, thr in itertools.product(variants, Ss, [64, 128, 256]):
        res, err = time_cfg(cfg, {"CRBM_TOPDOWN": td, "CRBM_GIBBS_S": S, "CRBM_GIBBS_THREADS": thr})
        print("%s S=%d thr=%d ->" % (td, S, thr), res if res else err, flush=True)
