"""Development tool: per-step time of consecutive short timed windows of chain launches (is a short window slow because of
what precedes it?).  usage: python tools/ramp_probe.py [cfg]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
    model = bench.build_model(cfg, 1, 0, 0)
    k = cfg["k"]
    for _ in range(500):
        model._call("crbm_gibbs_steps_async", k)
    model._call("crbm_sync")
    ms = ctypes.c_float()

    mhz = ctypes.c_float()

    def window(n):
        model._call("crbm_time_gibbs", k, n, ctypes.byref(ms))
        model._call("crbm_last_shader_clock", ctypes.byref(mhz))
        return "%.2f@%.0fMHz" % (1e3 * ms.value / n, mhz.value)
    print("10 windows of 20 steps back to back:", " ".join("%s" % window(20) for _ in range(10)), flush=True)
    print("window of 2000:", "%s" % window(2000), flush=True)
    print("5 windows of 20 right behind it:", " ".join("%s" % window(20) for _ in range(5)), flush=True)
    for gap in (0.001, 0.01, 0.1):
        time.sleep(gap)
        print("after %.0f ms idle, windows of 20:" % (1e3 * gap), " ".join("%s" % window(20) for _ in range(4)), flush=True)
    print("windows of 100:", " ".join("%s" % window(100) for _ in range(5)), flush=True)
