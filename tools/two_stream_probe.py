"""Development tool: does splitting the chains of a Gibbs launch over S handles (streams) whose launches
overlap hide the per-launch floor?  S threads, each timing n back-to-back launches of chains/S chains;
wall clock over all of them against one handle with all chains.   python tools/two_stream_probe.py cfg2 2000"""
import ctypes
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    for S in (1, 2, 4, 1, 2, 4):
        cfg = dict(bench.CONFIGS[name])
        cfg["chains"] //= S
        models = [bench.build_model(cfg, 1, 0, 0) for _ in range(S)]
        for m in models:
            m._call("crbm_gibbs_steps", 500)
        out = [ctypes.c_float() for _ in range(S)]
        bar = threading.Barrier(S + 1)

        def work(i):
            bar.wait()
            models[i]._call("crbm_time_gibbs", cfg["k"], n, ctypes.byref(out[i]))
        th = [threading.Thread(target=work, args=(i,)) for i in range(S)]
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        wall = time.perf_counter() - t0
        print("S=%d: wall %.2f us per step of all chains; per-handle device us/launch %s" % (
            S, 1e6 * wall / n, ["%.2f" % (1e3 * o.value / n) for o in out]), flush=True)
        del models
