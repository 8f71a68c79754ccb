"""Development tool: do chain launches of two halves of the batch on two streams hide each other's launch gaps?
Two handles of 4096 chains each (one stream per handle), launches enqueued alternately, against one handle of 8192.
usage: python tools/two_stream_gibbs.py [launches]   (CRBM_GIBBS_THREADS / CRBM_GIBBS_S choose the half-batch geometry)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(models, launches):
    for m in models:
        m._call("crbm_gibbs_steps", 10)
    for rep in range(2):
        for m in models:
            m._call("crbm_sync")
        t0 = time.perf_counter()
        for _ in range(launches):
            for m in models:
                m._call("crbm_gibbs_steps_async", 1)
        for m in models:
            m._call("crbm_sync")
        dt = time.perf_counter() - t0
    return 1e6 * dt / launches


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    cfg = dict(bench.CONFIGS[os.environ.get("CFG", "cfg2")])
    one = bench.build_model(cfg, 1, 0, 0)
    print("one handle, %d chains: %.2f us per step of the whole batch" % (cfg["chains"], run([one], n)), flush=True)
    del one
    parts = int(os.environ.get("PARTS", "2"))
    cfg["chains"] //= parts
    halves = [bench.build_model(cfg, 1, 0, 0) for _ in range(parts)]
    for i, m in enumerate(halves):
        m._call("crbm_set_shard", i * cfg["chains"])
    print("%d handles of %d chains: %.2f us per step of the whole batch" % (parts, cfg["chains"], run(halves, n)), flush=True)
