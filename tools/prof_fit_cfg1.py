"""Development tool: where the wall time of CRBM(10, 15).fit on 1000 x 200 bp (BASELINE config #1, one epoch) goes on the host:
cProfile of the second call.   usage: python tools/prof_fit_cfg1.py"""
import contextlib
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from crbm_amd import CRBM  # noqa: E402

if __name__ == "__main__":
    D = bench.synthetic_onehot(1000, 200, seed=1234)
    for rep in range(3):
        m = CRBM(10, 15, epochs=1, seed=2026)
        m._h()
        buf = io.StringIO()
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(buf):
            if rep == 2:
                pr.enable()
            m.fit(D)
            if rep == 2:
                pr.disable()
        print("fit call %d: %.2f ms" % (rep, 1e3 * (time.perf_counter() - t0)))
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
    print(s.getvalue()[:6000])
