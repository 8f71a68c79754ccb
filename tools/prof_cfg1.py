"""Development tool: epochs of the reference's default model (10 x 15, both strands, batchsize 20, PCD-5) on 1000 x 200 bp
-- BASELINE config #1 -- as crbm_train_epoch_resident runs them (target of rocprofv3 runs).  usage: python tools/prof_cfg1.py [epochs]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from crbm_amd import CRBM  # noqa: E402

if __name__ == "__main__":
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    m = CRBM(10, 15, seed=1)
    D = bench.synthetic_onehot(1000, 200, seed=2)
    m._upload(D, 0)
    m._call("crbm_train_epoch_resident", 20)
    t = time.perf_counter()
    for _ in range(epochs):
        m._call("crbm_train_epoch_resident", 20)
    dt = time.perf_counter() - t
    print("config #1: %.2f ms per epoch of 50 steps (%.1f us per step)" % (1e3 * dt / epochs, 1e6 * dt / epochs / 50))
