"""Data-set scale inference sweep: sequences/s of freeEnergy and
the motif-hit summary over a resident data set, with the upload (PCIe) timed
separately and the dense motifHitProbs path beside it for comparison.

usage: python tools/bench_sweep.py [n_sequences] [L] [K] [M] [ds]
"""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crbm_amd import CRBM  # noqa: E402
from crbm_amd._lib import fptr  # noqa: E402


def timed(fn, reps=3):
    fn()
    best = 1e30
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t)
    return best


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    M = int(sys.argv[4]) if len(sys.argv) > 4 else 15
    ds = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
    rng = np.random.default_rng(1234)
    codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    m = CRBM(K, M, doublestranded=ds, batchsize=64, seed=1)
    m.motifs.set_value(np.random.default_rng(42).standard_normal((K, 1, 4, M)).astype(np.float32))
    Lh = L - M + 1
    out = {"n": n, "L": L, "K": K, "M": M, "ds": ds}
    out["upload_s"] = timed(lambda: m._upload(codes, 0))
    fe = np.empty(n, np.float32)
    fem = np.empty((n, K), np.float32)
    out["free_energy_resident_s"] = timed(lambda: m._call("crbm_free_energy_resident", 0, n, fptr(fe), fptr(fem)))
    mx = np.empty((n, K), np.float32)
    mean = np.empty((n, K), np.float32)
    pos = np.empty((K, Lh), np.float32)
    out["hit_summary_resident_s"] = timed(
        lambda: m._call("crbm_hit_summary_resident", 0, n, fptr(mx), fptr(mean), fptr(pos)))
    out["hit_summary_codes_s"] = timed(lambda: m.motifHitSummary(codes), reps=2)
    out["free_energy_codes_s"] = timed(lambda: m.freeEnergy(codes), reps=2)
    nd = min(n, 100000)
    t = timed(lambda: m.motifHitProbs(codes[:nd]), reps=2)
    out["dense_hit_probs_s_per_%d" % nd] = t
    for k in list(out):
        if k.endswith("_s"):
            out[k[:-2] + "_seq_per_s"] = n / out[k]
    out["dense_hit_probs_seq_per_s"] = nd / t
    # the free-energy kernel streams 16*L algorithmic bytes per sequence (fp32 one-hot layout, SURVEY 8(d) convention)
    out["free_energy_algorithmic_GBps"] = n * 16.0 * L / out["free_energy_resident_s"] / 1e9
    print(json.dumps(out))


if __name__ == "__main__":
    main()
