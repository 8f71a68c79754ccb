"""Development tool: create / use / destroy handles in a loop -- partitioned chain launches (worker threads, streams,
events), training steps, parameter changes and read-backs interleaved; the chains of two handles with the same seed
must stay identical whatever the interleaving.   usage: python tools/stress_handles.py [rounds] [seed]"""
import gc
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crbm_amd import CRBM  # noqa: E402

if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    shapes = [(10, 15, False, 8192, 186), (10, 15, True, 2048, 200), (20, 15, True, 1024, 120), (4, 6, False, 512, 64), (50, 25, False, 256, 300)]
    for r in range(rounds):
        K, M, ds, B, Lf = shapes[int(rng.integers(0, len(shapes)))]
        parts = [None, "1", "2", "3", "4"][int(rng.integers(0, 5))]
        if parts is None:
            os.environ.pop("CRBM_CHAIN_PARTS", None)
        else:
            os.environ["CRBM_CHAIN_PARTS"] = parts
        a = CRBM(K, M, doublestranded=ds, batchsize=B, cd_k=1, fantasy_hidden_len=Lf, seed=7 + r)
        os.environ["CRBM_CHAIN_PARTS"] = "1"
        b = CRBM(K, M, doublestranded=ds, batchsize=B, cd_k=1, fantasy_hidden_len=Lf, seed=7 + r)
        W = (rng.standard_normal((K, 1, 4, M)) * 0.7).astype(np.float32)
        for m in (a, b):
            m.motifs.set_value(W)
            m.bias.set_value(m.bias.get_value() + 4.0)
        L = Lf + M - 1
        D = np.zeros((64, 1, 4, L), dtype=np.float32)
        D[np.arange(64)[:, None], 0, rng.integers(0, 4, size=(64, L)), np.arange(L)[None, :]] = 1
        for it in range(int(rng.integers(3, 9))):
            op = int(rng.integers(0, 6))
            k = int(rng.integers(1, 6))
            for m in (a, b):
                if op == 0:
                    m.gibbsSteps(k)
                elif op == 1:
                    for _ in range(k):
                        m._call("crbm_gibbs_steps_async", 1)
                    m._call("crbm_sync")
                elif op == 2:
                    m._trainingFct(D)
                elif op == 3:
                    m._call("crbm_gibbs_steps_async", k)
                    m.motifs.set_value(m.motifs.get_value() * np.float32(0.99))     # a parameter change behind queued launches
                elif op == 4:
                    m._call("crbm_gibbs_steps_async", k)
                    m.get_fantasy()                                                 # a read-back behind queued launches
                else:
                    m.freeEnergy(D)
        ha, hb = a.get_fantasy(), b.get_fantasy()
        same = all(np.array_equal(x, y) for x, y in zip(ha, hb) if x is not None)
        np.testing.assert_array_equal(a.motifs.get_value(), b.motifs.get_value())
        print("round %d: %d x %d ds=%d batch %d parts=%s: chains %s, %d units on" % (r, K, M, ds, B, parts or "auto", "identical" if same else "DIFFER", int(ha[0].sum())), flush=True)
        assert same
        del a, b
        gc.collect()
    print("STRESS OK")
