"""The committed fixtures must still be what the oracle produces (guards both
against drift)."""
import os

import numpy as np

from oracle.crbm_oracle import OracleCRBM


def test_oracle_reproduces_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "crbm_golden.npz"))
    for tag in ("ss", "ds"):
        ds = tag == "ds"
        o = OracleCRBM(int(g[tag + "_K"]), int(g[tag + "_M"]), doublestranded=ds, batchsize=int(g[tag + "_B"]),
                       cd_k=int(g[tag + "_cdk"]), fantasy_hidden_len=int(g[tag + "_Lf"]), seed=int(g[tag + "_seed"]),
                       rho=float(g[tag + "_rho"]), W=g[tag + "_W"])
        o.b = g[tag + "_b"].astype(np.float64)
        o.c = g[tag + "_c"].astype(np.float64)
        D = g[tag + "_D"]
        np.testing.assert_allclose(o._bottomUpActivity(D), g[tag + "_act"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(o.freeEnergy(D), g[tag + "_fe"], rtol=1e-6)
        for _ in range(int(g[tag + "_steps"])):
            o.train_step(D)
        np.testing.assert_allclose(o.W, g[tag + "_W_after"], rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(o.fantasy_h, g[tag + "_fh_after"])
