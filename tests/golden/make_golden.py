"""Generates tests/golden/crbm_golden.npz from the float64 oracle.

The reference itself cannot produce vectors here (Theano is not installed and
cannot be), so the fixtures come from oracle/crbm_oracle.py *after* it has been
pinned by tests/test_oracle.py against the control implementations held in the
reference's own tests.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.crbm_oracle import OracleCRBM, synthetic_onehot  # noqa: E402


def case(tag, K, M, ds, B, Lf, cdk, seed, rho, n, L, steps, out):
    rng = np.random.default_rng(seed)
    W = rng.standard_normal((K, 1, 4, M)).astype(np.float32)
    o = OracleCRBM(K, M, doublestranded=ds, batchsize=B, cd_k=cdk, fantasy_hidden_len=Lf, seed=seed, rho=rho, W=W)
    o.b = (o.b + 4.0).astype(np.float32).astype(np.float64)
    o.c = (rng.standard_normal((1, 4)) * 0.1).astype(np.float32).astype(np.float64)
    D = synthetic_onehot(n, L, seed=seed + 1)
    h = rng.binomial(1, 0.1, size=(n, K, 1, L - M + 1)).astype(np.float32)
    hp = rng.binomial(1, 0.1, size=(n, K, 1, L - M + 1)).astype(np.float32)
    f32 = lambda x: np.asarray(x, dtype=np.float32)
    out.update({
        tag + "_K": K, tag + "_M": M, tag + "_B": B, tag + "_Lf": Lf, tag + "_cdk": cdk, tag + "_seed": seed,
        tag + "_rho": rho, tag + "_steps": steps,
        tag + "_W": W, tag + "_b": f32(o.b), tag + "_c": f32(o.c), tag + "_D": D, tag + "_h": h, tag + "_hp": hp,
        tag + "_act": f32(o._bottomUpActivity(D)), tag + "_act_rc": f32(o._bottomUpActivity(D, True)),
        tag + "_hit": f32(o.motifHitProbs(D)), tag + "_fe": f32(o.freeEnergy(D)),
        tag + "_pv": f32(o._computeVgivenH(h, hp if ds else None)[0]),
    })
    for _ in range(steps):
        o.train_step(D)
    out.update({tag + "_W_after": f32(o.W), tag + "_b_after": f32(o.b), tag + "_c_after": f32(o.c),
                tag + "_fh_after": f32(o.fantasy_h)})


if __name__ == "__main__":
    out = {}
    case("ss", 10, 15, False, 8, 60, 1, 17, 0.01, 6, 80, 3, out)
    case("ds", 10, 5, True, 6, 200, 2, 23, 0.02, 5, 200, 2, out)     # reference-test shape: K=10, M=5, L=200
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "crbm_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
