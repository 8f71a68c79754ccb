"""The C restatement used as bench.py's CPU baseline (oracle/crbm_cpu.c) must
agree with the NumPy oracle: identical samples (shared Philox stream), hidden
probabilities within float32 rounding."""
import ctypes

import numpy as np
import pytest

from oracle import build_cpu
from oracle.crbm_oracle import OracleCRBM


@pytest.mark.parametrize("ds", [False, True])
def test_c_port_matches_numpy_oracle(ds):
    lib = ctypes.CDLL(build_cpu.build())
    K, M, B, Lf = 10, 15, 12, 50
    o = OracleCRBM(K, M, doublestranded=ds, batchsize=B, fantasy_hidden_len=Lf, seed=77)
    o.b = (o.b + 5.0).astype(np.float32).astype(np.float64)
    o.seq_offset = 3
    rng = np.random.default_rng(1)
    h = rng.binomial(1, 0.05, size=(B, K, 1, Lf)).astype(np.float32)
    hp = rng.binomial(1, 0.05, size=(B, K, 1, Lf)).astype(np.float32)
    o.fantasy_h = h.astype(np.float64)
    o.fantasy_h_prime = hp.astype(np.float64) if ds else None
    W = np.ascontiguousarray(o.W.reshape(K, 4, M), dtype=np.float32)
    b = np.ascontiguousarray(o.b.ravel(), dtype=np.float32)
    c = np.ascontiguousarray(o.c.ravel(), dtype=np.float32)
    v = np.zeros((B, 1, 4, Lf + M - 1), dtype=np.float32)
    ph = np.zeros_like(h)
    F = ctypes.POINTER(ctypes.c_float)
    P = lambda a: a.ctypes.data_as(F)
    for step in range(2):
        lib.crbm_cpu_gibbs_step(P(W), P(b), P(c), K, M, int(ds), P(h), P(hp), P(v), P(ph), None, B, Lf,
                                ctypes.c_uint64(77), ctypes.c_uint32(step), ctypes.c_uint32(3), 2)
        Pm, _, vm = o.gibbs_steps(1)
        assert (v != vm).mean() < 1e-3
        assert (h != o.fantasy_h).mean() < 1e-3
        np.testing.assert_allclose(ph, Pm, rtol=2e-4, atol=1e-7)
        if ds:
            assert (hp != o.fantasy_h_prime).mean() < 1e-3
    assert h.sum() > 0
