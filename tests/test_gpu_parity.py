"""GPU parity tests: the HIP path (through the C-ABI, via crbm_amd.CRBM) against
the float64 oracle on the same seeded inputs.  Shapes and checks mirror the
reference's tests/testcrbm.py (lines cited per test).  Tolerance for floating
point outputs: 1e-4 relative (BASELINE.json north_star); samples must be
identical given the shared Philox stream, except at |p - u| < 1e-6 ties.
"""
import os

import ctypes

import numpy as np
import pytest

from oracle.crbm_oracle import (OracleCRBM, synthetic_onehot, hidden_uniforms, visible_uniforms,
                                KIND_API_H, KIND_API_V, KIND_CHAIN_H, KIND_CHAIN_V, KIND_EVAL_H)

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def make_pair(K, M, ds=True, batchsize=8, cd_k=2, Lf=200, seed=5, wscale=1.0, bshift=0.0, **kw):
    """A crbm_amd.CRBM and an OracleCRBM with identical parameters and sampler."""
    import warnings
    from crbm_amd import CRBM
    A = kw.get("input_dims", 4)
    rng = np.random.default_rng(1000 + 7 * K + M)
    W = (rng.standard_normal((K, 1, A, M)) * wscale).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)          # "input_dims != 4 was not comprehensively tested" (convRBM.py:84-87)
        m = CRBM(K, M, doublestranded=ds, batchsize=batchsize, cd_k=cd_k, fantasy_hidden_len=Lf, seed=seed, **kw)
    o = OracleCRBM(K, M, doublestranded=ds, batchsize=batchsize, cd_k=cd_k, fantasy_hidden_len=Lf, seed=seed,
                   W=W, **{k: v for k, v in kw.items() if k in ("rho", "lambda_rate", "learning_rate", "momentum", "pooling", "input_dims")})
    b = (o.b + bshift).astype(np.float32)
    c = (rng.standard_normal((1, A)) * 0.1).astype(np.float32)
    m.motifs.set_value(W)
    m.bias.set_value(b)
    m.c.set_value(c)
    o.b, o.c = b.astype(np.float64), c.astype(np.float64)
    return m, o


def assert_samples(got, prob, u):
    want = (prob > u).astype(np.float32)
    bad = got != want
    assert np.all(np.abs(prob - u)[bad] < 1e-6), "sample differs away from a p == u tie"
    assert bad.mean() < 1e-4


TIE = 1e-6


def assert_chain_steps(model, o, steps):
    """`steps` Gibbs steps, one at a time from IDENTICAL states (the HIP chain is reset to the
    oracle's before every step): every visible letter and every hidden unit that differs from the
    oracle's must sit on a p == u tie (|p - u| < 1e-6, the only place where float32 and float64
    arithmetic may legitimately decide differently).  Returns the number of ties met."""
    ds = o.doublestranded
    B, K, _, Lf = o.fantasy_h.shape
    Lv = Lf + o.motif_length - 1
    idx = np.arange(B) + o.seq_offset
    ties = 0
    for _ in range(steps):
        h0, hp0 = o.fantasy_h.astype(np.float32), (o.fantasy_h_prime.astype(np.float32) if ds else None)
        model.set_fantasy(h0, hp0)
        t = o.gibbs_step
        uv = visible_uniforms(o.seed, t, idx, Lv, KIND_CHAIN_V)
        Pv, v = o._computeVgivenH(o.fantasy_h, o.fantasy_h_prime, uv)
        uh = hidden_uniforms(o.seed, t, idx, K, Lf, 0, KIND_CHAIN_H)
        P, h = o._computeHgivenV(v, False, uh)
        if ds:
            uhp = hidden_uniforms(o.seed, t, idx, K, Lf, 1, KIND_CHAIN_H)
            Pp, hp = o._computeHgivenV(v, True, uhp)
        model.gibbsSteps(1)
        gv = model.get_fantasy_visible()
        gh, ghp = model.get_fantasy()
        np.testing.assert_array_equal(gv.sum(axis=2), 1.0)                       # one 1 per position
        badv = (gv != v).any(axis=2)[:, 0]                                       # (B, Lv)
        if badv.any():
            cum = np.cumsum(Pv[:, 0], axis=1)[:, :max(Pv.shape[2] - 1, 1)]       # the inner thresholds (three for DNA)
            gap = np.min(np.abs(cum - uv[:, None, :]), axis=1)
            assert np.all(gap[badv] < TIE), "visible sample differs away from a tie"
            ties += int(badv.sum())
        clean = ~badv.any(axis=1)                                                # chains whose v agrees: h must too
        pool = int(getattr(o, "pooling", 1))
        for got, want, prob, u in ((gh, h, P, uh),) + (((ghp, hp, Pp, uhp),) if ds else ()):
            bad = (got != want) & clean[:, None, None, None]
            if bad.any() and pool == 1:
                assert np.all(np.abs(prob - u)[bad] < TIE), "hidden sample differs away from a tie"
                ties += int(bad.sum())
            elif bad.any():
                # pooled units: one draw per group (the uniform of its first position) against the
                # cumulative probabilities of the group (convRBM.py:259-267)
                badg = bad.reshape(B, K, 1, Lf // pool, pool).any(axis=4)
                cum = np.cumsum(prob.reshape(B, K, 1, Lf // pool, pool), axis=4)
                ug = u.reshape(B, K, 1, Lf // pool, pool)[..., :1]
                assert np.all(np.min(np.abs(cum - ug), axis=4)[badg] < TIE), "pooled hidden sample differs away from a tie"
                ties += int(badg.sum())
        o.fantasy_h, o.fantasy_h_prime = h, (hp if ds else None)
        o.last_v_model = v
        o.gibbs_step += 1
    assert ties <= 1e-4 * steps * B * K * Lf * (2 if ds else 1) + 2
    return ties


def assert_chain_equal_or_tied(model, o, before, steps):
    """The model's chain after `steps` Gibbs steps from `before` = (h, hp, gibbs_step) must EQUAL the oracle's (which
    has taken the same steps).  Where it does not, the steps are replayed one at a time from identical states and
    every differing sample must sit on a p == u tie (assert_chain_steps); the model then continues from the oracle's
    state.  Returns the number of ties (0: bit-identical)."""
    ds = o.doublestranded
    h, hp = model.get_fantasy()
    if np.array_equal(h, o.fantasy_h) and (not ds or np.array_equal(hp, o.fantasy_h_prime)):
        return 0
    h0, hp0, t0 = before
    after = (o.fantasy_h, o.fantasy_h_prime, o.gibbs_step)
    o.fantasy_h, o.fantasy_h_prime, o.gibbs_step = h0.astype(np.float64), (hp0.astype(np.float64) if ds else None), t0
    model.set_rng(gibbs_step=t0)
    ties = assert_chain_steps(model, o, steps)
    assert ties >= 1, "chains differ although no sample sits on a tie"
    assert o.gibbs_step == after[2]
    model.set_fantasy(o.fantasy_h.astype(np.float32), o.fantasy_h_prime.astype(np.float32) if ds else None)
    return ties


def install_oracle_state(model, o):
    """The HIP model takes over the oracle's complete state: parameters, velocities, chains, step counter."""
    ds = o.doublestranded
    model.motifs.set_value(o.W.astype(np.float32))
    model.bias.set_value(o.b.astype(np.float32))
    model.c.set_value(o.c.astype(np.float32))
    model.set_velocities(o.vW.astype(np.float32), o.vb.astype(np.float32), o.vc.astype(np.float32))
    model.set_fantasy(o.fantasy_h.astype(np.float32), o.fantasy_h_prime.astype(np.float32) if ds else None)
    model.set_rng(gibbs_step=o.gibbs_step)


def assert_train_steps(model, o, batches, twin, step=None, rtol=RTOL, atol=2e-6):
    """PCD-k updates one at a time.  After every update the chains and the last visible sample must be
    IDENTICAL to the oracle's and the parameters agree at `rtol`; where a sample differs, the step's chain
    is replayed on a fresh pair (`twin()` -> (model, oracle)) from the state before the step, one Gibbs
    step at a time from identical states (assert_chain_steps: every differing sample must sit on a
    p == u tie), and the HIP model then continues from the oracle's state.  `step(model, D)` runs one
    update (default: the host-array entry point).  Returns the number of updates that met a tie."""
    ds = o.doublestranded
    tied = 0
    for D in batches:
        before = {k: np.copy(getattr(o, k)) for k in ("W", "b", "c", "vW", "vb", "vc", "fantasy_h")}
        before["fantasy_h_prime"] = np.copy(o.fantasy_h_prime) if ds else None
        g0 = o.gibbs_step
        (step or (lambda m, d: m._trainingFct(d)))(model, D)
        o.train_step(D)
        h, hp = model.get_fantasy()
        same = np.array_equal(h, o.fantasy_h) and (not ds or np.array_equal(hp, o.fantasy_h_prime)) \
            and np.array_equal(model.get_fantasy_visible(), o.last_v_model)
        if same:
            np.testing.assert_allclose(model.motifs.get_value(), o.W, rtol=rtol, atol=atol)
            np.testing.assert_allclose(model.bias.get_value(), o.b, rtol=rtol, atol=atol)
            np.testing.assert_allclose(model.c.get_value(), o.c, rtol=rtol, atol=atol)
            vW, vb, vc = model.get_velocities()
            np.testing.assert_allclose(vW, o.vW, rtol=rtol, atol=atol)
            np.testing.assert_allclose(vb, o.vb, rtol=rtol, atol=atol)
            np.testing.assert_allclose(vc, o.vc, rtol=rtol, atol=atol)
        else:
            m2, o2 = twin()
            for k, v in before.items():
                setattr(o2, k, v)
            o2.gibbs_step = g0
            install_oracle_state(m2, o2)
            assert assert_chain_steps(m2, o2, o.cd_k) >= 1, "chains differ although no sample sits on a tie"
            tied += 1
            install_oracle_state(model, o)
    return tied


# ---- reference tests/testcrbm.py:148-200 ------------------------------------
@pytest.mark.parametrize("flip", [False, True])
def test_bottomup(flip):
    data = synthetic_onehot(11, 200, seed=3)
    nmot, mlen = 10, 5
    model, o = make_pair(nmot, mlen)
    w = model.motifs.get_value()[:, :, ::-1, ::-1] if flip else model.motifs.get_value()
    b = model.bias.get_value()
    output = model._bottomUpActivity(data, flip)
    output_control = np.zeros(output.shape)
    for seq in range(data.shape[0]):
        for s in range(data.shape[3] - w.shape[3] + 1):
            for m in range(w.shape[0]):
                output_control[seq, m, 0, s] += \
                    np.multiply(w[m, 0, :, :], data[seq, 0, :, s:(s + w.shape[3])]).sum() + b[0, m]
    np.testing.assert_allclose(output, output_control, rtol=1e-5, atol=1e-5)
    sig = 1. / (1. + np.exp(-output_control))
    np.testing.assert_allclose(sig, model._bottomUpProbabilityOfData(data, flip), rtol=1e-5, atol=1e-5)
    p, h = model._computeHgivenV(data, flip, rng_step=4)
    np.testing.assert_allclose(p, sig, rtol=1e-5, atol=1e-5)
    assert p.shape == (data.shape[0], nmot, 1, data.shape[3] - mlen + 1)
    np.testing.assert_allclose(p, o._computeHgivenV(data, flip)[0], rtol=RTOL, atol=1e-7)
    u = hidden_uniforms(model.seed, 4, np.arange(11), nmot, 196, 1 if flip else 0, KIND_API_H)
    assert_samples(h, p.astype(np.float64), u)


# ---- reference tests/testcrbm.py:319-504 ------------------------------------
@pytest.mark.parametrize("kind", ["zeros", "ones", "random"])
@pytest.mark.parametrize("ds", [False, True])
def test_topdown(kind, ds):
    nseq, seqlen, nmot, mlen = 11, 200, 10, 5
    rng = np.random.default_rng(8)
    shape = (nseq, nmot, 1, seqlen - mlen + 1)

    def hid():
        if kind == "zeros":
            return np.zeros(shape, dtype="float32")
        if kind == "ones":
            return np.ones(shape, dtype="float32")
        return rng.binomial(1, 0.1, size=shape).astype("float32")

    model, o = make_pair(nmot, mlen)
    if kind == "zeros":
        model.c.set_value(np.zeros((1, 4), dtype=np.float32))
        o.c = np.zeros((1, 4))
    data, datap = hid(), (hid() if ds else None)
    oa = model._topDownActivity(data, datap)
    op = model._topDownProbabilityOfHidden(data, datap)
    op2, sample = model._computeVgivenH(data, datap, rng_step=2)
    assert oa.shape == (nseq, 1, 4, seqlen) and op.shape == oa.shape and op2.shape == oa.shape
    np.testing.assert_equal(op, op2)                                     # :337
    ctrl = o._topDownActivity(data, datap)                               # pinned by test_oracle
    np.testing.assert_allclose(oa, ctrl, rtol=1e-5, atol=1e-5)           # :389
    ctrl_p = np.exp(ctrl) / np.exp(ctrl).sum(axis=2, keepdims=True)
    np.testing.assert_allclose(ctrl_p, op, rtol=1e-5, atol=1e-5)         # :397
    if kind == "zeros":
        np.testing.assert_allclose(0.25, op, rtol=1e-5, atol=1e-5)       # :340, :363
    np.testing.assert_array_equal(sample.sum(axis=2), 1.0)
    u = visible_uniforms(model.seed, 2, np.arange(nseq), seqlen, KIND_API_V)
    ref = o._topDownSample(ctrl_p, u)
    bad = (sample != ref).any(axis=2)[:, 0]
    gap = np.min(np.abs(np.cumsum(ctrl_p[:, 0], axis=1) - u[:, None, :]), axis=1)
    assert np.all(gap[bad] < TIE), "visible sample differs away from a tie"
    assert bad.sum() <= 1


@pytest.mark.parametrize("ds", [False, True])
def test_topdown_full(ds):
    """tests/testcrbm.py:257-316."""
    data = synthetic_onehot(11, 200, seed=9)
    model, _ = make_pair(10, 5, ds=ds, bshift=4.0)
    _, h1 = model._computeHgivenV(data, False, rng_step=1)
    h2 = model._computeHgivenV(data, True, rng_step=1)[1] if ds else None
    assert h1.sum() > 0
    poutput, soutput = model._computeVgivenH(h1, h2, rng_step=1)
    assert poutput.shape == data.shape
    np.testing.assert_allclose(poutput.sum(), data.shape[0] * data.shape[3], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(soutput.sum(), data.shape[0] * data.shape[3], rtol=1e-4, atol=1e-4)


# ---- evaluation: tests/testcrbm.py:116-146, :507-519 -------------------------
@pytest.mark.parametrize("ds", [False, True])
def test_hitprobs_free_energy_pfms(ds, tmp_path):
    data = synthetic_onehot(100, 200, seed=10)
    model, o = make_pair(10, 15, ds=ds, bshift=3.0)
    pred = model.motifHitProbs(data)
    assert pred.shape == (100, 10, 1, 186)                               # :129
    np.testing.assert_allclose(pred, o.motifHitProbs(data), rtol=RTOL, atol=1e-7)
    fe = model.freeEnergy(data)
    assert fe.shape == (100,)                                            # :519
    np.testing.assert_allclose(fe, o.freeEnergy(data), rtol=RTOL)
    np.testing.assert_allclose(model.freeEnergy(data, permotif=True), o.freeEnergy(data, True), rtol=RTOL)
    pfms = model.getPFMs()
    assert len(pfms) == 10
    for i in range(10):
        np.testing.assert_allclose(pfms[i].sum(), 15)                    # :146
        np.testing.assert_allclose(pfms[i], o.getPFMs()[i], rtol=1e-5)
    np.testing.assert_allclose(model._evaluateParams(), o.evaluateParams(), rtol=1e-5)
    mfe, nmh = model._evaluateData(data)
    omfe, onmh = o.evaluateData(data, eval_step=0)
    np.testing.assert_allclose(mfe, omfe, rtol=RTOL)
    # nmh is the mean of a sample drawn with shared uniforms (convRBM.py:469-472): the count of ones may
    # differ from the oracle's only by units on a p == u tie
    Pd = o._computeHgivenV(data)[0]
    ud = hidden_uniforms(model.seed, 0, np.arange(100), 10, 186, 0, KIND_EVAL_H)
    ones_ref, near = int((Pd > ud).sum()), int((np.abs(Pd - ud) < TIE).sum())
    assert abs(onmh * Pd.size - ones_ref) < 0.5
    assert abs(int(round(float(nmh) * Pd.size)) - ones_ref) <= near
    # save / load round trip (tests/testcrbm.py:58-98)
    fn = str(tmp_path / "model.pkl")
    model.saveModel(fn)
    from crbm_amd import CRBM
    model2 = CRBM.loadModel(fn)
    for attr in ("num_motifs", "motif_length", "epochs", "input_dims", "doublestranded", "batchsize",
                 "momentum", "pooling", "cd_k", "rho", "lambda_rate"):
        assert getattr(model, attr) == getattr(model2, attr)
    np.testing.assert_allclose(model.motifs.get_value(), model2.motifs.get_value())
    np.testing.assert_allclose(model2.motifHitProbs(data[:5]), pred[:5], rtol=1e-6)


# ---- persistent chain (convRBM.py:397-408) -----------------------------------
@pytest.mark.parametrize("K,M,ds,Lf", [(10, 15, False, 186), (10, 15, True, 200), (20, 15, True, 120),
                                       (50, 25, False, 100), (3, 4, True, 17), (100, 15, False, 90), (20, 40, True, 70)])
def test_gibbs_chain_matches_oracle(K, M, ds, Lf):
    B = 24
    model, o = make_pair(K, M, ds=ds, batchsize=B, Lf=Lf, wscale=1.5, bshift=5.0)
    rng = np.random.default_rng(4)
    h0 = rng.binomial(1, 0.05, size=(B, K, 1, Lf)).astype(np.float32)
    hp0 = rng.binomial(1, 0.05, size=(B, K, 1, Lf)).astype(np.float32) if ds else None
    model.set_fantasy(h0, hp0)
    o.fantasy_h, o.fantasy_h_prime = h0.astype(np.float64), (hp0.astype(np.float64) if ds else None)
    a, b = model.get_fantasy()
    np.testing.assert_array_equal(a, h0)
    assert_chain_steps(model, o, 3)                   # sample for sample, ties only
    assert o.fantasy_h.sum() > 0
    # and the k-step launch composes: 1 + 2 steps in two launches == the oracle's 3 steps
    model.set_fantasy(h0, hp0)
    model.set_rng(gibbs_step=0)
    o.fantasy_h, o.fantasy_h_prime = h0.astype(np.float64), (hp0.astype(np.float64) if ds else None)
    o.gibbs_step = 0
    model.gibbsSteps(1)
    model.gibbsSteps(2)
    o.gibbs_steps(3)
    assert_chain_equal_or_tied(model, o, (h0, hp0, 0), 3)      # identical, or a tie in step 1 or 2 that fanned out


@pytest.mark.parametrize("variant", ["dense", "sparse"])
@pytest.mark.parametrize("ds", [False, True])
def test_gibbs_topdown_variants_match_oracle(variant, ds, monkeypatch):
    """Both top-down variants of the Gibbs kernel (dense tables / walk over set bits)
    reproduce the oracle chain; the activity monitor counts the final state."""
    import ctypes
    from crbm_amd._lib import CrbmLaunchInfo
    monkeypatch.setenv("CRBM_TOPDOWN", variant)
    B, K, M, Lf = 40, 10, 15, 186 if not ds else 60
    model, o = make_pair(K, M, ds=ds, batchsize=B, Lf=Lf, wscale=1.5, bshift=5.0)
    rng = np.random.default_rng(14)
    h0 = rng.binomial(1, 0.1, size=(B, K, 1, Lf)).astype(np.float32)
    hp0 = rng.binomial(1, 0.1, size=(B, K, 1, Lf)).astype(np.float32) if ds else None
    model.set_fantasy(h0, hp0)
    o.fantasy_h, o.fantasy_h_prime = h0.astype(np.float64), (hp0.astype(np.float64) if ds else None)
    assert_chain_steps(model, o, 3)                   # sample for sample, ties only
    h, hp = model.get_fantasy()
    info = CrbmLaunchInfo()
    model._lib.crbm_get_launch_info(model._h(), ctypes.byref(info))
    assert info.gibbs_sparse == (1 if variant == "sparse" else 0)
    total = h.sum() + (hp.sum() if ds else 0.0)
    assert abs(info.activity_ppm - 1e6 * total / (B * K * Lf * (2 if ds else 1))) <= 1.0


def test_gibbs_variant_is_fixed_per_handle():
    """The top-down variant never follows the measured activity: two ranks, or a
    1-GPU and a G-GPU run of the same chains, must round -- and therefore sample -- identically.  The
    activity monitor still reports what the chains do."""
    import ctypes
    from crbm_amd._lib import CrbmLaunchInfo
    B, K, M, Lf = 64, 10, 15, 100

    def variant_after(bshift):
        model, _ = make_pair(K, M, ds=False, batchsize=B, Lf=Lf, bshift=bshift)
        model.gibbsSteps(2)
        info = CrbmLaunchInfo()
        model._lib.crbm_get_launch_info(model._h(), ctypes.byref(info))
        return info.gibbs_sparse, info.activity_ppm

    sparse, ppm = variant_after(-1.0)      # b ~ -10: almost nothing on
    assert sparse == 1 and 0 <= ppm < 30000
    sparse, ppm = variant_after(9.0)       # b ~ 0: about half of the units on
    assert sparse == 1 and ppm > 100000


def test_gibbs_rejects_bad_state():
    model, _ = make_pair(4, 5, ds=False, batchsize=2, Lf=10)
    bad = np.full((2, 4, 1, 10), 0.5, dtype=np.float32)
    with pytest.raises(Exception, match="0/1"):
        model.set_fantasy(bad)
    with pytest.raises(Exception, match="one-hot"):
        model.motifHitProbs(np.zeros((2, 1, 4, 30), dtype=np.float32))


# ---- the PCD-k update (convRBM.py:373-438) -------------------------------------
@pytest.mark.parametrize("K,M,ds", [(10, 15, True), (10, 15, False), (4, 5, True), (20, 15, True), (50, 25, False),
                                    (100, 15, True), (20, 40, False)])
def test_train_step_trace(K, M, ds):
    B, Lf, n, L = 16, 40, 13, M + 57          # data batch != fantasy batch, lengths differ
    pair = lambda: make_pair(K, M, ds=ds, batchsize=B, Lf=Lf, cd_k=2, bshift=4.0, rho=0.02)
    model, o = pair()
    # three updates, each checked on W, b, c AND the velocities with the chains required identical (or, on a p == u tie,
    # replayed step by step: assert_train_steps)
    assert_train_steps(model, o, [synthetic_onehot(n, L, seed=100 + step) for step in range(3)], pair)


def test_host_reduced_data_parallel_equals_single():
    """a G-rank step on N rows == the 1-rank step on the same rows.
    Two handles on one GPU play two ranks; the host sums their packed buffers
    (what the RCCL all-reduce does) and both apply the same update."""
    import ctypes
    from crbm_amd._lib import fptr
    K, M, ds, B, Lf, n, L = 10, 15, True, 16, 30, 12, 60
    single, o = make_pair(K, M, ds=ds, batchsize=B, Lf=Lf, cd_k=1, bshift=4.0)
    D = synthetic_onehot(n, L, seed=31)
    single._trainingFct(D)
    ranks = []
    for r in range(2):
        m, _ = make_pair(K, M, ds=ds, batchsize=B, Lf=Lf, cd_k=1, bshift=4.0)
        m.rank, m.world_size = r, 2
        ranks.append(m)
    sums = []
    for r, m in enumerate(ranks):
        h = m._h()
        cnt = m._lib.crbm_sums_count(h)
        buf = np.zeros(cnt, dtype=np.float32)
        lo, hi = m._shard_rows(0, n)
        rows = np.ascontiguousarray(D[lo:hi])
        m._call("crbm_train_local", fptr(rows), hi - lo, L, fptr(buf))
        sums.append(buf)
    total = (sums[0].astype(np.float64) + sums[1]).astype(np.float32)
    for m in ranks:
        m._call("crbm_train_apply", fptr(total), L)
        np.testing.assert_allclose(m.motifs.get_value(), single.motifs.get_value(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(m.bias.get_value(), single.bias.get_value(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(m.c.get_value(), single.c.get_value(), rtol=1e-5, atol=1e-7)
    hs, _ = single.get_fantasy()
    h0, _ = ranks[0].get_fantasy()
    h1, _ = ranks[1].get_fantasy()
    np.testing.assert_array_equal(np.concatenate([h0, h1]), hs)          # identical samples


def test_rccl_single_rank_communicator():
    """The RCCL path with a 1-rank communicator must reproduce the plain step."""
    import ctypes
    from crbm_amd import _lib, dist
    a, _ = make_pair(10, 15, batchsize=8, Lf=30, cd_k=1, bshift=4.0)
    b, _ = make_pair(10, 15, batchsize=8, Lf=30, cd_k=1, bshift=4.0)
    uid = dist.exchange_unique_id(0, 1)
    buf = (ctypes.c_uint8 * _lib.UNIQUE_ID_BYTES).from_buffer_copy(uid)
    b._call("crbm_comm_init", buf, 1, 0)
    D = synthetic_onehot(8, 50, seed=2)
    a._trainingFct(D)
    b._trainingFct(D)
    np.testing.assert_array_equal(a.motifs.get_value(), b.motifs.get_value())
    b._call("crbm_comm_destroy")


# ---- fit(): tests/testcrbm.py:100-114 ------------------------------------------
@pytest.mark.parametrize("ds", [True, False])
def test_training_fit(ds, capsys):
    from crbm_amd import CRBM
    data = synthetic_onehot(100, 200, seed=12)
    model = CRBM(num_motifs=10, motif_length=15, doublestranded=ds, epochs=1, seed=1)
    w0 = model.motifs.get_value()
    bad = data.copy()
    bad[3, 0, :, 7] = 0.25
    with pytest.raises(Exception, match="one-hot"):
        model.fit(bad)
    model.fit(data)
    model.trainModel(data, data[:10])
    out = capsys.readouterr().out
    assert "BatchSize: 20" in out and "Epoch 0: FE=" in out and "Training finished after" in out
    w1 = model.motifs.get_value()
    assert np.isfinite(w1).all() and not np.array_equal(w0, w1)
    c = model.c.get_value()
    np.testing.assert_allclose(c[0], c[0, ::-1], atol=1e-6)             # symmetrised letter statistics


def test_fit_matches_oracle_one_epoch():
    """Config #1 plumbing shape at small scale: resident data set + batch loop."""
    K, M = 6, 9
    model, o = make_pair(K, M, ds=True, batchsize=20, cd_k=5, Lf=200, bshift=4.0, epochs=1)
    data = synthetic_onehot(50, 60, seed=13)       # 3 batches, last one short
    model.epochs = 1
    model.fit(data)
    for lo, hi in o._iterateBatchIndices(50, 20):
        o.train_step(data[lo:hi])
    np.testing.assert_allclose(model.motifs.get_value(), o.W, rtol=RTOL, atol=5e-6)
    np.testing.assert_allclose(model.bias.get_value(), o.b, rtol=RTOL, atol=5e-6)


def test_config1_fit_at_full_size(capsys):
    """BASELINE config #1 at its stated size: `CRBM(10, 15).fit` on 1000 synthetic 200-bp sequences with the
    reference's defaults (batchsize 20, cd_k 5, doublestranded, chains of hidden length 200; convRBM.py:68-71,
    :168, :570-634) -- 50 PCD-5 updates through the resident-data epoch loop -- against the oracle's 50
    `train_step`s: identical chains and 1e-4 on W, b, c, and the epoch's status line against the oracle's
    evaluation.  Should a sample have landed on a p == u tie, the epoch is replayed update by update
    (assert_train_steps: ties only)."""
    import re
    from crbm_amd import CRBM
    data = synthetic_onehot(1000, 200, seed=1234)
    W = np.random.default_rng(42).standard_normal((10, 1, 4, 15)).astype(np.float32)

    def pair():
        m = CRBM(num_motifs=10, motif_length=15, epochs=1, seed=2026)
        m.motifs.set_value(W)
        return m, OracleCRBM(10, 15, epochs=1, seed=2026, W=W)
    m, o = pair()
    assert (m.batchsize, m.cd_k, m.doublestranded, m.fantasy_hidden_len) == (20, 5, True, 200)
    m.fit(data)
    out = capsys.readouterr().out
    bounds = o._iterateBatchIndices(1000, 20)
    assert len(bounds) == 50
    for lo, hi in bounds:
        o.train_step(data[lo:hi])
    h, hp = m.get_fantasy()
    if np.array_equal(h, o.fantasy_h) and np.array_equal(hp, o.fantasy_h_prime):
        np.testing.assert_allclose(m.motifs.get_value(), o.W, rtol=RTOL, atol=5e-6)
        np.testing.assert_allclose(m.bias.get_value(), o.b, rtol=RTOL, atol=5e-6)
        np.testing.assert_allclose(m.c.get_value(), o.c, rtol=RTOL, atol=5e-6)
        # the status line (convRBM.py:616-630): batch means of evaluateData over the test set (= training set)
        ev = [o.evaluateData(data[lo:hi], eval_step=i) for i, (lo, hi) in enumerate(bounds)]
        line = [l for l in out.splitlines() if l.startswith("Epoch 0:")][0]
        got = dict(re.findall(r"(\w+)=\s*(-?[0-9.]+)", line))
        assert abs(float(got["FE"]) - np.mean([e[0] for e in ev])) < 2e-3
        assert abs(float(got["NumH"]) - np.mean([e[1] for e in ev])) < 2e-4
        twn, ic, medic = o.evaluateParams()
        assert abs(float(got["WNorm"]) - twn) < 1.1e-2 and abs(float(got["IC"]) - ic) < 2e-3 and abs(float(got["medIC"]) - medic) < 2e-3
    else:
        m2, o2 = pair()
        assert assert_train_steps(m2, o2, [data[lo:hi] for lo, hi in bounds], pair, atol=5e-6) >= 1


# ---- golden fixtures ---------------------------------------------------------------
def test_golden_fixtures(golden_dir):
    from crbm_amd import CRBM
    g = np.load(os.path.join(golden_dir, "crbm_golden.npz"))
    for tag in ("ss", "ds"):
        ds = tag == "ds"
        K, M = int(g[tag + "_K"]), int(g[tag + "_M"])
        m = CRBM(K, M, doublestranded=ds, batchsize=int(g[tag + "_B"]), cd_k=int(g[tag + "_cdk"]),
                 fantasy_hidden_len=int(g[tag + "_Lf"]), seed=int(g[tag + "_seed"]), rho=float(g[tag + "_rho"]))
        m.motifs.set_value(g[tag + "_W"]); m.bias.set_value(g[tag + "_b"]); m.c.set_value(g[tag + "_c"])
        D = g[tag + "_D"]
        np.testing.assert_allclose(m._bottomUpActivity(D), g[tag + "_act"], rtol=RTOL, atol=1e-5)
        np.testing.assert_allclose(m._bottomUpActivity(D, True), g[tag + "_act_rc"], rtol=RTOL, atol=1e-5)
        np.testing.assert_allclose(m.motifHitProbs(D), g[tag + "_hit"], rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(m.freeEnergy(D), g[tag + "_fe"], rtol=RTOL)
        np.testing.assert_allclose(m._topDownProbabilityOfHidden(g[tag + "_h"], g[tag + "_hp"] if ds else None),
                                   g[tag + "_pv"], rtol=RTOL, atol=1e-7)
        for step in range(int(g[tag + "_steps"])):
            m._trainingFct(D)
        h, _ = m.get_fantasy()
        if np.array_equal(h, g[tag + "_fh_after"]):
            np.testing.assert_allclose(m.motifs.get_value(), g[tag + "_W_after"], rtol=RTOL, atol=5e-6)
            np.testing.assert_allclose(m.bias.get_value(), g[tag + "_b_after"], rtol=RTOL, atol=5e-6)
            np.testing.assert_allclose(m.c.get_value(), g[tag + "_c_after"], rtol=RTOL, atol=5e-6)
            continue
        # the chain left the fixture's: legitimate only through a sample on a p == u tie -- replay the
        # fixture's training steps one by one beside the oracle that wrote it (tests/golden/make_golden.py)
        def pair():
            mm = CRBM(K, M, doublestranded=ds, batchsize=int(g[tag + "_B"]), cd_k=int(g[tag + "_cdk"]),
                      fantasy_hidden_len=int(g[tag + "_Lf"]), seed=int(g[tag + "_seed"]), rho=float(g[tag + "_rho"]))
            oo = OracleCRBM(K, M, doublestranded=ds, batchsize=int(g[tag + "_B"]), cd_k=int(g[tag + "_cdk"]),
                            fantasy_hidden_len=int(g[tag + "_Lf"]), seed=int(g[tag + "_seed"]), rho=float(g[tag + "_rho"]),
                            W=g[tag + "_W"])
            oo.b, oo.c = g[tag + "_b"].astype(np.float64), g[tag + "_c"].astype(np.float64)
            mm.motifs.set_value(g[tag + "_W"]); mm.bias.set_value(g[tag + "_b"]); mm.c.set_value(g[tag + "_c"])
            return mm, oo
        m3, o3 = pair()
        assert assert_train_steps(m3, o3, [D] * int(g[tag + "_steps"]), pair, atol=5e-6) >= 1
        np.testing.assert_allclose(o3.W, g[tag + "_W_after"], rtol=1e-6, atol=1e-6)      # the oracle still writes this fixture


# ---- full BASELINE sizes: size-independent properties (the oracle is too slow here) ----------
def _cfg2_model(chains, offset=0, seed=2026, **kw):
    from crbm_amd import CRBM
    m = CRBM(10, 15, doublestranded=False, batchsize=chains, cd_k=1, fantasy_hidden_len=186, seed=seed, **kw)
    m.motifs.set_value(np.random.default_rng(42).standard_normal((10, 1, 4, 15)).astype(np.float32))
    m._h()
    m._call("crbm_set_shard", offset)
    return m


def test_full_size_chain_properties():
    """Config #2 at 8192 chains: determinism, k-step composition, shard
    invariance (what 1 -> N GPUs relies on), one-hot visible samples, and
    agreement of the first chains with the oracle."""
    B = 8192
    a = _cfg2_model(B)
    a.gibbsSteps(3)
    ha, _ = a.get_fantasy()
    va = a.get_fantasy_visible()
    np.testing.assert_array_equal(va.sum(axis=2), 1.0)                  # one 1 per position
    assert 0.001 < ha.mean() < 0.2
    b = _cfg2_model(B)                                                   # same seed: identical
    b.gibbsSteps(1); b.gibbsSteps(2)                                     # 1 + 2 == 3
    hb, _ = b.get_fantasy()
    np.testing.assert_array_equal(ha, hb)
    lo = _cfg2_model(B // 2, 0)                                          # two "ranks"
    hi = _cfg2_model(B // 2, B // 2)
    lo.gibbsSteps(3); hi.gibbsSteps(3)
    np.testing.assert_array_equal(np.concatenate([lo.get_fantasy()[0], hi.get_fantasy()[0]]), ha)
    c = _cfg2_model(B, seed=7)                                           # another seed: different chain
    c.gibbsSteps(3)
    assert (c.get_fantasy()[0] != ha).mean() > 1e-3
    # the first 32 chains are bit-identical to a 32-chain run (chains are independent, counters are
    # keyed by the global chain index) and that run follows the oracle sample for sample (ties only)
    small = _cfg2_model(32)
    small.gibbsSteps(3)
    np.testing.assert_array_equal(small.get_fantasy()[0], ha[:32])
    o = OracleCRBM(10, 15, doublestranded=False, batchsize=32, cd_k=1, fantasy_hidden_len=186, seed=2026,
                   W=np.random.default_rng(42).standard_normal((10, 1, 4, 15)).astype(np.float32))
    twin = _cfg2_model(32)
    assert_chain_steps(twin, o, 3)


def test_full_size_geometry_independence(monkeypatch):
    """The chain must not depend on tile size / block size (launch geometry)."""
    B = 4096
    ref = _cfg2_model(B)
    ref.gibbsSteps(2)
    h0, _ = ref.get_fantasy()
    for S, T in ((3, 128), (16, 256), (5, 64)):
        monkeypatch.setenv("CRBM_GIBBS_S", str(S))
        monkeypatch.setenv("CRBM_GIBBS_THREADS", str(T))
        m = _cfg2_model(B)
        m.gibbsSteps(2)
        np.testing.assert_array_equal(m.get_fantasy()[0], h0)
    monkeypatch.delenv("CRBM_GIBBS_S"); monkeypatch.delenv("CRBM_GIBBS_THREADS")


def _unpack_sums(buf, K, M):
    """the packed raw-sum buffer of include/crbm_amd.h -> dict of arrays"""
    KAM = K * 4 * M
    row = 3 * KAM + 3 * K + 4
    d, m = buf[:row + 1], buf[row + 1:]
    out = {"vh_d": d[0:KAM], "vh_dp": d[KAM:2 * KAM], "h_d": d[2 * KAM:2 * KAM + K], "h_dp": d[2 * KAM + K:2 * KAM + 2 * K],
           "sw": d[2 * KAM + 2 * K:3 * KAM + 2 * K], "sb": d[3 * KAM + 2 * K:3 * KAM + 3 * K],
           "v_d": d[3 * KAM + 3 * K:3 * KAM + 3 * K + 4], "n_d": d[row],
           "vh_m": m[0:KAM], "vh_mp": m[KAM:2 * KAM], "h_m": m[2 * KAM:2 * KAM + K], "h_mp": m[2 * KAM + K:2 * KAM + 2 * K],
           "v_m": m[2 * KAM + 2 * K:2 * KAM + 2 * K + 4], "n_m": m[2 * KAM + 2 * K + 4]}
    return out


@pytest.mark.parametrize("K,M,ds,L,cd_k", [(50, 25, False, 1000, 5), (20, 15, True, 500, 1)])
def test_baseline_config_full_batch(K, M, ds, L, cd_k):
    """BASELINE configs #4 (50 x 25, 8192 x 4x1000, PCD-5) and #5 (20 x 15 doublestranded, 8192 x 4x500
    per GPU) at their FULL batch: the real cd_k chain of all 8192 chains, whose first chains are
    (a) bit-identical to an 8-chain run of the same handle type and (b) through that run checked
    against the oracle sample for sample (ties only); plus the packed raw sums of a 64-row
    sub-batch (both halves of the statistics, through crbm_train_local) against the oracle's."""
    import ctypes
    from crbm_amd import CRBM
    from crbm_amd._lib import fptr
    Lf, B = L - M + 1, 8192
    W = np.random.default_rng(42).standard_normal((K, 1, 4, M)).astype(np.float32)

    def make(chains):
        m = CRBM(K, M, doublestranded=ds, batchsize=chains, cd_k=cd_k, fantasy_hidden_len=Lf, seed=3)
        m.motifs.set_value(W)
        return m
    big = make(B)
    big.gibbsSteps(cd_k)
    hb, hbp = big.get_fantasy()
    assert hb.shape == (B, K, 1, Lf) and 0.0005 < hb.mean() < 0.2
    vb_ = big.get_fantasy_visible()
    np.testing.assert_array_equal(vb_.sum(axis=2), 1.0)
    small = make(8)
    small.gibbsSteps(cd_k)
    hs, hsp = small.get_fantasy()
    np.testing.assert_array_equal(hb[:8], hs)
    if ds:
        np.testing.assert_array_equal(hbp[:8], hsp)
    np.testing.assert_array_equal(vb_[:8], small.get_fantasy_visible())
    del hb, hbp, vb_, big
    # the 8-chain run against the oracle, step by step
    o = OracleCRBM(K, M, doublestranded=ds, batchsize=8, cd_k=cd_k, fantasy_hidden_len=Lf, seed=3, W=W)
    twin = make(8)
    assert_chain_steps(twin, o, cd_k)
    # raw sums of a training step on a 64-row sub-batch with 8 chains (data half + k Gibbs steps + model half)
    o2 = OracleCRBM(K, M, doublestranded=ds, batchsize=8, cd_k=cd_k, fantasy_hidden_len=Lf, seed=3, W=W)
    m2 = make(8)
    D = synthetic_onehot(64, L, seed=5)
    h2_ = m2._h()
    cnt = m2._lib.crbm_sums_count(h2_)
    buf = np.zeros(cnt, dtype=np.float32)
    m2._call("crbm_train_local", fptr(D), 64, L, fptr(buf))
    P_m, P_mp, v_m = o2.gibbs_steps(cd_k)
    h2, h2p = m2.get_fantasy()
    if np.array_equal(h2, o2.fantasy_h) and (not ds or np.array_equal(h2p, o2.fantasy_h_prime)):
        ref = o2.local_sums(D, P_m, P_mp, v_m)
        got = _unpack_sums(buf, K, M)
        for key in ("vh_d", "h_d", "sw", "sb", "v_d", "vh_m", "h_m", "v_m") + (("vh_dp", "h_dp", "vh_mp", "h_mp") if ds else ()):
            # (K,4,M) blocks: the fourth letter is derived as H - (the other three) when M is not a multiple of 16
            # (crbm_layout.h, NL), so its absolute error follows H = sum_a VH[k,a,j] -- 4e-7 of the largest H
            atol = 1e-5 + (1.6e-6 * float(np.abs(ref[key]).max()) if key.startswith(("vh", "sw")) else 0.0)
            np.testing.assert_allclose(got[key], np.ravel(ref[key]), rtol=RTOL, atol=atol, err_msg=key)
        assert got["n_d"] == 64 and got["n_m"] == 8
    else:                                   # a tie flipped a unit: replayed step by step, every difference must sit on a tie
        zeros = np.zeros_like(h2)
        assert_chain_equal_or_tied(m2, o2, (zeros, zeros if ds else None, 0), cd_k)


def test_resume_equals_uninterrupted_oracle(tmp_path):
    """SURVEY 8(f)-3 against the oracle, not against ourselves: the oracle runs two training steps
    in one go; the HIP model runs step -> saveState -> loadState -> step.  (The reference's own
    saveModel/loadModel would restart momentum and chains from zero, convRBM.py:226-235.)"""
    from crbm_amd import CRBM
    K, M, ds = 6, 9, True
    a, o = make_pair(K, M, ds=ds, batchsize=16, Lf=40, cd_k=2, bshift=4.0, rho=0.03)
    D1, D2 = synthetic_onehot(24, 70, seed=3), synthetic_onehot(24, 70, seed=4)
    twin = lambda: make_pair(K, M, ds=ds, batchsize=16, Lf=40, cd_k=2, bshift=4.0, rho=0.03)
    assert_train_steps(a, o, [D1], twin)               # identical chains (or a replayed tie), W, b, c, velocities at 1e-4
    fn = str(tmp_path / "state.pkl")
    a.saveState(fn)
    del a
    b = CRBM.loadState(fn)
    assert_train_steps(b, o, [D2], twin)               # the oracle was never interrupted: velocities and chains carried over
    # the reference-format file of the same model still loads, with momentum and chains reset
    fm = str(tmp_path / "model.pkl")
    b.saveModel(fm)
    c = CRBM.loadModel(fm)
    np.testing.assert_array_equal(c.motifs.get_value(), b.motifs.get_value())


def test_long_sequences_use_exact_division():
    """rows longer than 65536 positions (one sequence per tile) must not wrap the
    reciprocal-multiply division: activations, probabilities, samples, the dense v|h pass and the
    free energy of a 70 001-bp sequence against the oracle."""
    K, M, ds, L, n = 3, 7, True, 70001, 2
    model, o = make_pair(K, M, ds=ds, batchsize=2, Lf=10, bshift=5.0)
    D = synthetic_onehot(n, L, seed=9)
    np.testing.assert_allclose(model._bottomUpActivity(D), o._bottomUpActivity(D), rtol=1e-5, atol=2e-5)
    P, S = model._computeHgivenV(D, flip_motif=True, rng_step=3)
    Lh = L - M + 1
    u = hidden_uniforms(model.seed, 3, np.arange(n), K, Lh, 1, KIND_API_H)
    ref = o._bottomUpProbability(o._bottomUpActivity(D, True))
    np.testing.assert_allclose(P, ref, rtol=RTOL, atol=1e-7)
    assert_samples(S, ref, u)
    np.testing.assert_allclose(model.motifHitProbs(D), o.motifHitProbs(D), rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(model.freeEnergy(D), o.freeEnergy(D), rtol=RTOL)
    rng = np.random.default_rng(1)
    h = rng.binomial(1, 0.02, size=(n, K, 1, Lh)).astype(np.float32)
    hp = rng.binomial(1, 0.02, size=(n, K, 1, Lh)).astype(np.float32)
    np.testing.assert_allclose(model._topDownActivity(h, hp), o._topDownActivity(h, hp), rtol=1e-5, atol=2e-5)


def test_exact_path_queue_of_double_stranded_chains():
    """Double-stranded chain kernels queue the hidden units their coarse random field cannot decide
    (2^-12 of all units) per block and resolve them after the pass.  One chain per tile with
    2 x 64 x 2000 hidden units per step makes ~62 such units per block and step -- around the queue's
    capacity, so both the queue and its overflow path (resolved in place) run -- and every sample is
    compared with the oracle, ties only; then the same through a training step (fused or not)."""
    K, M, Lf, B = 64, 8, 2000, 3
    model, o = make_pair(K, M, ds=True, batchsize=B, Lf=Lf, cd_k=2, bshift=4.0, wscale=0.5)
    import ctypes
    from crbm_amd._lib import CrbmLaunchInfo
    info = CrbmLaunchInfo()
    hnd = model._h()                                   # creates the device handle (and model._lib)
    model._lib.crbm_get_launch_info(hnd, ctypes.byref(info))
    assert info.gibbs_seqs_per_tile <= 2, info.gibbs_seqs_per_tile
    ties = assert_chain_steps(model, o, 3)
    assert ties >= 0
    # the same with a queue of three entries: nearly every undecided unit takes the overflow path
    os.environ["CRBM_JIT_DEFINES"] = "-DCRBM_FIXQ_CAP=3"
    try:
        model1, o1 = make_pair(K, M, ds=True, batchsize=B, Lf=Lf, cd_k=2, bshift=4.0, wscale=0.5)
        assert_chain_steps(model1, o1, 2)
    finally:
        del os.environ["CRBM_JIT_DEFINES"]
    # a medium model whose tiles hold several chains (queue shared by the chains of a tile)
    model2, o2 = make_pair(20, 15, ds=True, batchsize=16, Lf=486, cd_k=2, bshift=4.0)
    assert_chain_steps(model2, o2, 3)
    D = synthetic_onehot(12, 500, seed=3)
    model2._trainingFct(D)
    o2.train_step(D)
    np.testing.assert_allclose(model2.motifs.get_value(), o2.W, rtol=RTOL, atol=2e-6)


def test_full_size_statistics_are_additive():
    """Config #2 at 8192 data rows: the packed raw sums of a batch equal the sum
    of the raw sums of its two halves (what the RCCL all-reduce relies on), and
    the letter counts / normalisers are exact."""
    from crbm_amd._lib import fptr
    n, L = 8192, 200
    D = synthetic_onehot(n, L, seed=77)

    def local(rows, offset, chains):
        m = _cfg2_model(chains, offset)
        cnt = m._lib.crbm_sums_count(m._handle)
        buf = np.zeros(cnt, dtype=np.float32)
        m._call("crbm_train_local", fptr(np.ascontiguousarray(rows)), rows.shape[0], L, fptr(buf))
        return buf

    whole = local(D, 0, 512)
    lo = local(D[:n // 2], 0, 256)
    hi = local(D[n // 2:], 256, 256)
    both = lo.astype(np.float64) + hi
    np.testing.assert_allclose(whole, both, rtol=2e-5, atol=1e-3)
    K, M = 10, 15
    KAM = K * 4 * M
    v_d = whole[3 * KAM + 3 * K:3 * KAM + 3 * K + 4]
    np.testing.assert_array_equal(v_d, D.sum(axis=(0, 1, 3)))           # exact letter counts
    assert whole[3 * KAM + 3 * K + 4] == n                               # n_d
    # sum_a VH[k,a,j] == sum_s P[k,s] for every j (one-hot columns), interior columns only differ by edges
    vh = whole[:KAM].reshape(K, 4, M)
    h = whole[2 * KAM:2 * KAM + K]
    np.testing.assert_allclose(vh.sum(axis=1), np.repeat(h[:, None], M, axis=1), rtol=1e-4)


def test_full_state_checkpoint_roundtrip(tmp_path):
    """velocities, chains and the sampler counter survive a
    save/load, so training resumes bit-identically (the reference's
    saveModel/loadModel resets them, convRBM.py:226-235)."""
    from crbm_amd import CRBM
    D = synthetic_onehot(64, 80, seed=3)
    a, _ = make_pair(6, 9, ds=True, batchsize=32, Lf=40, cd_k=2, bshift=4.0)
    a._trainingFct(D)
    fn = str(tmp_path / "state.pkl")
    a.saveState(fn)
    b = CRBM.loadState(fn)
    a._trainingFct(D)
    b._trainingFct(D)
    np.testing.assert_array_equal(a.motifs.get_value(), b.motifs.get_value())
    np.testing.assert_array_equal(a.get_fantasy()[0], b.get_fantasy()[0])
    np.testing.assert_array_equal(a.get_velocities()[0], b.get_velocities()[0])


def test_inference_streams_in_slabs(monkeypatch):
    """hit probabilities / free energy / evaluateData over inputs
    larger than the staging budget are processed slab by slab with identical
    results (rows keep their global sampler index)."""
    data = synthetic_onehot(300, 120, seed=21)
    a, _ = make_pair(10, 15, ds=True, bshift=4.0)
    ref_hit, ref_fe, ref_fem = a.motifHitProbs(data), a.freeEnergy(data), a.freeEnergy(data, True)
    ref_eval = a._evaluateData(data)
    monkeypatch.setenv("CRBM_SLAB_BYTES", str(64 * 1024))        # ~ 12 sequences per slab
    b, _ = make_pair(10, 15, ds=True, bshift=4.0)
    np.testing.assert_array_equal(b.motifHitProbs(data), ref_hit)
    np.testing.assert_array_equal(b.freeEnergy(data), ref_fe)
    np.testing.assert_array_equal(b.freeEnergy(data, True), ref_fem)
    got = b._evaluateData(data)
    np.testing.assert_allclose(got[0], ref_eval[0], rtol=1e-6)
    assert got[1] == ref_eval[1]


@pytest.mark.parametrize("K,M,ds,L,n", [(1, 1, False, 7, 3), (1, 1, True, 1, 2), (5, 8, True, 8, 4),
                                        (64, 32, True, 45, 3), (33, 17, False, 300, 5), (16, 16, True, 64, 6),
                                        # beyond 64 motifs (masks of 3..8 words) and beyond 32-letter motifs (two-word
                                        # letter windows): the reference takes any positive K, M (convRBM.py:72-108)
                                        (100, 15, False, 120, 4), (20, 40, True, 150, 4), (70, 33, True, 90, 3),
                                        (130, 7, True, 60, 3), (192, 4, False, 40, 2), (256, 4, False, 40, 2), (4, 64, True, 100, 3),
                                        # beyond the LDS-resident kernels (more than 256 motifs, more than 64 letters, tables
                                        # that exceed the LDS): the generic kernels -- round 3 REFUSED these
                                        (300, 10, False, 60, 3), (120, 40, True, 100, 3), (8, 100, False, 160, 3), (8, 100, True, 130, 2),
                                        (256, 4, True, 40, 2), (1, 65, True, 65, 2), (257, 1, False, 9, 2)])
def test_edge_shapes(K, M, ds, L, n):
    """Smallest and largest supported models, L == M (a single hidden position),
    mask-word and letter-window boundaries: activations, hit probabilities,
    free energy, a Gibbs chain and a training step against the oracle."""
    check_model_against_oracle(K, M, ds, L, n)


def check_model_against_oracle(K, M, ds, L, n, **kw):
    Lf = max(1, L - M + 1)
    model, o = make_pair(K, M, ds=ds, batchsize=4, Lf=Lf, cd_k=2, bshift=3.0, wscale=0.7, **kw)
    D = synthetic_onehot(n, L, seed=K + M, A=kw.get("input_dims", 4))
    np.testing.assert_allclose(model._bottomUpActivity(D), o._bottomUpActivity(D), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(model._bottomUpActivity(D, True), o._bottomUpActivity(D, True), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(model.motifHitProbs(D), o.motifHitProbs(D), rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(model.freeEnergy(D), o.freeEnergy(D), rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(model.freeEnergy(D, True), o.freeEnergy(D, True), rtol=RTOL, atol=2e-5)
    P = o.motifHitProbs(D)
    s = model.motifHitSummary(D)
    np.testing.assert_allclose(s["max"], P.max(axis=(2, 3)), rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(s["mean"], P.mean(axis=(2, 3)), rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(s["position_mean"], P.mean(axis=(0, 2)), rtol=RTOL, atol=1e-7)
    assert_chain_steps(model, o, 2)                    # sample for sample, ties only
    model.set_fantasy(o.fantasy_h.astype(np.float32), o.fantasy_h_prime.astype(np.float32) if ds else None)
    twin = lambda: make_pair(K, M, ds=ds, batchsize=4, Lf=Lf, cd_k=2, bshift=3.0, wscale=0.7, **kw)
    assert_train_steps(model, o, [D, D], twin, atol=5e-6)
    return model


@pytest.mark.parametrize("A,K,M,ds,L,n,pool", [(3, 1, 1, True, 30, 5, 1), (3, 6, 5, True, 60, 7, 1), (20, 12, 9, False, 80, 6, 1),
                                               (5, 7, 6, True, 47, 5, 2), (1, 3, 4, False, 40, 4, 1), (25, 40, 12, False, 70, 5, 1),
                                               (64, 5, 8, True, 50, 4, 1)])
def test_other_alphabets(A, K, M, ds, L, n, pool, capsys):
    """input_dims != 4: the reference warns ("not comprehensively tested") and runs (convRBM.py:84-87; its own constructor
    test builds a 3-letter model, tests/testcrbm.py:30-31).  Here such models run on the generic kernels with one byte per
    letter: activations on both strands (rc(W) = W[:, ::-1, ::-1] over A letters), hit probabilities and summaries, free
    energies, the chain sample for sample (ties only), PCD steps, fit() with its status line, the byte-code input form,
    save/load -- each against the oracle.  20 and 25 letters: protein alphabets; 64: the largest the v|h kernel takes."""
    import warnings
    from crbm_amd import CRBM
    kw = dict(input_dims=A)
    if pool > 1:
        kw["pooling"] = pool
    model = check_model_against_oracle(K, M, ds, L, n, **kw)
    o2 = OracleCRBM(K, M, doublestranded=ds, input_dims=A, W=model.motifs.get_value(), **({"pooling": pool} if pool > 1 else {}))
    o2.b, o2.c = model.bias.get_value().astype(np.float64), model.c.get_value().astype(np.float64)
    np.testing.assert_allclose(model._evaluateParams(), o2.evaluateParams(), rtol=1e-4, atol=1e-6)
    pf = model.getPFMs()
    assert pf[0].shape == (A, M)
    np.testing.assert_allclose(np.sum(pf[0], axis=0), 1.0, rtol=1e-12)
    # letter codes (one byte per letter) are the same input as the one-hot array
    D = synthetic_onehot(9, L, seed=77, A=A)
    codes = np.ascontiguousarray(D[:, 0].argmax(axis=1).astype(np.uint8))
    np.testing.assert_array_equal(model.freeEnergy(codes), model.freeEnergy(D))
    np.testing.assert_array_equal(model.motifHitProbs(codes), model.motifHitProbs(D))
    if A < 255:
        bad = codes.copy()
        bad[0, 0] = A
        with pytest.raises(Exception, match="one-hot"):
            model.freeEnergy(bad)
    with pytest.raises(Exception, match="shape"):
        model.freeEnergy(synthetic_onehot(3, L, seed=1, A=A + 1))
    model.epochs = 2
    model.fit(D, D[:4])
    out = capsys.readouterr().out
    assert "Epoch 1: FE=" in out and np.isfinite(model.motifs.get_value()).all()
    # save / load keeps the alphabet (convRBM.py:177-236)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "m.pkl")
        model.saveModel(path)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", UserWarning)
            again = CRBM.loadModel(path)
    assert again.input_dims == A
    np.testing.assert_array_equal(again.motifs.get_value(), model.motifs.get_value())
    np.testing.assert_array_equal(again.freeEnergy(D), model.freeEnergy(D))
    # the full training state resumes exactly: step -> saveState -> loadState -> step equals two steps
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "s.state")
        model._trainingFct(D)
        model.saveState(path)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", UserWarning)
            resumed = CRBM.loadState(path)
        model._trainingFct(D)
        resumed._trainingFct(D)
    np.testing.assert_array_equal(resumed.motifs.get_value(), model.motifs.get_value())
    np.testing.assert_array_equal(resumed.c.get_value(), model.c.get_value())
    for x, y in zip(resumed.get_fantasy(), model.get_fantasy()):
        if x is not None:
            np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(resumed.get_fantasy_visible(), model.get_fantasy_visible())


def test_generic_kernels_on_a_model_the_specialised_ones_take(monkeypatch, capsys):
    """CRBM_FORCE_BIG=1 puts the reference's default model (10 x 15, both strands) on the generic kernels: the same
    checks against the oracle, fit() with its status line, a checkpoint that resumes on the specialised kernels --
    and the chain of the generic path IS the chain of the specialised path (same counters, same rules)."""
    from crbm_amd import CRBM, _lib
    monkeypatch.setenv("CRBM_FORCE_BIG", "1")
    big = check_model_against_oracle(10, 15, True, 90, 5)
    info = _lib.CrbmLaunchInfo()
    big._lib.crbm_get_launch_info(big._h(), ctypes.byref(info))
    assert info.gibbs_grid == 0 and info.chain_parts == 1          # the generic path has no specialised geometry
    D = synthetic_onehot(37, 90, seed=3)
    big.epochs = 2
    big.fit(D)
    assert "Epoch 1: FE=" in capsys.readouterr().out
    # the two paths sample the same chain
    monkeypatch.delenv("CRBM_FORCE_BIG")
    fast, o = make_pair(10, 15, ds=True, batchsize=4, Lf=76, cd_k=2, bshift=3.0, wscale=0.7)
    monkeypatch.setenv("CRBM_FORCE_BIG", "1")
    slow, _ = make_pair(10, 15, ds=True, batchsize=4, Lf=76, cd_k=2, bshift=3.0, wscale=0.7)
    fast.gibbsSteps(3)
    slow.gibbsSteps(3)
    for a, b in zip(fast.get_fantasy(), slow.get_fantasy()):
        assert (a != b).mean() < 1e-3                              # identical up to p == u ties (float rounding of the two paths)
    assert fast.get_fantasy()[0].sum() > 0


@pytest.mark.parametrize("force_big", [False, True])
def test_epoch_evaluation_in_one_call_is_the_loop_over_batches(force_big, monkeypatch):
    """fit()'s per-epoch evaluation (convRBM.py:616-625) runs as ONE library call (crbm_eval_epoch_resident): the same
    numbers, bit for bit, as the loop over crbm_eval_data_resident it replaces -- same sampler step per batch, a short last
    batch weighing as much as a full one -- and the sampler has moved on by one step per batch."""
    if force_big:
        monkeypatch.setenv("CRBM_FORCE_BIG", "1")
    model, _ = make_pair(10, 15, ds=True, batchsize=20, Lf=60, bshift=4.0)
    D = synthetic_onehot(130, 74, seed=8)                     # 7 mini-batches, the last one of 10 rows
    model._upload(D, 0)
    model._call("crbm_dataset_select", 0)
    model.set_rng(gibbs_step=0, eval_step=5)
    mfe, nmh = ctypes.c_float(), ctypes.c_float()
    sfe = snmh = 0.0
    nb = 0
    for start in range(0, 130, 20):
        model._call("crbm_eval_data_resident", start, min(start + 20, 130), ctypes.byref(mfe), ctypes.byref(nmh))
        sfe, snmh, nb = sfe + mfe.value, snmh + nmh.value, nb + 1
    assert model.get_rng()[2] == 5 + nb
    model.set_rng(gibbs_step=0, eval_step=5)
    a, b = ctypes.c_double(), ctypes.c_double()
    model._call("crbm_eval_epoch_resident", 20, ctypes.byref(a), ctypes.byref(b))
    assert a.value == sfe / nb and b.value == snmh / nb
    assert b.value > 0 and model.get_rng()[2] == 5 + nb


# ---- pooling > 1 (convRBM.py:245-267, :586-599, :664-665) ---------------------------------
@pytest.mark.parametrize("K,M,ds,pool", [(6, 7, True, 2), (10, 15, False, 4), (5, 8, True, 3)])
def test_pooling(K, M, ds, pool, capsys):
    """Pooled hidden units: probabilities exp(x)/(pool + sum exp) over groups of `pool` positions, one
    multinomial draw per group, free energy log(1 + sum exp), the PCD update with the pooled sparsity
    slope, and fit()'s truncation -- each against the oracle (which is pinned for pooling by
    tests/test_oracle.py: probability formula and finite differences of the penalty)."""
    check_pooling(K, M, ds, pool, capsys)


@pytest.mark.parametrize("K,M,ds,pool,force", [(6, 7, True, 2, True), (10, 15, False, 4, True), (70, 70, True, 3, False), (300, 6, False, 2, False)])
def test_pooling_on_the_generic_kernels(K, M, ds, pool, force, capsys, monkeypatch):
    """The same checks on the generic kernels: two models the specialised kernels take (CRBM_FORCE_BIG=1) and two they do
    not (70-letter motifs, 300 motifs) -- round 3 and the first half of round 4 refused pooled models of that size."""
    if force:
        monkeypatch.setenv("CRBM_FORCE_BIG", "1")
    check_pooling(K, M, ds, pool, capsys)


def check_pooling(K, M, ds, pool, capsys):
    from crbm_amd import CRBM
    Lf, B, n = 12 * pool, 8, 9
    L = 10 * pool + M - 1
    rng = np.random.default_rng(50 + K)
    W = (rng.standard_normal((K, 1, 4, M)) * 0.8).astype(np.float32)
    def twin():
        m_ = CRBM(K, M, doublestranded=ds, batchsize=B, cd_k=2, pooling=pool, fantasy_hidden_len=Lf, seed=11, rho=0.03)
        o_ = OracleCRBM(K, M, doublestranded=ds, batchsize=B, cd_k=2, pooling=pool, fantasy_hidden_len=Lf, seed=11, rho=0.03, W=W)
        b_ = (o_.b + 4.0).astype(np.float32)
        m_.motifs.set_value(W)
        m_.bias.set_value(b_)
        o_.b = b_.astype(np.float64)
        return m_, o_
    m, o = twin()
    D = synthetic_onehot(n, L, seed=4)
    # bottom-up: probability and sample (one draw per group)
    P, S = m._computeHgivenV(D, rng_step=6)
    Pref = o._bottomUpProbability(o._bottomUpActivity(D))
    np.testing.assert_allclose(P, Pref, rtol=RTOL, atol=1e-7)
    Lh = L - M + 1
    u = hidden_uniforms(m.seed, 6, np.arange(n), K, Lh, 0, KIND_API_H)
    Sref = o._bottomUpSample(Pref, u)
    grp = S.reshape(n, K, 1, Lh // pool, pool)
    assert np.all(grp.sum(axis=4) <= 1) and S.sum() > 0
    bad = (S != Sref).reshape(n, K, 1, Lh // pool, pool).any(axis=4)
    if bad.any():
        cum = np.cumsum(Pref.reshape(n, K, 1, Lh // pool, pool), axis=4)
        ug = u.reshape(n, K, 1, Lh // pool, pool)[..., :1]
        assert np.all(np.min(np.abs(cum - ug), axis=4)[bad] < TIE)
    # evaluation
    np.testing.assert_allclose(m.motifHitProbs(D), o.motifHitProbs(D), rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(m.freeEnergy(D), o.freeEnergy(D), rtol=RTOL)
    np.testing.assert_allclose(m.freeEnergy(D, True), o.freeEnergy(D, True), rtol=RTOL, atol=1e-5)
    with pytest.raises(Exception, match="pooling"):
        m.freeEnergy(synthetic_onehot(2, L + 1, seed=1))          # hidden length not a multiple of pooling
    # the chain and a training step
    before = (m.get_fantasy()[0], m.get_fantasy()[1], o.gibbs_step)
    m.gibbsSteps(2)
    o.gibbs_steps(2)
    assert_chain_equal_or_tied(m, o, before, 2)        # identical, or replayed step by step onto a tie of a pooling group
    h, hp = m.get_fantasy()
    assert np.all(h.reshape(B, K, 1, Lf // pool, pool).sum(axis=4) <= 1)
    assert_train_steps(m, o, [D], twin)
    # fit() truncates the sequences so that the hidden length divides (convRBM.py:586-599)
    m.epochs = 1
    m.fit(synthetic_onehot(20, L + 1, seed=8))
    assert "Epoch 0: FE=" in capsys.readouterr().out


@pytest.mark.parametrize("K,M,ds,pool", [(300, 10, False, 1), (120, 40, True, 1), (257, 1, False, 1), (150, 6, True, 2), (70, 33, True, 1)])
def test_slabbed_statistics_of_generic_models(K, M, ds, pool, monkeypatch):
    """Generic DNA models (motif_length <= 64) take their statistics from the specialised MFMA kernel, a slab of at most 64
    motifs at a time (crbm_api.hip, slab_launch_stats): the packed raw sums of a training step -- both halves, through
    crbm_train_local -- equal those of the plain generic statistics kernel (CRBM_SLAB_STATS=0) within float rounding and the
    oracle's within the tolerance of every other statistics test.  257 = 4 x 64 + 1: the last slab is moved back to end at K.
    (70 x 33 double-stranded runs on the specialised kernels anyway: CRBM_FORCE_BIG puts it on one slab of 64 and one of 6.)"""
    import ctypes
    from crbm_amd import CRBM
    from crbm_amd._lib import fptr
    if K == 70:
        monkeypatch.setenv("CRBM_FORCE_BIG", "1")
    Lf, B, n = 12 * pool, 6, 10
    L = Lf + M - 1
    W = (np.random.default_rng(K + M).standard_normal((K, 1, 4, M)) * 0.6).astype(np.float32)
    D = synthetic_onehot(n, L, seed=17)

    def run(slabs):
        monkeypatch.setenv("CRBM_SLAB_STATS", "1" if slabs else "0")
        m = CRBM(K, M, doublestranded=ds, batchsize=B, cd_k=2, pooling=pool, fantasy_hidden_len=Lf, seed=9, rho=0.02)
        m.motifs.set_value(W)
        m.bias.set_value((m.bias.get_value() + 3.0).astype(np.float32))
        h_ = m._h()
        buf = np.zeros(m._lib.crbm_sums_count(h_), dtype=np.float32)
        m._call("crbm_train_local", fptr(D), n, L, fptr(buf))
        return m, buf
    m1, with_slabs = run(True)
    m0, plain = run(False)
    for a, b in zip(m1.get_fantasy(), m0.get_fantasy()):
        if a is not None:
            np.testing.assert_array_equal(a, b)                 # the chain does not depend on the statistics path
    # free energies (convRBM.py:657-697) slab by slab: per sequence and per motif against the generic kernel and the oracle
    np.testing.assert_allclose(m1.freeEnergy(D), m0.freeEnergy(D), rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(m1.freeEnergy(D, True), m0.freeEnergy(D, True), rtol=RTOL, atol=2e-5)
    # evaluateData (convRBM.py:517-522): the sampled activity is counted by the slabbed h|v too
    e1, e0 = m1._evaluateData(D), m0._evaluateData(D)
    assert abs(e1[0] - e0[0]) <= 1e-5 * abs(e0[0]) and abs(e1[1] - e0[1]) <= 2.0 / (n * K * (L - M + 1)) and e1[1] > 0
    got, ref = _unpack_sums(with_slabs, K, M), _unpack_sums(plain, K, M)
    for key in ref:
        scale = float(np.abs(ref[key]).max()) if np.size(ref[key]) else 0.0
        np.testing.assert_allclose(got[key], ref[key], rtol=RTOL, atol=1e-5 + 2e-6 * scale, err_msg=key)
    assert np.abs(got["vh_d"]).max() > 0 and np.abs(got["sw"]).max() > 0 and got["n_d"] == n and got["n_m"] == B
    # and the oracle, from the chain the handle sampled (ties aside, it is the oracle's own)
    o = OracleCRBM(K, M, doublestranded=ds, batchsize=B, cd_k=2, pooling=pool, fantasy_hidden_len=Lf, seed=9, rho=0.02, W=W)
    o.b = m1.bias.get_value().astype(np.float64)
    P_m, P_mp, v_m = o.gibbs_steps(2)
    h1, h1p = m1.get_fantasy()
    np.testing.assert_allclose(m1.freeEnergy(D), o.freeEnergy(D), rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(m1.freeEnergy(D, True), o.freeEnergy(D, True), rtol=RTOL, atol=2e-5)
    if np.array_equal(h1, o.fantasy_h) and (not ds or np.array_equal(h1p, o.fantasy_h_prime)):
        want = o.local_sums(D, P_m, P_mp, v_m)
        for key in ("vh_d", "h_d", "sw", "sb", "v_d", "vh_m", "h_m", "v_m") + (("vh_dp", "h_dp", "vh_mp", "h_mp") if ds else ()):
            atol = 1e-5 + (1.6e-6 * float(np.abs(want[key]).max()) if key.startswith(("vh", "sw")) else 0.0)
            np.testing.assert_allclose(got[key], np.ravel(want[key]), rtol=RTOL, atol=atol, err_msg=key)


def test_models_beyond_the_lds_train_on_the_generic_kernels(capsys):
    """Round 3 refused these (tables of both strands plus one chain exceed the LDS; 16 column roles of the statistics
    block): now fit() runs them -- against the oracle in test_edge_shapes, here through the reference's entry point."""
    from crbm_amd import CRBM
    for K, M in ((120, 40), (256, 4)):
        np.random.seed(K)
        m = CRBM(K, M, epochs=1, doublestranded=True, batchsize=4, cd_k=2, fantasy_hidden_len=30, seed=5)
        m.fit(synthetic_onehot(9, M + 29, seed=1))
        assert "Epoch 0: FE=" in capsys.readouterr().out
        assert np.isfinite(m.motifs.get_value()).all() and len(m.getPFMs()) == K
    with pytest.raises(Exception, match="65536"):
        CRBM(70000, 4)
    with pytest.raises(Exception, match="512"):
        CRBM(4, 600)


def test_pooling_must_divide_the_chain_length():
    from crbm_amd import CRBM
    m = CRBM(4, 5, pooling=3)                     # the reference's chains are 200 long: 3 does not divide
    with pytest.raises(Exception, match="pooling"):
        m.gibbsSteps(1)


# ---- packed input and data-set scale sweeps ------------------------
@pytest.mark.parametrize("ds", [False, True])
def test_codes_input_equals_onehot(ds):
    """(n,L) uint8 letter codes give bit-identical results to the fp32 one-hot array."""
    from crbm_amd.sequences import codesToOneHot
    rng = np.random.default_rng(31)
    codes = rng.integers(0, 4, size=(37, 150), dtype=np.uint8)
    data = codesToOneHot(codes)
    m, o = make_pair(10, 15, ds=ds, bshift=4.0)
    np.testing.assert_array_equal(m.motifHitProbs(codes), m.motifHitProbs(data))
    np.testing.assert_array_equal(m.freeEnergy(codes), m.freeEnergy(data))
    np.testing.assert_array_equal(m.freeEnergy(codes, True), m.freeEnergy(data, True))
    np.testing.assert_allclose(m.freeEnergy(codes), o.freeEnergy(data), rtol=RTOL, atol=1e-6)
    bad = codes.copy()
    bad[5, 9] = 4
    with pytest.raises(Exception, match="one-hot"):
        m.freeEnergy(bad)


@pytest.mark.parametrize("K,M,ds,L", [(10, 15, False, 200), (10, 15, True, 200), (20, 15, True, 1200), (3, 4, False, 50)])
def test_hit_summary_matches_oracle(K, M, ds, L):
    """utils.py:113-116, :154, :305 reductions of motifHitProbs, fused on the device
    (L=1200 with K=20 needs several position chunks: the atomic combine path)."""
    n = 45
    data = synthetic_onehot(n, L, seed=41)
    m, o = make_pair(K, M, ds=ds, bshift=4.0)
    P = o.motifHitProbs(data)
    got = m.motifHitSummary(data)
    np.testing.assert_allclose(got["max"], P.max(axis=(2, 3)), rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(got["mean"], P.mean(axis=(2, 3)), rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(got["position_mean"], P.mean(axis=(0, 2)), rtol=RTOL, atol=1e-7)
    dense = m.motifHitProbs(data)
    np.testing.assert_allclose(got["max"], dense.max(axis=(2, 3)), rtol=1e-6)
    only = m.motifHitSummary(data, position_mean=False)
    np.testing.assert_array_equal(only["max"], got["max"])
    assert "position_mean" not in only


def test_hit_summary_is_the_same_bits_in_every_run(monkeypatch):
    """The position sums (combined across waves, blocks, slabs and the two streams of a host-input sweep) and the
    means over several position chunks are accumulated in 64-bit fixed point: repeated sweeps over the same data give
    IDENTICAL arrays, whatever order the blocks and streams finish in."""
    monkeypatch.setenv("CRBM_SLAB_BYTES", str(256 * 1024))         # many slabs, alternating streams
    rng = np.random.default_rng(8)
    for K, M, ds, n, L in ((10, 15, False, 700, 200), (20, 15, True, 300, 1200)):   # one chunk of positions / several
        m, o = make_pair(K, M, ds=ds, bshift=4.0)
        codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
        first = m.motifHitSummary(codes)
        for _ in range(3):
            again = m.motifHitSummary(codes)
            for key in ("max", "mean", "position_mean"):
                np.testing.assert_array_equal(first[key], again[key], err_msg=key)
        from crbm_amd.sequences import codesToOneHot
        P = o.motifHitProbs(codesToOneHot(codes[:64]))
        np.testing.assert_allclose(m.motifHitSummary(codes[:64])["mean"], P.mean(axis=(2, 3)), rtol=RTOL, atol=1e-7)


def test_resident_sweeps_and_slots(monkeypatch):
    """Rows of a resident data set give the same numbers as host input; two slots coexist."""
    import ctypes
    from crbm_amd._lib import fptr
    rng = np.random.default_rng(33)
    ca = rng.integers(0, 4, size=(64, 90), dtype=np.uint8)
    cb = rng.integers(0, 4, size=(40, 70), dtype=np.uint8)
    monkeypatch.setenv("CRBM_SLAB_BYTES", str(32 * 1024))
    m, _ = make_pair(10, 15, ds=True, bshift=4.0)
    m._upload(ca, 0)
    m._upload(cb, 1)
    K = 10
    for slot, c, lo, hi in ((0, ca, 5, 61), (1, cb, 0, 40)):
        m._call("crbm_dataset_select", slot)
        n, Lh = hi - lo, c.shape[1] - 15 + 1
        fe = np.empty(n, np.float32); fem = np.empty((n, K), np.float32)
        m._call("crbm_free_energy_resident", lo, hi, fptr(fe), fptr(fem))
        np.testing.assert_array_equal(fe, m.freeEnergy(c[lo:hi]))
        np.testing.assert_array_equal(fem, m.freeEnergy(c[lo:hi], True))
        hp = np.empty((n, K, 1, Lh), np.float32)
        m._call("crbm_hit_probs_resident", lo, hi, fptr(hp))
        np.testing.assert_array_equal(hp, m.motifHitProbs(c[lo:hi]))
        mx = np.empty((n, K), np.float32); mean = np.empty((n, K), np.float32); pos = np.empty((K, Lh), np.float32)
        m._call("crbm_hit_summary_resident", lo, hi, fptr(mx), fptr(mean), fptr(pos))
        s = m.motifHitSummary(c[lo:hi])
        np.testing.assert_array_equal(mx, s["max"])
        np.testing.assert_allclose(pos, s["position_mean"], rtol=1e-5)   # (another split into slabs: the waves' float partial sums group differently)
    with pytest.raises(Exception, match="slot"):
        m._call("crbm_dataset_select", 2)
    m._call("crbm_dataset_select", 0)
    with pytest.raises(Exception, match="out of bounds"):
        m._call("crbm_free_energy_resident", 0, 65, fptr(np.empty(65, np.float32)), None)


def test_fit_with_codes_and_test_set(capsys):
    """fit() on letter codes == fit() on the one-hot array, with a separate test set."""
    from crbm_amd.sequences import codesToOneHot
    rng = np.random.default_rng(35)
    train = rng.integers(0, 4, size=(50, 60), dtype=np.uint8)
    test = rng.integers(0, 4, size=(30, 60), dtype=np.uint8)
    a, _ = make_pair(6, 9, ds=True, batchsize=20, cd_k=2, bshift=4.0, epochs=2)
    b, _ = make_pair(6, 9, ds=True, batchsize=20, cd_k=2, bshift=4.0, epochs=2)
    a.fit(train, test)
    out_a = capsys.readouterr().out
    b.fit(codesToOneHot(train), codesToOneHot(test))
    out_b = capsys.readouterr().out
    np.testing.assert_array_equal(a.motifs.get_value(), b.motifs.get_value())
    status = lambda s: [l for l in s.splitlines() if l.startswith("Epoch")]
    assert status(out_a) == status(out_b) and len(status(out_a)) == 2
    # the status line is the batch mean of _evaluateData over the test set (convRBM.py:617-625)
    fe = np.mean([a._evaluateData(codesToOneHot(test[lo:hi]))[0] for lo, hi in a._iterateBatchIndices(30, 20)])
    assert "FE={:1.3f}".format(fe) in status(out_a)[-1]


def test_epoch_entry_equals_step_loop():
    """crbm_train_epoch_resident (one sync per epoch) == the per-batch loop of convRBM.py:612-615."""
    rng = np.random.default_rng(37)
    codes = rng.integers(0, 4, size=(70, 64), dtype=np.uint8)       # 4 batches of 20, last one short
    a, _ = make_pair(6, 9, ds=True, batchsize=20, cd_k=2, bshift=4.0)
    b, _ = make_pair(6, 9, ds=True, batchsize=20, cd_k=2, bshift=4.0)
    a._upload(codes, 0)
    b._upload(codes, 0)
    for lo, hi in a._iterateBatchIndices(70, 20):
        a._call("crbm_train_step_resident", lo, hi)
    b._call("crbm_train_epoch_resident", 20)
    np.testing.assert_array_equal(a.motifs.get_value(), b.motifs.get_value())
    np.testing.assert_array_equal(a.get_fantasy()[0], b.get_fantasy()[0])
