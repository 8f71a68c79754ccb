"""Pins the CPU oracle (oracle/crbm_oracle.py).

The reference cannot be imported here (Theano absent), so the oracle is pinned
by restating the *control implementations* the reference's own tests use
(reference tests/testcrbm.py) and by independent SciPy primitives.  Every test
names the reference lines it restates.
"""
import numpy as np
import pytest
import scipy.signal

from oracle.crbm_oracle import (OracleCRBM, philox4x32, hidden_uniforms,
                                visible_uniforms, synthetic_onehot, letters_of)


def sigmoid(act):                      # reference tests/testcrbm.py:8-9
    return 1. / (1. + np.exp(-act))


# --------------------------------------------------------------------------
# Philox-4x32 known-answer vectors (Random123 kat_vectors, published) for the seven rounds the
# kernels, the oracle and the C port draw, and for the ten-round variant of the same routine
# --------------------------------------------------------------------------
@pytest.mark.parametrize("rounds,ctr,key,expect", [
    (7, (0, 0, 0, 0), (0, 0),
     (0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48)),
    (7, (0xffffffff,) * 4, (0xffffffff,) * 2,
     (0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662)),
    (7, (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a)),
    (10, (0, 0, 0, 0), (0, 0),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    (10, (0xffffffff,) * 4, (0xffffffff,) * 2,
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    (10, (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
])
def test_philox_known_answers(rounds, ctr, key, expect):
    out = philox4x32(*ctr, *key, rounds=rounds)
    assert tuple(int(x) for x in out) == expect
    if rounds == 7:                                   # ... and seven is what everything draws by default
        assert tuple(int(x) for x in philox4x32(*ctr, *key)) == expect


def test_uniform_layout():
    u = hidden_uniforms(7, 3, [5, 9], K=10, Lh=6)
    assert u.shape == (2, 10, 1, 6)
    assert (u >= 0).all() and (u < 1).all()
    # unit (n=9,k=6,s=4): group 0, slot 6 -> bits 72..83 of the 128-bit output,
    # i.e. bits 8..19 of word 2, of the coarse (sub 0) and fine (sub 1) calls
    rc = philox4x32(9, 4, (1 << 28), 3, 7, 0)
    rf = philox4x32(9, 4, (1 << 28) | (1 << 16), 3, 7, 0)
    coarse, fine = (int(rc[2]) >> 8) & 0xFFF, (int(rf[2]) >> 8) & 0xFFF
    assert u[1, 6, 0, 4] == (coarse * 4096 + fine) * 2.0 ** -24
    # slot 5 straddles words 1 and 2 (bits 60..71)
    c5 = ((int(rc[1]) >> 28) | (int(rc[2]) << 4)) & 0xFFF
    f5 = ((int(rf[1]) >> 28) | (int(rf[2]) << 4)) & 0xFFF
    assert u[1, 5, 0, 4] == (c5 * 4096 + f5) * 2.0 ** -24
    u25 = hidden_uniforms(7, 3, [5], K=25, Lh=3, strand=1)
    r2 = philox4x32(5, 2, (1 << 28) | (1 << 24) | 2, 3, 7, 0)       # group 2, slot 0 -> k = 20
    r2f = philox4x32(5, 2, (1 << 28) | (1 << 24) | (1 << 16) | 2, 3, 7, 0)
    assert u25[0, 20, 0, 2] == ((int(r2[0]) & 0xFFF) * 4096 + (int(r2f[0]) & 0xFFF)) * 2.0 ** -24
    v = visible_uniforms(7, 3, [5, 9], L=11)
    r = philox4x32(5, 10 >> 2, 2 << 28, 3, 7, 0)
    assert v[0, 10] == (int(r[10 & 3]) >> 8) * 2.0 ** -24
    # float32-exact
    assert np.array_equal(u.astype(np.float32).astype(np.float64), u)


# --------------------------------------------------------------------------
# bottom-up: reference tests/testcrbm.py:154-200
# --------------------------------------------------------------------------
@pytest.mark.parametrize("flip", [False, True])
def test_bottomup_control(flip):
    data = synthetic_onehot(11, 200, seed=3)
    nmot, mlen = 10, 5
    model = OracleCRBM(nmot, mlen, W=np.random.default_rng(0).standard_normal((nmot, 1, 4, mlen)))
    w = model.W[:, :, ::-1, ::-1] if flip else model.W        # :173-176
    b = model.b
    output = model._bottomUpActivity(data, flip)
    output_control = np.zeros(output.shape)
    for seq in range(data.shape[0]):                          # :182-189
        for s in range(data.shape[3] - w.shape[3] + 1):
            for m in range(w.shape[0]):
                output_control[seq, m, 0, s] += \
                    np.multiply(w[m, 0, :, :], data[seq, 0, :, s:(s + w.shape[3])]).sum() + b[0, m]
    np.testing.assert_allclose(output, output_control, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(sigmoid(output_control), model._bottomUpProbability(output),
                               rtol=1e-5, atol=1e-5)           # :193-194
    p, _ = model._computeHgivenV(data, flip)                   # :196-200
    np.testing.assert_allclose(p, sigmoid(output_control), rtol=1e-5, atol=1e-5)
    assert p.shape == (data.shape[0], nmot, 1, data.shape[3] - mlen + 1)


def test_bottomup_matches_scipy_correlate():
    """Theano conv2d valid/filter_flip=False == correlate2d(mode='valid');
    filter_flip=True == correlation with W[::-1, ::-1]."""
    data = synthetic_onehot(3, 40, seed=4)
    model = OracleCRBM(4, 7, W=np.random.default_rng(1).standard_normal((4, 1, 4, 7)))
    for flip in (False, True):
        act = model._bottomUpActivity(data, flip)
        for n in range(3):
            for k in range(4):
                w = model.W[k, 0]
                if flip:
                    ref = scipy.signal.convolve2d(data[n, 0].astype(float), w, mode='valid')
                else:
                    ref = scipy.signal.correlate2d(data[n, 0].astype(float), w, mode='valid')
                np.testing.assert_allclose(act[n, k], ref + model.b[0, k], atol=1e-12)


# --------------------------------------------------------------------------
# top-down: reference tests/testcrbm.py:202-220, :319-504
# --------------------------------------------------------------------------
def controlTopDownActivity(w, c, data, datap=None):            # :202-220
    seqlen = data.shape[3] + w.shape[3] - 1
    nseq, nmot, mlen = data.shape[0], w.shape[0], w.shape[3]
    output_control = np.zeros((nseq, 1, 4, seqlen))
    output_control += c[np.newaxis, 0, :, np.newaxis]
    for seq in range(nseq):
        for pos in range(seqlen - mlen, -1, -1):
            for m in range(nmot):
                output_control[seq, 0, :, pos:(pos + mlen)] += \
                    w[m, 0, :, :] * data[seq, m, 0, pos] + \
                    w[m, 0, ::-1, ::-1] * (0.0 if datap is None else datap[seq, m, 0, pos])
    return output_control


def _hidden(kind, rng, nseq=11, nmot=10, n=196):
    if kind == "zeros":
        return np.zeros((nseq, nmot, 1, n))
    if kind == "ones":
        return np.ones((nseq, nmot, 1, n))
    return rng.binomial(1, 0.1, size=(nseq, nmot, 1, n)).astype(float)   # :443-444


@pytest.mark.parametrize("kind", ["zeros", "ones", "random"])
@pytest.mark.parametrize("ds", [False, True])
def test_topdown_control(kind, ds):
    rng = np.random.default_rng(5)
    nmot, mlen = 10, 5
    model = OracleCRBM(nmot, mlen, W=rng.standard_normal((nmot, 1, 4, mlen)))
    model.c = rng.standard_normal((1, 4)) * (0 if kind == "zeros" else 0.3)
    h = _hidden(kind, rng)
    hp = _hidden(kind, rng) if ds else None
    oa = model._topDownActivity(h, hp)
    op = model._topDownProbability(oa)
    op2, _ = model._computeVgivenH(h, hp)
    ctrl = controlTopDownActivity(model.W, model.c, h, hp)
    assert oa.shape == (11, 1, 4, 200)                         # :384-386
    np.testing.assert_allclose(oa, ctrl, rtol=1e-5, atol=1e-5)  # :389
    ctrl_p = np.exp(ctrl) / np.exp(ctrl).sum(axis=2, keepdims=True)   # :393-394
    np.testing.assert_allclose(ctrl_p, op, rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(op, op2)                     # :337
    if kind == "zeros":
        np.testing.assert_allclose(0.25, op, rtol=1e-5, atol=1e-5)     # :340, :363


def test_topdown_matches_scipy_full_convolution():
    rng = np.random.default_rng(6)
    model = OracleCRBM(3, 6, W=rng.standard_normal((3, 1, 4, 6)))
    h = rng.binomial(1, 0.2, size=(2, 3, 1, 30)).astype(float)
    act = model._topDownActivity(h, None)
    for n in range(2):
        ref = sum(scipy.signal.convolve2d(h[n, k], model.W[k, 0], mode='full') for k in range(3))
        np.testing.assert_allclose(act[n, 0], ref, atol=1e-12)


@pytest.mark.parametrize("ds", [False, True])
def test_topdown_full_sampling_properties(ds):
    """reference tests/testcrbm.py:257-316: probabilities sum to N*L and the
    sample has exactly one 1 per position."""
    data = synthetic_onehot(11, 200, seed=8)
    model = OracleCRBM(10, 5, doublestranded=ds, seed=11)
    K, Lh = 10, 196
    idx = np.arange(11)
    _, h1 = model._computeHgivenV(data, False, hidden_uniforms(11, 0, idx, K, Lh, 0))
    h2 = model._computeHgivenV(data, True, hidden_uniforms(11, 0, idx, K, Lh, 1))[1] if ds else None
    p, s = model._computeVgivenH(h1, h2, visible_uniforms(11, 0, idx, 200))
    assert p.shape == data.shape
    np.testing.assert_allclose(p.sum(), 11 * 200, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(s.sum(), 11 * 200, rtol=1e-4, atol=1e-4)
    assert set(np.unique(s)) <= {0.0, 1.0}
    np.testing.assert_array_equal(s.sum(axis=2), 1.0)


def test_sampling_rules():
    model = OracleCRBM(2, 3, doublestranded=False)
    P = np.array([0.2, 0.5, 0.9, 0.0]).reshape(1, 4, 1, 1)
    u = np.array([0.2, 0.49, 0.95, 0.0]).reshape(1, 4, 1, 1)
    model.num_motifs = 4
    np.testing.assert_array_equal(model._bottomUpSample(P, u).ravel(), [0, 1, 0, 0])  # h = p > u
    pv = np.array([[0.1, 0.2, 0.3, 0.4]]).T.reshape(1, 1, 4, 1).repeat(5, axis=3)
    uu = np.array([[0.0, 0.1, 0.299, 0.61, 0.9999999]])
    s = model._topDownSample(pv, uu)
    np.testing.assert_array_equal(letters_of(s), [[0, 1, 1, 3, 3]])


# --------------------------------------------------------------------------
# statistics / sparsity / update: specified by source text (convRBM.py:327-451)
# --------------------------------------------------------------------------
def test_statistics_bruteforce():
    rng = np.random.default_rng(9)
    N, K, M, L = 3, 4, 5, 17
    Lh = L - M + 1
    model = OracleCRBM(K, M, W=rng.standard_normal((K, 1, 4, M)))
    data = synthetic_onehot(N, L, seed=10)
    P = rng.random((N, K, 1, Lh))
    Pp = rng.random((N, K, 1, Lh))
    vh = model._collectVHStatistics(P, data)
    ref = np.zeros((K, 1, 4, M))
    for k in range(K):
        for a in range(4):
            for j in range(M):
                ref[k, 0, a, j] = sum(data[n, 0, a, s + j] * P[n, k, 0, s]
                                      for n in range(N) for s in range(Lh)) / (N * Lh)
    np.testing.assert_allclose(vh, ref, atol=1e-12)
    avh, ah, av = model._collectUpdateStatistics(P, Pp, data)
    refp = model._collectVHStatistics(Pp, data)
    np.testing.assert_allclose(avh, (ref + refp[:, :, ::-1, ::-1]) / 2, atol=1e-12)   # :367
    np.testing.assert_allclose(ah[0], (P.mean(axis=(0, 2, 3)) + Pp.mean(axis=(0, 2, 3))) / 2)
    f = data.mean(axis=(0, 1, 3))
    np.testing.assert_allclose(av[0], f + f[::-1])            # :345 (a sum)
    np.testing.assert_allclose(av.sum(), 2.0)


@pytest.mark.parametrize("pooling", [1, 2, 3])
def test_sparsity_gradient_matches_finite_differences(pooling):
    """the closed form of T.grad(entropy penalty) (convRBM.py:440-451), for independent units and
    for pooled groups (hidden length 18 = 21 - 4 + 1, divisible by 1, 2 and 3)"""
    rng = np.random.default_rng(12)
    K, M = 3, 4
    model = OracleCRBM(K, M, doublestranded=True, rho=0.05, pooling=pooling,
                       W=rng.standard_normal((K, 1, 4, M)))
    model.b = model.b + rng.standard_normal((1, K)) * 0.1 + 3.0
    data = synthetic_onehot(6, 21, seed=13)
    reg_W, reg_b = model._gradientSparsityConstraintEntropy(data)
    eps = 1e-6
    num_W = np.zeros_like(model.W)
    for idx in np.ndindex(*model.W.shape):
        w0 = model.W[idx]
        model.W[idx] = w0 + eps
        ep = model.sparsity_penalty(data)
        model.W[idx] = w0 - eps
        em = model.sparsity_penalty(data)
        model.W[idx] = w0
        num_W[idx] = -(ep - em) / (2 * eps)
    np.testing.assert_allclose(reg_W, num_W, rtol=1e-5, atol=1e-8)
    num_b = np.zeros_like(model.b)
    for k in range(K):
        b0 = model.b[0, k]
        model.b[0, k] = b0 + eps
        ep = model.sparsity_penalty(data)
        model.b[0, k] = b0 - eps
        em = model.sparsity_penalty(data)
        model.b[0, k] = b0
        num_b[0, k] = -(ep - em) / (2 * eps)
    np.testing.assert_allclose(reg_b, num_b, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("ds", [False, True])
def test_train_step_equals_raw_sum_form(ds):
    """The data-parallel formulation (raw sums -> finalize) reproduces
    convRBM.py:373-438, also when the data batch and length differ from the
    fantasy batch (short last slice; hidden length hard-coded 200)."""
    def make():
        return OracleCRBM(4, 5, doublestranded=ds, batchsize=6, cd_k=2, seed=3,
                          fantasy_hidden_len=30,
                          W=np.random.default_rng(14).standard_normal((4, 1, 4, 5)))
    D = synthetic_onehot(5, 41, seed=15)
    a, b = make(), make()
    a.train_step(D)
    P_m, P_mp, v_m = b.gibbs_steps(b.cd_k)
    b.finalize_from_sums(b.local_sums(D, P_m, P_mp, v_m), 41, 30)
    for x, y in ((a.W, b.W), (a.b, b.b), (a.c, b.c), (a.vW, b.vW)):
        np.testing.assert_allclose(x, y, rtol=1e-10, atol=1e-13)
    np.testing.assert_array_equal(a.fantasy_h, b.fantasy_h)
    assert a.gibbs_step == 2
    assert a.last_v_model.shape == (6, 1, 4, 34)
    # c keeps c[A]=c[T], c[C]=c[G] (symmetrised letter statistics)
    np.testing.assert_allclose(a.c[0], a.c[0, ::-1], atol=1e-15)


def test_gibbs_chain_is_persistent_and_deterministic():
    def make():
        return OracleCRBM(3, 4, doublestranded=True, batchsize=4, seed=21,
                          fantasy_hidden_len=12,
                          W=np.random.default_rng(2).standard_normal((3, 1, 4, 4)) * 2)
    a, b = make(), make()
    a.gibbs_steps(3)
    b.gibbs_steps(1)
    b.gibbs_steps(2)
    np.testing.assert_array_equal(a.fantasy_h, b.fantasy_h)
    np.testing.assert_array_equal(a.fantasy_h_prime, b.fantasy_h_prime)
    assert a.fantasy_h.sum() > 0
    # shard invariance: chains 2..3 on a "second rank" equal the tail of the batch
    c = make()
    c.fantasy_h = c.fantasy_h[2:]
    c.fantasy_h_prime = c.fantasy_h_prime[2:]
    c.seq_offset = 2
    c.gibbs_steps(3)
    np.testing.assert_array_equal(a.fantasy_h[2:], c.fantasy_h)


# --------------------------------------------------------------------------
# evaluation (convRBM.py:466-514, :636-697)
# --------------------------------------------------------------------------
@pytest.mark.parametrize("ds", [False, True])
def test_free_energy_bruteforce(ds):
    rng = np.random.default_rng(16)
    K, M, N, L = 3, 4, 5, 15
    model = OracleCRBM(K, M, doublestranded=ds, W=rng.standard_normal((K, 1, 4, M)))
    model.c = rng.standard_normal((1, 4))
    model.b = model.b + 5
    data = synthetic_onehot(N, L, seed=17)
    fe = model.freeEnergy(data)
    fem = model.freeEnergy(data, permotif=True)
    assert fe.shape == (N,)                                    # tests/testcrbm.py:519
    assert fem.shape == (N, K)
    x = model._bottomUpActivity(data)
    xp = model._bottomUpActivity(data, True)
    for n in range(N):
        cterm = sum(data[n, 0, a, p] * model.c[0, a] for a in range(4) for p in range(L))
        tot = 0.0
        for k in range(K):
            t = -np.log(1 + np.exp(x[n, k, 0])).sum()
            if ds:
                t -= np.log(1 + np.exp(xp[n, k, 0])).sum()
            np.testing.assert_allclose(fem[n, k], t - cterm, rtol=1e-12)
            tot += t
        np.testing.assert_allclose(fe[n], (tot - cterm) / L, rtol=1e-12)
    np.testing.assert_allclose(model._meanFreeEnergy(data), fe.mean(), rtol=1e-12)


def test_hit_probs_branches():
    data = synthetic_onehot(4, 30, seed=18)
    W = np.random.default_rng(19).standard_normal((5, 1, 4, 6))
    ds = OracleCRBM(5, 6, doublestranded=True, W=W)
    ss = OracleCRBM(5, 6, doublestranded=False, W=W)
    x, xp = ss._bottomUpActivity(data), ss._bottomUpActivity(data, True)
    np.testing.assert_allclose(ds.motifHitProbs(data), sigmoid(ds._bottomUpActivity(data)))
    np.testing.assert_allclose(ss.motifHitProbs(data), sigmoid(x + xp))       # 2*b included
    assert ss.motifHitProbs(data).shape == (4, 5, 1, 25)      # tests/testcrbm.py:129


def test_evaluate_params_and_pfms():
    model = OracleCRBM(10, 15)
    twn, ic, medic = model.evaluateParams()
    W = model.W
    np.testing.assert_allclose(twn, np.sqrt((W ** 2).mean()))
    pwm = np.exp(W) / np.exp(W).sum(axis=2, keepdims=True)
    ent = -(pwm * np.log2(pwm)).sum(axis=2)
    np.testing.assert_allclose(ic, 2 - ent.mean())
    np.testing.assert_allclose(medic, 2 - np.mean(np.sort(ent, axis=2)[:, :, 7]))
    pfms = model.getPFMs()
    assert len(pfms) == 10                                     # tests/testcrbm.py:143
    for p in pfms:
        np.testing.assert_allclose(p.sum(), 15)                # :146
    # zero W: every column is uniform => IC 0
    z = OracleCRBM(2, 3, W=np.zeros((2, 1, 4, 3)))
    np.testing.assert_allclose(z.evaluateParams(), [0, 0, 0], atol=1e-12)


def test_init_and_batching():
    m = OracleCRBM(10, 15)
    np.testing.assert_allclose(m.b, -9.0099, atol=1e-4)
    assert m.fantasy_h.shape == (20, 10, 1, 200)               # convRBM.py:168
    m2 = OracleCRBM(10, 15, rho=0.0)                           # :136-140
    assert m2.rho == 1.0 / 300
    m3 = OracleCRBM(10, 15, rho=0.0, doublestranded=False)
    assert m3.rho == 1.0 / 150
    assert OracleCRBM._iterateBatchIndices(45, 20) == [[0, 20], [20, 40], [40, 45]]
    assert OracleCRBM._iterateBatchIndices(40, 20) == [[0, 20], [20, 40]]


def test_pooling_probability():
    """convRBM.py:245-257 for pooling > 1: exp(x)/(pool + sum exp)."""
    m = OracleCRBM(2, 3, pooling=2, doublestranded=False)
    act = np.random.default_rng(20).standard_normal((2, 2, 1, 6))
    p = m._bottomUpProbability(act)
    x = act.reshape(2, 2, 1, 3, 2)
    ref = np.exp(x) / (2 + np.exp(x).sum(axis=4, keepdims=True))
    np.testing.assert_allclose(p, ref.reshape(2, 2, 1, 6))


def test_vh_statistics_as_the_reference_writes_them():
    """_collectVHStatistics literally (convRBM.py:327-337): conv2d(data.dimshuffle(1,0,2,3), P.dimshuffle(1,0,2,3),
    border_mode="valid", filter_flip=False) -- the N sequences become the CHANNELS of one 4 x L image, the K motifs' 1 x Lh
    probability rows the filters -- divided by prod(P.dimshuffle(1,0,2,3).shape[1:]) = N * 1 * Lh and shuffled back.
    Theano's conv2d without filter flip is a per-channel valid cross-correlation summed over channels, which
    scipy.signal.correlate2d gives independently of the oracle's einsum."""
    import scipy.signal
    rng = np.random.default_rng(12)
    N, K, M, L = 5, 3, 4, 23
    Lh = L - M + 1
    o = OracleCRBM(K, M, doublestranded=True, batchsize=N, seed=1)
    D = synthetic_onehot(N, L, seed=3).astype(np.float64)
    P = rng.random((N, K, 1, Lh))
    image = np.transpose(D, (1, 0, 2, 3))            # (1, N, 4, L): batch 1, N channels
    filters = np.transpose(P, (1, 0, 2, 3))          # (K, N, 1, Lh): K filters of N channels
    out = np.zeros((1, K, 4, M))
    for k in range(K):
        for n in range(N):
            out[0, k] += scipy.signal.correlate2d(image[0, n], filters[k, n], mode="valid")
    out /= np.prod(filters.shape[1:])
    want = np.transpose(out, (1, 0, 2, 3))           # (K,1,4,M)
    np.testing.assert_allclose(o._collectVHStatistics(P, D), want, rtol=1e-12, atol=1e-14)
    # and the strand merge of _collectUpdateStatistics (:364-368): (VH + VH'[:, :, ::-1, ::-1]) / 2, (H + H') / 2
    Pp = rng.random((N, K, 1, Lh))
    avh, ah, av = o._collectUpdateStatistics(P, Pp, D)
    vhp = o._collectVHStatistics(Pp, D)
    np.testing.assert_allclose(avh, (want + vhp[:, :, ::-1, ::-1]) / 2.0, rtol=1e-12)
    np.testing.assert_allclose(ah, (P.mean(axis=(0, 2, 3)) + Pp.mean(axis=(0, 2, 3)))[None, :] / 2.0, rtol=1e-12)
    a = D.mean(axis=(0, 1, 3))
    np.testing.assert_allclose(av, (a + a[::-1])[None, :], rtol=1e-12)
