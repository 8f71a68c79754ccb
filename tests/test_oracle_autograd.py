"""The oracle against a second, structurally different restatement of the reference's graph (CPU, float64).

Theano cannot be imported here, so the oracle's closed forms for the rows no reference test pins -- the
statistics, the entropy-sparsity gradient, the PCD update -- are checked against a transcription of the reference's
graph AS IT IS WRITTEN, in another framework: every `conv` of convRBM.py becomes `torch.nn.functional.conv2d` with the
reference's own axis shuffles and flips (Theano `filter_flip=False` = cross-correlation = conv2d; `filter_flip=True` =
the filter flipped in both spatial axes; `border_mode='full'` = padding of kernel size - 1), and the sparsity gradient
comes from AUTOGRAD of the penalty (the reference calls T.grad, convRBM.py:447-449) instead of the closed form the
oracle and the kernels use.  Same injected uniforms, same sampling rule; the parameters after PCD steps must agree to
float64 rounding.  Test infrastructure only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.crbm_oracle import (OracleCRBM, synthetic_onehot, hidden_uniforms, visible_uniforms, KIND_CHAIN_H, KIND_CHAIN_V)

T64 = torch.float64


def conv(x, w, border_mode="valid", filter_flip=False):
    """theano.tensor.nnet.conv2d(x, w, border_mode=, filter_flip=) on NCHW tensors"""
    if filter_flip:
        w = torch.flip(w, dims=(2, 3))
    pad = (w.shape[2] - 1, w.shape[3] - 1) if border_mode == "full" else (0, 0)
    return F.conv2d(x, w, padding=pad)


class GraphCRBM:
    """convRBM.py:238-451 line by line on torch tensors"""

    def __init__(self, o):
        self.motifs = torch.tensor(o.W, dtype=T64, requires_grad=True)          # (K,1,A,M)
        self.bias = torch.tensor(o.b, dtype=T64, requires_grad=True)            # (1,K)
        self.c = torch.tensor(o.c, dtype=T64)                                   # (1,A)
        self.vW, self.vb, self.vc = (torch.tensor(x, dtype=T64) for x in (o.vW, o.vb, o.vc))
        self.fantasy_h = torch.tensor(o.fantasy_h, dtype=T64)
        self.fantasy_h_prime = torch.tensor(o.fantasy_h_prime, dtype=T64) if o.doublestranded else None
        self.ds, self.pool, self.rho = o.doublestranded, o.pooling, o.rho
        self.mu, self.alpha, self.sp = o.momentum, o.learning_rate, o.lambda_rate

    def bottomUpActivity(self, data, flip_motif=False):                          # :238-243
        out = conv(data, self.motifs, filter_flip=flip_motif)
        return out + self.bias.permute(1, 0)[None, :, :, None].reshape(1, -1, 1, 1)

    def bottomUpProbability(self, act):                                          # :245-257
        pool = self.pool
        x = act.reshape(act.shape[0], act.shape[1], act.shape[2], act.shape[3] // pool, pool)
        norm = torch.sum(1.0 + torch.exp(x), dim=4, keepdim=True)
        return (torch.exp(x) / norm).reshape(act.shape)

    def bottomUpSample(self, probs, u):                                          # :259-267, one uniform per pooling group
        pool = self.pool
        p = probs.reshape(*probs.shape[:3], probs.shape[3] // pool, pool)
        ug = u.reshape(*probs.shape[:3], probs.shape[3] // pool, pool)[..., :1]
        cum = torch.cumsum(p, dim=4)
        before = torch.cat([torch.zeros_like(cum[..., :1]), cum[..., :-1]], dim=4)
        return ((cum > ug) & (before <= ug)).to(T64).reshape(probs.shape)

    def computeHgivenV(self, data, flip=False, u=None):                          # :269-275
        prob = self.bottomUpProbability(self.bottomUpActivity(data, flip))
        return prob, (None if u is None else self.bottomUpSample(prob, u))

    def topDownActivity(self, h, hprime):                                        # :277-292
        W = self.motifs.permute(1, 0, 2, 3)
        C = conv(h, W, border_mode="full", filter_flip=True)
        out = torch.sum(C, dim=1, keepdim=True)
        if hprime is not None:
            C = conv(hprime, torch.flip(W, dims=(2, 3)), border_mode="full", filter_flip=True)
            out = out + torch.sum(C, dim=1, keepdim=True)
        return out + self.c[None, :, :, None].reshape(1, 1, -1, 1)

    def computeVgivenH(self, h, hprime, u):                                      # :294-325
        act = self.topDownActivity(h, hprime)
        prob = torch.exp(act) / torch.sum(torch.exp(act), dim=2, keepdim=True)   # _softmax, :699-702
        cum = torch.cumsum(prob[:, 0], dim=1)                                    # first letter with cumsum > u
        A = prob.shape[2]
        idx = torch.sum(cum[:, :A - 1, :] <= u[:, None, :], dim=1)
        return prob, F.one_hot(idx, A).permute(0, 2, 1)[:, None].to(T64)

    def collectVHStatistics(self, prob_of_H, data):                              # :327-337
        d = data.permute(1, 0, 2, 3)
        p = prob_of_H.permute(1, 0, 2, 3)
        avh = conv(d, p, border_mode="valid", filter_flip=False)
        avh = avh / float(np.prod(p.shape[1:]))
        return avh.permute(1, 0, 2, 3)

    def collectUpdateStatistics(self, P, Pp, data):                              # :339-371
        avh = self.collectVHStatistics(P, data)
        ah = torch.mean(P, dim=(0, 2, 3))[None, :]
        if Pp is not None:
            avhp = self.collectVHStatistics(Pp, data)
            ahp = torch.mean(Pp, dim=(0, 2, 3))[None, :]
            avh = (avh + torch.flip(avhp, dims=(2, 3))) / 2.0
            ah = (ah + ahp) / 2.0
        a = torch.mean(data, dim=(0, 1, 3))[None, :]
        av = a + torch.flip(a, dims=(1,))                                        # inc_subtensor(a, a[:, ::-1]), :345
        return avh, ah, av

    def gradientSparsityConstraintEntropy(self, data):                           # :440-451
        prob, _ = self.computeHgivenV(data)
        q = self.rho
        p = torch.mean(prob, dim=(0, 2, 3))
        pen = torch.mean(q * torch.log(p) + (1 - q) * torch.log(1 - p))
        gW, gb = torch.autograd.grad(pen, (self.motifs, self.bias))
        return -gW, -gb

    def updateWeightsOnMinibatch(self, D, k, uniforms):                          # :373-438
        with torch.no_grad():
            Pd, _ = self.computeHgivenV(D)
            Pdp = self.computeHgivenV(D, True)[0] if self.ds else None
            GW_d, Gb_d, Gc_d = self.collectUpdateStatistics(Pd, Pdp, D)
            h, hp = self.fantasy_h, self.fantasy_h_prime
            for i in range(k):
                uv, uh, uhp = uniforms[i]
                _, v = self.computeVgivenH(h, hp, uv)
                Pm, h = self.computeHgivenV(v, False, uh)
                Pmp, hp = self.computeHgivenV(v, True, uhp) if self.ds else (None, None)
            GW_m, Gb_m, Gc_m = self.collectUpdateStatistics(Pm, Pmp, v)
        regW, regb = self.gradientSparsityConstraintEntropy(D)
        with torch.no_grad():
            self.vW = self.mu * self.vW + self.alpha * (GW_d - GW_m - self.sp * regW)
            self.vb = self.mu * self.vb + self.alpha * (Gb_d - Gb_m - self.sp * regb)
            self.vc = self.mu * self.vc + self.alpha * (Gc_d - Gc_m)
            self.motifs = (self.motifs + self.vW).detach().requires_grad_(True)
            self.bias = (self.bias + self.vb).detach().requires_grad_(True)
            self.c = self.c + self.vc
            self.fantasy_h, self.fantasy_h_prime = h, hp


@pytest.mark.parametrize("K,M,ds,pool,A", [(6, 5, True, 1, 4), (10, 15, False, 1, 4), (4, 6, True, 2, 4), (5, 7, False, 3, 4),
                                           (3, 4, True, 1, 3), (4, 5, False, 1, 20)])
def test_pcd_steps_match_the_graph_transcription(K, M, ds, pool, A):
    Lf = 24
    L = M - 1 + Lf
    rng = np.random.default_rng(11 * K + M)
    o = OracleCRBM(K, M, doublestranded=ds, batchsize=5, cd_k=2, fantasy_hidden_len=Lf, seed=17, pooling=pool, input_dims=A,
                   rho=0.05, W=rng.standard_normal((K, 1, A, M)) * 0.8)
    o.b = o.b + 5.0
    o.c = rng.standard_normal((1, A)) * 0.2
    o.fantasy_h = rng.binomial(1, 0.1, size=o.fantasy_h.shape).astype(np.float64)
    if ds:
        o.fantasy_h_prime = rng.binomial(1, 0.1, size=o.fantasy_h.shape).astype(np.float64)
    if pool > 1:                                              # a valid pooled state: at most one unit on per group
        for hh in ([o.fantasy_h, o.fantasy_h_prime] if ds else [o.fantasy_h]):
            hh.reshape(5, K, 1, -1, pool)[..., 1:] = 0.0
    g = GraphCRBM(o)
    idx = np.arange(5)
    for step in range(3):
        D = synthetic_onehot(7, L, seed=50 + step, A=A)
        uni = []
        for i in range(o.cd_k):
            t = o.gibbs_step + i
            uv = torch.tensor(visible_uniforms(o.seed, t, idx, L, KIND_CHAIN_V), dtype=T64)
            uh = torch.tensor(hidden_uniforms(o.seed, t, idx, K, Lf, 0, KIND_CHAIN_H), dtype=T64)
            uhp = torch.tensor(hidden_uniforms(o.seed, t, idx, K, Lf, 1, KIND_CHAIN_H), dtype=T64) if ds else None
            uni.append((uv, uh, uhp))
        g.updateWeightsOnMinibatch(torch.tensor(D, dtype=T64), o.cd_k, uni)
        o.train_step(D)
        np.testing.assert_array_equal(g.fantasy_h.numpy(), o.fantasy_h)       # the same chain, sample for sample
        if ds:
            np.testing.assert_array_equal(g.fantasy_h_prime.numpy(), o.fantasy_h_prime)
        np.testing.assert_allclose(g.motifs.detach().numpy(), o.W, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(g.bias.detach().numpy(), o.b, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(g.c.numpy(), o.c, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(g.vW.numpy(), o.vW, rtol=1e-9, atol=1e-13)
    assert o.fantasy_h.sum() > 0


def test_free_energy_is_minus_log_of_the_summed_out_hidden_layer():
    """convRBM.py:657-676 against the definition: exp(-F(v) L) = sum over all hidden configurations of exp(-E(v, h)) for a
    model small enough to enumerate (pooling 1: units independent, so the sum factorises -- checked by brute force over the
    2^(K Lh) states of a 2-motif model on a 4-position hidden layer)."""
    import itertools
    K, M, L = 2, 3, 6
    rng = np.random.default_rng(5)
    o = OracleCRBM(K, M, doublestranded=False, W=rng.standard_normal((K, 1, 4, M)))
    o.b = rng.standard_normal((1, K))
    o.c = rng.standard_normal((1, 4)) * 0.3
    D = synthetic_onehot(3, L, seed=9)
    x = o._bottomUpActivity(D)                                  # (n,K,1,Lh)
    Lh = L - M + 1
    for n in range(3):
        vis = float((D[n, 0] * o.c.reshape(4, 1)).sum())
        z = 0.0
        for bits in itertools.product((0, 1), repeat=K * Lh):
            h = np.array(bits, dtype=np.float64).reshape(K, Lh)
            z += np.exp((h * x[n, :, 0, :]).sum() + vis)       # -E(v,h) = sum h x + sum v c
        np.testing.assert_allclose(o.freeEnergy(D)[n], -np.log(z) / L, rtol=1e-12)


@pytest.mark.parametrize("K,M,ds,pool,A", [(6, 5, True, 1, 4), (7, 9, False, 1, 4), (4, 6, True, 2, 4), (5, 7, False, 3, 4), (4, 5, True, 1, 20)])
def test_evaluation_graphs_match_the_transcription(K, M, ds, pool, A):
    """convRBM.py:466-514 and :657-697 transcribed: free energy per data point (divided by the sequence length) and per motif
    (not divided), the mean free energy, the status-line metrics (rms of the filters, information content against
    log2 of the alphabet size, its median form) and the two branches of getHitProbs -- doublestranded: the forward strand's
    probabilities; single-stranded: the probability of the SUM of both strands' activations (:507-514)."""
    Lf = 6 * pool
    L = M - 1 + Lf
    rng = np.random.default_rng(3 * K + M)
    o = OracleCRBM(K, M, doublestranded=ds, pooling=pool, input_dims=A, W=rng.standard_normal((K, 1, A, M)) * 0.9)
    o.b = o.b + 6.0
    o.c = rng.standard_normal((1, A)) * 0.3
    g = GraphCRBM(o)
    Dn = synthetic_onehot(5, L, seed=K, A=A)
    D = torch.tensor(Dn, dtype=T64)

    def fe(per_motif):
        axes = (2, 3) if per_motif else (1, 2, 3)
        with torch.no_grad():
            x = g.bottomUpActivity(D)
            x = x.reshape(*x.shape[:3], x.shape[3] // pool, pool)
            f = -torch.sum(torch.log(1.0 + torch.sum(torch.exp(x), dim=4)), dim=axes)
            if ds:
                x = g.bottomUpActivity(D, True)
                x = x.reshape(*x.shape[:3], x.shape[3] // pool, pool)
                f = f - torch.sum(torch.log(1.0 + torch.sum(torch.exp(x), dim=4)), dim=axes)
            vis = torch.sum(D * g.c.reshape(1, 1, -1, 1), dim=(1, 2, 3))
            return (f - vis[:, None]) if per_motif else (f - vis) / D.shape[3]
    np.testing.assert_allclose(o.freeEnergy(Dn), fe(False).numpy(), rtol=1e-12)
    np.testing.assert_allclose(o.freeEnergy(Dn, True), fe(True).numpy(), rtol=1e-12)
    np.testing.assert_allclose(o._meanFreeEnergy(Dn), float(torch.sum(fe(False)) / D.shape[0]), rtol=1e-12)
    with torch.no_grad():
        W = g.motifs
        twn = torch.sqrt(torch.mean(W ** 2))
        pwm = torch.exp(W) / torch.exp(W).sum(dim=2, keepdim=True)
        ent = torch.sum(-pwm * torch.log2(pwm), dim=2)                       # (K,1,M)
        ic = np.log2(W.shape[2]) - torch.mean(ent)
        medic = np.log2(W.shape[2]) - torch.mean(torch.sort(ent, dim=2)[0][:, :, ent.shape[2] // 2])
        if ds:
            hits = g.bottomUpProbability(g.bottomUpActivity(D))
        else:
            hits = g.bottomUpProbability(g.bottomUpActivity(D) + g.bottomUpActivity(D, True))
    np.testing.assert_allclose(o.evaluateParams(), [float(twn), float(ic), float(medic)], rtol=1e-12)
    np.testing.assert_allclose(o.motifHitProbs(Dn), hits.numpy(), rtol=1e-12)
    pf = o.getPFMs()
    np.testing.assert_allclose(np.stack(pf), pwm[:, 0].numpy(), rtol=1e-12)
