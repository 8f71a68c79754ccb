"""CPU oracle for the CRBM training hot path -- TEST INFRASTRUCTURE ONLY.

This module is a float64 NumPy restatement of the arithmetic that the
reference (`schulter/crbm`, file ``secomo/convRBM.py``) delegates to Theano.
It exists to *check* the HIP path; nothing in the product package
(``crbm_amd/``) imports it.  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import or execute anything under
``oracle/``.

Parity status
-------------
The reference cannot be imported in this pipeline (its engine, Theano, is an
un-vendored, un-pinned dependency declared in the reference's setup.py:20 and is
not installed; there is no network).  The oracle is therefore pinned by the
known-answer *control implementations* that the reference's own tests hold
(tests/testcrbm.py:182-198 forward activation/sigmoid, :202-220 top-down
activation, :393-394 softmax, :340/:363 all-zero-hidden => 0.25) which are
restated in ``tests/test_oracle.py``, plus an independent cross-check against
``scipy.signal.correlate2d/convolve2d``.  Items that no reference test pins
numerically (sampling rule, statistics, sparsity gradient, momentum update,
free energy values, evaluateParams metrics) follow the reference's source text
only -- for those rows parity is "unpinned" and DESIGN.md says so.

Randomness
----------
The reference seeds Theano's MRG31k3p stream from the wall clock
(convRBM.py:155), so no two reference runs agree and its stream cannot be
reproduced.  The oracle and the HIP kernels share a counter-based generator
instead: Philox-4x32-7 (Salmon et al., SC'11; the Random123 constants; seven
rounds is the shortest variant its authors report as passing BigCrush), keyed
by ``seed`` and indexed by (sequence, position, motif-group, strand, kind,
Gibbs step).  Every uniform has 24 bits (exactly representable in float32):
``(r >> 8) * 2**-24`` for visible positions, ``(coarse*4096 + fine) * 2**-24``
from two 12-bit Philox fields for hidden units (see ``hidden_uniforms``), and
the sampling rules are the reference's threshold forms:
``h = 1 if p > u`` (convRBM.py:259-267, multinomial with one outcome) and
"first letter whose cumulative probability exceeds u" (convRBM.py:301-310).
"""
from __future__ import annotations

import numpy as np
import scipy.stats

__all__ = [
    "philox4x32", "hidden_uniforms", "visible_uniforms", "OracleCRBM",
    "synthetic_onehot", "letters_of", "onehot_of", "reverse_complement_filter",
    "KIND_CHAIN_H", "KIND_CHAIN_V", "KIND_EVAL_H", "KIND_API_H", "KIND_API_V",
]

# ----------------------------------------------------------------------------
# Philox-4x32 (published algorithm; Random123 v1.x constants).  The kernels, this
# oracle and oracle/crbm_cpu.c draw PHILOX_ROUNDS = 7 rounds; the implementation is
# pinned against Random123's known-answer vectors for 7 and for 10 rounds
# (tests/test_oracle.py).
# ----------------------------------------------------------------------------
_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)

# "kind" field of the counter (bits 28..31 of word 2)
KIND_CHAIN_H = 1   # hidden sample inside the persistent Gibbs chain
KIND_CHAIN_V = 2   # visible sample inside the persistent Gibbs chain
KIND_EVAL_H = 3    # hidden sample drawn by evaluateData (convRBM.py:469-472)
KIND_API_H = 4     # hidden sample of a stand-alone _computeHgivenV call
KIND_API_V = 5     # visible sample of a stand-alone _computeVgivenH call


PHILOX_ROUNDS = 7


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=PHILOX_ROUNDS):
    """Vectorised Philox-4x32. Inputs broadcast; returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint64) & _MASK
    c1 = np.asarray(c1, dtype=np.uint64) & _MASK
    c2 = np.asarray(c2, dtype=np.uint64) & _MASK
    c3 = np.asarray(c3, dtype=np.uint64) & _MASK
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(rounds):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0), lo1,
                          hi0 ^ c3 ^ np.uint64(k1), lo0)
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32),
            c2.astype(np.uint32), c3.astype(np.uint32))


def _u01(r):
    """24-bit uniform in [0,1): exactly the float32 value the kernels use."""
    return (np.asarray(r, dtype=np.uint32) >> np.uint32(8)).astype(np.float64) * 2.0 ** -24


def _word2(kind, strand, sub, group):
    """Counter word 2: kind | strand | sub-stream (0 coarse, 1 fine) | group."""
    return (int(kind) << 28) | (int(strand) << 24) | (int(sub) << 16) | group


def _fields12(r):
    """The ten 12-bit fields of a Philox output read as one 128-bit
    little-endian number v0 | v1<<32 | v2<<64 | v3<<96; shape (..., 10)."""
    lo = r[0].astype(np.uint64) | (r[1].astype(np.uint64) << np.uint64(32))
    hi = r[2].astype(np.uint64) | (r[3].astype(np.uint64) << np.uint64(32))
    out = []
    for i in range(10):
        bit = 12 * i
        if bit + 12 <= 64:
            f = lo >> np.uint64(bit)
        elif bit >= 64:
            f = hi >> np.uint64(bit - 64)
        else:
            f = (lo >> np.uint64(bit)) | (hi << np.uint64(64 - bit))
        out.append(f & np.uint64(0xFFF))
    return np.stack(out, axis=-1)


def hidden_uniforms(seed, step, seq_index, K, Lh, strand=0, kind=KIND_CHAIN_H):
    """Uniforms for hidden units, shape (len(seq_index), K, 1, Lh).

    Unit (n, k, s) of ``strand`` belongs to sampler group g = k // 10, slot
    i = k % 10.  Two Philox calls per (n, s, strand, g) -- counter
    (n, s, kind<<28 | strand<<24 | sub<<16 | g, step) with sub = 0 ("coarse")
    and sub = 1 ("fine") -- each yield ten 12-bit fields; the unit's 24-bit
    uniform is (coarse_i * 4096 + fine_i) / 2**24.  (The kernels evaluate the
    fine call lazily: the coarse field alone settles all but ~2**-12 of the
    decisions.)
    """
    seq_index = np.asarray(seq_index, dtype=np.uint64)
    n = seq_index[:, None, None]
    g = np.arange((K + 9) // 10, dtype=np.uint64)[None, :, None]
    s = np.arange(Lh, dtype=np.uint64)[None, None, :]
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    parts = []
    for sub in (0, 1):
        w2 = np.uint64(_word2(kind, strand, sub, 0)) | g
        r = philox4x32(n, s, w2, np.uint64(step & 0xFFFFFFFF), k0, k1)
        parts.append(_fields12(r))                       # (N, G, Lh, 10)
    U = parts[0] * np.uint64(4096) + parts[1]
    u = U.astype(np.float64) * 2.0 ** -24
    u = np.moveaxis(u, 3, 2).reshape(len(seq_index), -1, Lh)[:, :K, :]   # (N, G*10, Lh)
    return u[:, :, None, :]


def visible_uniforms(seed, step, seq_index, L, kind=KIND_CHAIN_V):
    """Uniforms for visible positions, shape (len(seq_index), L).

    Position (n, p) uses component ``p & 3`` of
    ``philox(counter=(n, p>>2, kind<<28, step), key=seed)``.
    """
    seq_index = np.asarray(seq_index, dtype=np.uint64)
    n = seq_index[:, None]
    pg = np.arange((L + 3) // 4, dtype=np.uint64)[None, :]
    r = philox4x32(n, pg, np.uint64(_word2(kind, 0, 0, 0)),
                   np.uint64(step & 0xFFFFFFFF),
                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = np.stack([_u01(x) for x in r], axis=2)          # (N, PG, 4)
    return u.reshape(len(seq_index), -1)[:, :L]


# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------
def synthetic_onehot(n, L, seed=1234, dtype=np.float32, A=4):
    """(n,1,A,L) one-hot array, letters i.i.d. uniform (A = input_dims, 4 for DNA)."""
    rng = np.random.default_rng(seed)
    letters = rng.integers(0, A, size=(n, L))
    return onehot_of(letters, dtype, A)


def onehot_of(letters, dtype=np.float32, A=4):
    letters = np.asarray(letters)
    n, L = letters.shape
    out = np.zeros((n, 1, A, L), dtype=dtype)
    out[np.arange(n)[:, None], 0, letters, np.arange(L)[None, :]] = 1
    return out


def letters_of(onehot):
    return np.argmax(np.asarray(onehot)[:, 0], axis=1)


def reverse_complement_filter(W):
    """rc(W) = W[:, :, ::-1, ::-1] (convRBM.py:285, tests/testcrbm.py:173-174)."""
    return W[:, :, ::-1, ::-1]


def _softmax_axis2(x):
    """convRBM.py:699-702 -- exp(x)/sum_a exp(x); evaluated with the max
    subtracted (mathematically identical, no overflow)."""
    x = x - x.max(axis=2, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=2, keepdims=True)


def _softplus(x):
    return np.logaddexp(0.0, x)


# ----------------------------------------------------------------------------
# the model
# ----------------------------------------------------------------------------
class OracleCRBM:
    """float64 restatement of ``secomo.CRBM`` (convRBM.py:25-726).

    Method names follow the reference's private Theano graph builders so the
    parity tests read like tests/testcrbm.py.  All state is plain NumPy.
    """

    def __init__(self, num_motifs, motif_length, epochs=100, input_dims=4,
                 doublestranded=True, batchsize=20, learning_rate=0.1,
                 momentum=0.95, pooling=1, cd_k=5, rho=0.01, lambda_rate=0.1,
                 fantasy_hidden_len=200, seed=0, W=None):
        # convRBM.py:111-123
        self.num_motifs = num_motifs
        self.motif_length = motif_length
        self.input_dims = input_dims
        self.doublestranded = bool(doublestranded)
        self.batchsize = batchsize
        self.learning_rate = learning_rate
        self.momentum = momentum
        self.rho = rho
        self.lambda_rate = lambda_rate
        self.pooling = pooling
        self.cd_k = cd_k
        self.epochs = epochs
        K, M, A = num_motifs, motif_length, input_dims
        # convRBM.py:127-133 (the caller supplies W: the reference draws it
        # from NumPy's global unseeded RNG)
        if W is None:
            W = np.random.default_rng(42).standard_normal((K, 1, A, M))
        self.W = np.asarray(W, dtype=np.float32).astype(np.float64).reshape(K, 1, A, M)
        # convRBM.py:136-140
        if not rho:
            rho = 1.0 / (K * M)
            if self.doublestranded:
                rho = rho / 2.0
            self.rho = rho
        # convRBM.py:143-149 (cast to float32 as floatX=float32 does)
        b = np.zeros((1, K)) + scipy.stats.norm.ppf(self.rho, 0, np.sqrt(M))
        self.b = b.astype(np.float32).astype(np.float64)
        self.c = np.zeros((1, A))                                  # :151-152
        self.vW = np.zeros_like(self.W)                            # :158-166
        self.vb = np.zeros_like(self.b)
        self.vc = np.zeros_like(self.c)
        self.fantasy_hidden_len = fantasy_hidden_len               # :168 (200)
        self.fantasy_h = np.zeros((batchsize, K, 1, fantasy_hidden_len))
        self.fantasy_h_prime = (np.zeros_like(self.fantasy_h)
                                if self.doublestranded else None)  # :170-173
        self.seed = int(seed)
        self.gibbs_step = 0       # Philox counter word 3 for chain draws
        self.seq_offset = 0       # global index of fantasy chain 0 (multi-GPU)
        self.last_v_model = None

    # ---- bottom-up -------------------------------------------------------
    def _bottomUpActivity(self, data, flip_motif=False):
        """convRBM.py:238-243: valid cross-correlation + shared bias."""
        W = reverse_complement_filter(self.W) if flip_motif else self.W
        data = np.asarray(data, dtype=np.float64)
        N, _, A, L = data.shape
        K, _, _, M = W.shape
        Lh = L - M + 1
        win = np.lib.stride_tricks.sliding_window_view(data[:, 0], M, axis=2)  # (N,A,Lh,M)
        out = np.einsum('nasj,kaj->nks', win, W[:, 0], optimize=True)
        out = out[:, :, None, :] + self.b.reshape(1, K, 1, 1)
        assert out.shape == (N, K, 1, Lh)
        return out

    def _bottomUpProbability(self, act):
        """convRBM.py:245-257: exp(x) / sum_pool(1+exp(x)); sigmoid if pool=1."""
        pool = self.pooling
        N, K, one, Lh = act.shape
        x = act.reshape(N, K, one, Lh // pool, pool)
        if pool == 1:
            p = 1.0 / (1.0 + np.exp(-x))
        else:
            norm = np.sum(1.0 + np.exp(x), axis=4, keepdims=True)
            p = np.exp(x) / norm
        return p.reshape(N, K, one, Lh)

    def _bottomUpSample(self, probs, u):
        """convRBM.py:259-267: multinomial over each pooling group given one
        uniform per group (``u`` has the group shape, or the full shape with
        only every ``pool``-th entry used)."""
        pool = self.pooling
        N, K, one, Lh = probs.shape
        p = probs.reshape(N, K, one, Lh // pool, pool)
        if u.shape == probs.shape:
            u = u.reshape(N, K, one, Lh // pool, pool)[..., :1]
        else:
            u = u.reshape(N, K, one, Lh // pool, 1)
        cum = np.cumsum(p, axis=4)
        first = (cum > u) & (np.concatenate(
            [np.zeros_like(cum[..., :1]), cum[..., :-1]], axis=4) <= u)
        return first.astype(np.float64).reshape(N, K, one, Lh)

    def _computeHgivenV(self, data, flip_motif=False, u=None):
        """convRBM.py:269-275. ``u`` = uniforms shaped like the output."""
        prob = self._bottomUpProbability(self._bottomUpActivity(data, flip_motif))
        sample = None if u is None else self._bottomUpSample(prob, u)
        return [prob, sample]

    # ---- top-down --------------------------------------------------------
    def _topDownActivity(self, h, hprime=None):
        """convRBM.py:277-292: full (transposed) convolution + visible bias.
        y[n,0,a,p] = c[a] + sum_{k,j} W[k,a,j] h[n,k,p-j] (+ rc(W) with h')."""
        h = np.asarray(h, dtype=np.float64)
        N, K, _, Lh = h.shape
        M = self.motif_length
        A = self.input_dims
        L = Lh + M - 1
        out = np.zeros((N, 1, A, L))
        for j in range(M):
            out[:, 0, :, j:j + Lh] += np.einsum('nks,ka->nas', h[:, :, 0, :],
                                                self.W[:, 0, :, j])
        if hprime is not None:
            Wrc = reverse_complement_filter(self.W)
            hp = np.asarray(hprime, dtype=np.float64)
            for j in range(M):
                out[:, 0, :, j:j + Lh] += np.einsum('nks,ka->nas', hp[:, :, 0, :],
                                                    Wrc[:, 0, :, j])
        return out + self.c.reshape(1, 1, A, 1)

    def _topDownProbability(self, activity):
        """convRBM.py:294-299, :699-702: softmax over the letter axis."""
        return _softmax_axis2(activity)

    def _topDownSample(self, prob, u):
        """convRBM.py:301-315: per (n,p) pick the first letter a with
        cumsum_a(P) > u.  If rounding leaves cumsum_A <= u (probability
        ~2**-24 per draw) the last letter is taken, so the column is always
        one-hot -- the reference's own tests require exactly one 1 per
        position (tests/testcrbm.py:286-287)."""
        N, _, A, L = prob.shape
        cum = np.cumsum(prob[:, 0], axis=1)                    # (N,A,L)
        idx = np.sum(cum[:, :A - 1, :] <= u[:, None, :], axis=1)   # (N,L)
        return onehot_of(idx, np.float64, A)

    def _computeVgivenH(self, h, hprime=None, u=None):
        """convRBM.py:317-325."""
        prob = self._topDownProbability(self._topDownActivity(h, hprime))
        sample = None if u is None else self._topDownSample(prob, u)
        return [prob, sample]

    # ---- statistics ------------------------------------------------------
    def _collectVHStatistics(self, prob_of_H, data):
        """convRBM.py:327-337: VH[k,0,a,j] = sum_{n,s} v[n,a,s+j] P[n,k,s] / (N*Lh)."""
        N, K, _, Lh = prob_of_H.shape
        M = self.motif_length
        win = np.lib.stride_tricks.sliding_window_view(
            np.asarray(data, dtype=np.float64)[:, 0], Lh, axis=2)   # (N,A,M,Lh)
        vh = np.einsum('najs,nks->kaj', win, prob_of_H[:, :, 0, :], optimize=True)
        assert vh.shape[2] == M
        return (vh / float(N * Lh))[:, None, :, :]

    def _collectVStatistics(self, data):
        """convRBM.py:339-347: mean letter frequency, then a += a[::-1]."""
        a = np.mean(np.asarray(data, dtype=np.float64), axis=(0, 1, 3))[None, :]
        return a + a[:, ::-1]

    def _collectHStatistics(self, prob_of_H):
        """convRBM.py:349-356."""
        return np.mean(prob_of_H, axis=(0, 2, 3))[None, :]

    def _collectUpdateStatistics(self, prob_of_H, prob_of_H_prime, data):
        """convRBM.py:358-371."""
        avh = self._collectVHStatistics(prob_of_H, data)
        ah = self._collectHStatistics(prob_of_H)
        if prob_of_H_prime is not None:
            avh_p = self._collectVHStatistics(prob_of_H_prime, data)
            ah_p = self._collectHStatistics(prob_of_H_prime)
            avh = (avh + avh_p[:, :, ::-1, ::-1]) / 2.0
            ah = (ah + ah_p) / 2.0
        av = self._collectVStatistics(data)
        return avh, ah, av

    def _pooled_slope(self, P):
        """sum over the units i of a pooling group of dP_i/dx_s for every position s:
        with P_i = exp(x_i) / (pool + sum_j exp(x_j)), dP_i/dx_s = delta_is P_i - P_i P_s,
        hence P_s (1 - sum_i P_i); P (1 - P) for pooling == 1."""
        pool = self.pooling
        N, K, one, Lh = P.shape
        grp = P.reshape(N, K, one, Lh // pool, pool)
        return (grp * (1.0 - grp.sum(axis=4, keepdims=True))).reshape(N, K, one, Lh)

    def _gradientSparsityConstraintEntropy(self, data):
        """convRBM.py:440-451 (T.grad of the entropy penalty) in closed form:
        forward strand only; returns (-dE/dW, -dE/db) for
        E = mean_k[q ln p_k + (1-q) ln(1-p_k)], p_k = mean_{n,s} P[n,k,s].
        Checked against finite differences of ``sparsity_penalty`` (tests/test_oracle.py)."""
        P = self._computeHgivenV(data)[0]
        N, K, _, Lh = P.shape
        q = self.rho
        p = np.mean(P, axis=(0, 2, 3))
        g = (q / p - (1.0 - q) / (1.0 - p)) / K
        dP = self._pooled_slope(P)
        win = np.lib.stride_tricks.sliding_window_view(
            np.asarray(data, dtype=np.float64)[:, 0], Lh, axis=2)   # (N,A,M,Lh)
        S_W = np.einsum('najs,nks->kaj', win, dP[:, :, 0, :], optimize=True) / (N * Lh)
        S_b = np.sum(dP, axis=(0, 2, 3)) / (N * Lh)
        reg_motif = -(g[:, None, None] * S_W)[:, None, :, :]
        reg_bias = -(g * S_b)[None, :]
        return reg_motif, reg_bias

    def sparsity_penalty(self, data):
        """The scalar E whose gradient the reference takes (for finite-difference checks)."""
        P = self._computeHgivenV(data)[0]
        p = np.mean(P, axis=(0, 2, 3))
        q = self.rho
        return np.mean(q * np.log(p) + (1 - q) * np.log(1 - p))

    # ---- the PCD-k step ----------------------------------------------------
    def gibbs_steps(self, k):
        """Advance the persistent chain by k Gibbs steps (convRBM.py:397-408).
        Returns (P_h, P_h', v) of the last step."""
        h, hp = self.fantasy_h, self.fantasy_h_prime
        nf, K, _, Lf = h.shape
        Lv = Lf + self.motif_length - 1
        idx = np.arange(nf) + self.seq_offset
        P = Pp = v = None
        for _ in range(k):
            t = self.gibbs_step
            uv = visible_uniforms(self.seed, t, idx, Lv, KIND_CHAIN_V)
            _, v = self._computeVgivenH(h, hp, uv)
            uh = hidden_uniforms(self.seed, t, idx, K, Lf, 0, KIND_CHAIN_H)
            P, h = self._computeHgivenV(v, False, uh)
            if self.doublestranded:
                uhp = hidden_uniforms(self.seed, t, idx, K, Lf, 1, KIND_CHAIN_H)
                Pp, hp = self._computeHgivenV(v, True, uhp)
            self.gibbs_step += 1
        self.fantasy_h, self.fantasy_h_prime = h, hp
        self.last_v_model = v
        return P, Pp, v

    def train_step(self, D):
        """convRBM.py:373-438 -- one SGD-with-momentum update on mini-batch D."""
        D = np.asarray(D, dtype=np.float64)
        # data phase (:377-388); the positive-phase samples are never used
        P_d = self._computeHgivenV(D)[0]
        P_dp = self._computeHgivenV(D, True)[0] if self.doublestranded else None
        G_W_d, G_b_d, G_c_d = self._collectUpdateStatistics(P_d, P_dp, D)
        # model phase (:391-413)
        P_m, P_mp, v_m = self.gibbs_steps(self.cd_k)
        G_W_m, G_b_m, G_c_m = self._collectUpdateStatistics(P_m, P_mp, v_m)
        # sparsity (:418) and momentum update (:415-436)
        reg_W, reg_b = self._gradientSparsityConstraintEntropy(D)
        mu, alpha, sp = self.momentum, self.learning_rate, self.lambda_rate
        self.vW = mu * self.vW + alpha * (G_W_d - G_W_m - sp * reg_W)
        self.vb = mu * self.vb + alpha * (G_b_d - G_b_m - sp * reg_b)
        self.vc = mu * self.vc + alpha * (G_c_d - G_c_m)
        self.W = self.W + self.vW
        self.b = self.b + self.vb
        self.c = self.c + self.vc

    # ---- raw-sum form used by the data-parallel path -------------------------
    def local_sums(self, D, P_m, P_mp, v_m):
        """The un-normalised statistic sums one rank contributes to the
        all-reduce; ``finalize_from_sums`` turns the summed
        buffer into the update.  Layout documented in include/crbm_amd.h."""
        D = np.asarray(D, dtype=np.float64)
        ds = self.doublestranded
        out = {}

        def vh_raw(P, data):
            N, K, _, Lh = P.shape
            return self._collectVHStatistics(P, data)[:, 0] * (N * Lh)

        P_d = self._computeHgivenV(D)[0]
        out['vh_d'] = vh_raw(P_d, D)
        out['h_d'] = P_d.sum(axis=(0, 2, 3))
        dP = self._pooled_slope(P_d)
        out['sw'] = vh_raw(dP, D)
        out['sb'] = dP.sum(axis=(0, 2, 3))
        out['v_d'] = D.sum(axis=(0, 1, 3))
        out['vh_m'] = vh_raw(P_m, v_m)
        out['h_m'] = P_m.sum(axis=(0, 2, 3))
        out['v_m'] = v_m.sum(axis=(0, 1, 3))
        if ds:
            P_dp = self._computeHgivenV(D, True)[0]
            out['vh_dp'] = vh_raw(P_dp, D)
            out['h_dp'] = P_dp.sum(axis=(0, 2, 3))
            out['vh_mp'] = vh_raw(P_mp, v_m)
            out['h_mp'] = P_mp.sum(axis=(0, 2, 3))
        out['n_d'] = float(D.shape[0])
        out['n_m'] = float(P_m.shape[0])
        return out

    def finalize_from_sums(self, s, L_data, Lh_model):
        """Normalise globally-summed raw statistics and apply the update."""
        M, K = self.motif_length, self.num_motifs
        Lh_d = L_data - M + 1
        cnt_d = s['n_d'] * Lh_d
        cnt_m = s['n_m'] * Lh_model
        vh_d, h_d = s['vh_d'] / cnt_d, s['h_d'] / cnt_d
        vh_m, h_m = s['vh_m'] / cnt_m, s['h_m'] / cnt_m
        if self.doublestranded:
            vh_d = (vh_d + (s['vh_dp'] / cnt_d)[:, ::-1, ::-1]) / 2
            h_d = (h_d + s['h_dp'] / cnt_d) / 2
            vh_m = (vh_m + (s['vh_mp'] / cnt_m)[:, ::-1, ::-1]) / 2
            h_m = (h_m + s['h_mp'] / cnt_m) / 2
        v_d = s['v_d'] / (s['n_d'] * L_data)
        v_d = v_d + v_d[::-1]
        v_m = s['v_m'] / (s['n_m'] * (Lh_model + M - 1))
        v_m = v_m + v_m[::-1]
        p = s['h_d'] / cnt_d                     # forward strand only (:443)
        q = self.rho
        g = (q / p - (1 - q) / (1 - p)) / K
        reg_W = -(g[:, None, None] * s['sw'] / cnt_d)
        reg_b = -(g * s['sb'] / cnt_d)
        mu, alpha, sp = self.momentum, self.learning_rate, self.lambda_rate
        self.vW = mu * self.vW + alpha * (vh_d - vh_m - sp * reg_W)[:, None]
        self.vb = mu * self.vb + alpha * (h_d - h_m - sp * reg_b)[None, :]
        self.vc = mu * self.vc + alpha * (v_d - v_m)[None, :]
        self.W = self.W + self.vW
        self.b = self.b + self.vb
        self.c = self.c + self.vc

    # ---- evaluation --------------------------------------------------------
    def _freeEnergyForData(self, D):
        """convRBM.py:657-676 -> (N,), divided by L."""
        D = np.asarray(D, dtype=np.float64)
        pool = self.pooling

        def term(flip):
            x = self._bottomUpActivity(D, flip)
            N, K, one, Lh = x.shape
            x = x.reshape(N, K, one, Lh // pool, pool)
            if pool == 1:
                return -np.sum(_softplus(x[..., 0]), axis=(1, 2, 3))
            return -np.sum(np.log(1.0 + np.sum(np.exp(x), axis=4)), axis=(1, 2, 3))

        fe = term(False)
        if self.doublestranded:
            fe = fe + term(True)
        fe = fe - np.sum(D * self.c.reshape(1, 1, -1, 1), axis=(1, 2, 3))
        return fe / D.shape[3]

    def _freeEnergyPerMotif(self, D):
        """convRBM.py:678-697 -> (N,K), NOT divided by L."""
        D = np.asarray(D, dtype=np.float64)
        pool = self.pooling

        def term(flip):
            x = self._bottomUpActivity(D, flip)
            N, K, one, Lh = x.shape
            x = x.reshape(N, K, one, Lh // pool, pool)
            if pool == 1:
                return -np.sum(_softplus(x[..., 0]), axis=(2, 3))
            return -np.sum(np.log(1.0 + np.sum(np.exp(x), axis=4)), axis=(2, 3))

        fe = term(False)
        if self.doublestranded:
            fe = fe + term(True)
        return fe - np.sum(D * self.c.reshape(1, 1, -1, 1), axis=(1, 2, 3))[:, None]

    def freeEnergy(self, data, permotif=False):
        """convRBM.py:549-568."""
        return self._freeEnergyPerMotif(data) if permotif else self._freeEnergyForData(data)

    def _meanFreeEnergy(self, D):
        """convRBM.py:636-638."""
        return np.sum(self._freeEnergyForData(D)) / D.shape[0]

    def motifHitProbs(self, data):
        """convRBM.py:507-514: ds -> sigma(x); ss -> sigma(x + x')."""
        if self.doublestranded:
            return self._bottomUpProbability(self._bottomUpActivity(data))
        return self._bottomUpProbability(
            self._bottomUpActivity(data) + self._bottomUpActivity(data, True))

    def evaluateData(self, D, eval_step=0, seq_offset=0):
        """convRBM.py:466-472, :487-491 -> [mean free energy, mean sampled H]."""
        N = D.shape[0]
        P = self._computeHgivenV(D)[0]
        u = hidden_uniforms(self.seed, eval_step, np.arange(N) + seq_offset,
                            self.num_motifs, P.shape[3], 0, KIND_EVAL_H)
        H = self._bottomUpSample(P, u)
        return [self._meanFreeEnergy(D), float(np.mean(H))]

    def evaluateParams(self):
        """convRBM.py:475-485 -> [rms(W), IC, median IC]."""
        W = self.W
        twn = np.sqrt(np.mean(W ** 2))
        pwm = _softmax_axis2(W)
        ent = np.sum(-pwm * np.log2(pwm), axis=2)            # (K,1,M)
        A = W.shape[2]
        ic = np.log2(A) - np.mean(ent)
        medic = np.log2(A) - np.mean(np.sort(ent, axis=2)[:, :, ent.shape[2] // 2])
        return [float(twn), float(ic), float(medic)]

    def getPFMs(self):
        """convRBM.py:640-655."""
        out = []
        for m in self.W:
            e = np.exp(m[0])
            out.append(e / e.sum(axis=0, keepdims=True))
        return out

    @staticmethod
    def _iterateBatchIndices(totalsize, nbatchsize):
        """convRBM.py:722-726."""
        return [[i, i + nbatchsize] if i + nbatchsize <= totalsize else [i, totalsize]
                for i in range(totalsize)[0::nbatchsize]]
