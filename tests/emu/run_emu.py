"""TEST INFRASTRUCTURE: drives the CPU-thread emulation of the HIP kernels
(libcrbm_emu.so, built from tests/emu/emu_main.cpp with ASan+UBSan) and checks
every kernel against the float64 oracle.  Run by tests/test_emu.py in a
subprocess with the sanitizer runtime preloaded."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.crbm_oracle import (OracleCRBM, synthetic_onehot, hidden_uniforms, visible_uniforms,  # noqa: E402
                                KIND_API_H, KIND_API_V, KIND_CHAIN_H, KIND_CHAIN_V, letters_of)

lib = ctypes.CDLL(os.path.join(HERE, os.environ.get("CRBM_EMU_LIB", "libcrbm_emu.so")))
F = ctypes.POINTER(ctypes.c_float)
U = ctypes.POINTER(ctypes.c_uint32)


def fp(a):
    return None if a is None else a.ctypes.data_as(F)


def up(a):
    return None if a is None else a.ctypes.data_as(U)


def f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def model_arrays(o):
    return f32(o.W.reshape(o.num_motifs, o.input_dims, o.motif_length)), f32(o.b.ravel()), f32(o.c.ravel())


def case_info(cid):
    out = (ctypes.c_int * 8)()
    assert lib.emu_case_info(cid, out) == 0
    K, M, DS, G, TABLES, NW, DENSE, POOL = list(out)
    return dict(K=K, M=M, ds=bool(DS), G=G, TABLES=TABLES, NW=NW, DENSE=DENSE, POOL=POOL)


def build_tables(cid, o):
    info = case_info(cid)
    W, b, c = model_arrays(o)
    t = np.zeros(info["TABLES"], dtype=np.float32)
    assert lib.emu_tables(cid, fp(W), fp(b), fp(c), fp(t)) == 0
    return t


def encode(data):
    n, A, L = data.shape[0], data.shape[2], data.shape[3]
    LW = lib.emu_letter_words_any(A, L)
    letters = np.zeros((n, LW), dtype=np.uint32)
    flags = np.zeros(4, dtype=np.uint32)
    if A == 4:
        lib.emu_encode(fp(f32(data)), up(letters), up(flags), n, L, 3)
    else:                                   # any other alphabet: rows of bytes
        lib.emu_encode_any(fp(f32(data)), None, up(letters), up(flags), n, L, A, 3)
    return letters, int(flags[0])


def pack_hidden(h, NW):
    n, K, _, Lh = h.shape
    masks = np.zeros((n, Lh, NW), dtype=np.uint32)
    flags = np.zeros(4, dtype=np.uint32)
    d = f32(h)
    lib.emu_pack_hidden(fp(d), up(masks), up(flags), n, K, Lh, NW, 0)
    return masks, int(flags[0])


def unpack_hidden(masks, K):
    n, Lh, NW = masks.shape
    d = np.zeros((n, K, 1, Lh), dtype=np.float32)
    flags = np.zeros(4, dtype=np.uint32)
    lib.emu_pack_hidden(fp(d), up(np.ascontiguousarray(masks)), up(flags), n, K, Lh, NW, 1)
    return d


def check_samples(name, got, prob, u):
    """identical up to ties: a mismatch needs |p-u| < 1e-6."""
    want = (prob > u).astype(np.float32)
    bad = got != want
    if bad.any():
        assert np.all(np.abs(prob - u)[bad] < 1e-6), "%s: sample mismatch away from a tie" % name
    return int(bad.sum())


def close_vh(got, want, rtol=2e-5):
    """A (K,4,M) block of raw VH sums against the oracle's.  Where the fourth letter is derived as
    H - (the other three) (models whose motif length is not a multiple of 16: crbm_layout.h, NL) its absolute
    error follows the scale of H = sum_a VH[k,a,j], not its own: 4e-7 of the largest H is allowed."""
    want = np.asarray(want, dtype=np.float64).ravel()
    np.testing.assert_allclose(got, want, rtol=rtol, atol=1e-6 + 4e-7 * 4 * float(np.abs(want).max()))


def make_oracle(K, M, ds, seed=0, batch=4, Lf=20, cd_k=2, wscale=1.0, A=4, **kw):
    rng = np.random.default_rng(100 + K * 31 + M)
    o = OracleCRBM(K, M, doublestranded=ds, batchsize=batch, cd_k=cd_k, fantasy_hidden_len=Lf, seed=seed, input_dims=A,
                   W=rng.standard_normal((K, 1, A, M)) * wscale, **kw)
    o.c = f32(rng.standard_normal((1, A)) * 0.2).astype(np.float64)
    o.b = f32(o.b + 6.0 + rng.standard_normal((1, K)) * 0.5).astype(np.float64)   # livelier hidden units
    return o


# case ids of the Cfg<K,M,DS,G> instantiations in emu_main.cpp:
# 0 (10,5,ss,G2) 1 (10,15,ds,G3) 2 (2,5,ds,G4) 3 (3,4,ss,G1) 4 (20,15,ds,G2: sparse v|h)
# 5 (50,25,ss,G2: two mask words) 6 (7,32,ds,G3: sparse, 96-bit letter window) 7 (10,15,ss,G3: config #2)
# 8 (64,32,ds,G1: the largest model the kernels take; statistics only -- three column roles)
# 9 (4,5,ds,G2,pooling 2) 10 (10,15,ss,G3,pooling 3): pooled hidden units (convRBM.py:245-267)
# 11 (100,15,ss,G1: four mask words) 12 (20,40,ds,G2: two-word letter windows, three filter-column tiles)
# 13 (70,33,ds,G1: three mask words -- 4-byte aligned masks -- and two-word windows)
LARGE_CASES = [11, 12, 13]
ALL_CASES = list(range(8))
POOLED_CASES = [9, 10]
# One OS thread per GPU thread makes barrier-heavy kernels slow on 8 cores: the
# default run keeps the CPU suite to a few minutes, CRBM_EMU_FULL=1 runs all.
FULL = os.environ.get("CRBM_EMU_FULL", "0") == "1"
CASES = ALL_CASES if FULL else [1, 2, 3]
GIBBS_CASES = ALL_CASES if FULL else [1, 5, 7]
TRAIN_CASES = ALL_CASES[:6] if FULL else [1, 2, 5]


def oracle_for(cid, **kw):
    info = case_info(cid)
    return info, make_oracle(info["K"], info["M"], info["ds"], pooling=info["POOL"], **kw)


def test_encode_pack():
    d = synthetic_onehot(5, 37, seed=1)
    letters, flag = encode(d)
    assert flag == 0
    LW = lib.emu_letter_words(37)
    ref = letters_of(d)
    for n in range(5):
        for p in range(37):
            assert (int(letters[n, p >> 4]) >> (2 * (p & 15))) & 3 == ref[n, p]
    assert np.all(letters[:, LW - 2:] == 0)
    back = np.zeros_like(d)
    lib.emu_decode(up(letters), fp(back), 5, 37, LW, 2)
    assert np.array_equal(back, d)
    bad = d.copy(); bad[2, 0, :, 5] = 0
    assert encode(bad)[1] == 1
    bad = d.copy(); bad[1, 0, :, 7] = 0.5
    assert encode(bad)[1] == 1
    rng = np.random.default_rng(2)
    for K, NW in ((10, 1), (40, 2)):
        h = rng.binomial(1, 0.3, size=(3, K, 1, 9)).astype(np.float32)
        m, fl = pack_hidden(h, NW)
        assert fl == 0
        assert np.array_equal(unpack_hidden(m, K), h)
        h2 = h.copy(); h2[0, 0, 0, 0] = 0.25
        assert pack_hidden(h2, NW)[1] == 2
    # reduce_partials_kernel on a synthetic K=1, M=1 layout (KAM = 4): row = 3*4 + 3 + 4 = 19 columns
    # [vh 4][vh' 4][h 1][h' 1][sw 4][sb 1][v 4]; 300 rows exercise the 4-rows-in-flight loop and its tail
    nrows, row = 300, 19
    part = rng.standard_normal((nrows, row)).astype(np.float32)
    for ds_, want_ in ((1, 1), (0, 0)):
        skip_b, skip_l = (10, 5) if not want_ else (row, 0)      # model half drops sw, sb
        sums = np.full(row - skip_l + 1, -1.0, dtype=np.float32)
        lib.emu_reduce(fp(part), fp(sums), nrows, row, 1, 4, ds_, want_, skip_b, skip_l, ctypes.c_float(42.0))
        full = part.astype(np.float64).sum(axis=0)
        exp = np.concatenate([full[0:4], full[4:8] * ds_, full[8:9], full[9:10] * ds_, full[10:14] * want_,
                              full[14:15] * want_, full[15:19]])
        keep = [i for i in range(row) if not (skip_b <= i < skip_b + skip_l)]
        np.testing.assert_allclose(sums[:len(keep)], exp[keep], rtol=1e-5, atol=1e-5)
        assert sums[len(keep)] == 42.0
    print("encode/pack/reduce ok")


def test_hgv(cases=None):
    for cid in (cases or CASES):
        info, o = oracle_for(cid)
        K, M, ds = info["K"], info["M"], info["ds"]
        tables = build_tables(cid, o)
        n, L = 5, M + 27
        d = synthetic_onehot(n, L, seed=K)
        letters, _ = encode(d)
        Lh = L - M + 1
        for mode in (0, 1, 2):
            act = np.zeros((n, K, 1, Lh), dtype=np.float32)
            prob = np.zeros_like(act)
            smp = np.zeros_like(act)
            ones = ctypes.c_ulonglong(0)
            rc = lib.emu_hgv(cid, fp(tables), up(letters), n, L, mode, fp(act), fp(prob),
                             fp(smp), ctypes.byref(ones), ctypes.c_uint64(77), 5, 3, KIND_API_H, 2, 2, 128)
            assert rc == 0
            if mode == 2:
                ref = o._bottomUpActivity(d) + o._bottomUpActivity(d, True)
            else:
                ref = o._bottomUpActivity(d, mode == 1)
            np.testing.assert_allclose(act, ref, rtol=1e-5, atol=1e-5)
            p = 1 / (1 + np.exp(-ref))
            np.testing.assert_allclose(prob, p, rtol=1e-5, atol=1e-6)
            u = hidden_uniforms(77, 5, np.arange(n) + 3, K, Lh, 1 if mode == 1 else 0, KIND_API_H)
            ties = check_samples("hgv", smp, prob.astype(np.float64), u)
            assert ones.value == int(smp.sum())
            # probabilities only (no sampling path)
            prob2 = np.zeros_like(act)
            lib.emu_hgv(cid, fp(tables), up(letters), n, L, mode, None, fp(prob2), None, None,
                        ctypes.c_uint64(77), 5, 3, KIND_API_H, 3, 1, 64)
            np.testing.assert_allclose(prob2, p, rtol=1e-5, atol=1e-6)
        print("hgv ok", cid, (K, M, ds), "ties", ties)


def test_vgh():
    for cid in CASES:
        info, o = oracle_for(cid)
        K, M, ds = info["K"], info["M"], info["ds"]
        W, b, c = model_arrays(o)
        rng = np.random.default_rng(K)
        n, Lh = 4, 23
        h = rng.binomial(1, 0.15, size=(n, K, 1, Lh)).astype(np.float32)
        hp = rng.binomial(1, 0.15, size=(n, K, 1, Lh)).astype(np.float32) if ds else None
        if K == 3:
            h = rng.standard_normal(h.shape).astype(np.float32)      # any finite values
        L = Lh + M - 1
        act = np.zeros((n, 1, 4, L), dtype=np.float32)
        prob = np.zeros_like(act)
        smp = np.zeros_like(act)
        lib.emu_vgh(fp(W), fp(c), K, M, fp(h), fp(hp), n, Lh, fp(act), fp(prob), fp(smp),
                    ctypes.c_uint64(9), 2, 1, 2, 3, 64)
        ref = o._topDownActivity(h, hp)
        np.testing.assert_allclose(act, ref, rtol=1e-5, atol=1e-5)
        pref = o._topDownProbability(ref)
        np.testing.assert_allclose(prob, pref, rtol=1e-5, atol=1e-6)
        u = visible_uniforms(9, 2, np.arange(n) + 1, L, KIND_API_V)
        sref = o._topDownSample(pref, u)
        bad = (smp != sref).any(axis=2)[:, 0]
        if bad.any():
            cum = np.cumsum(pref[:, 0], axis=1)
            gap = np.min(np.abs(cum - u[:, None, :]), axis=1)
            assert np.all(gap[bad] < 1e-6)
        np.testing.assert_array_equal(smp.sum(axis=2), 1.0)
        print("vgh ok", (K, M, ds))


def run_gibbs(cid, o, tables, S, steps, grid, threads, sparse=1, ones=None):
    info = case_info(cid)
    K, M, ds, NW = info["K"], info["M"], info["ds"], info["NW"]
    hm, _ = pack_hidden(f32(o.fantasy_h), NW)
    hmp = pack_hidden(f32(o.fantasy_h_prime), NW)[0] if ds else np.zeros_like(hm)
    B, Lf = o.fantasy_h.shape[0], o.fantasy_h.shape[3]
    Lv = Lf + M - 1
    lws = lib.emu_gibbs(cid, fp(tables), up(hm), up(hmp), None, B, Lf, S, steps,
                        ctypes.c_uint64(o.seed), o.gibbs_step, o.seq_offset, grid, threads, sparse, None)
    assert lws > 0
    vout = np.zeros((B, lws), dtype=np.uint32)
    rc = lib.emu_gibbs(cid, fp(tables), up(hm), up(hmp), up(vout), B, Lf, S, steps,
                       ctypes.c_uint64(o.seed), o.gibbs_step, o.seq_offset, grid, threads, sparse,
                       up(ones) if ones is not None else None)
    assert rc == lws
    v = np.zeros((B, 1, 4, Lv), dtype=np.float32)
    lib.emu_decode(up(vout), fp(v), B, Lv, lws, 2)
    return unpack_hidden(hm, K), (unpack_hidden(hmp, K) if ds else None), v, vout, lws


TIE = 1e-6


def replay_chain_ties(cid, o, tables, S, steps, grid, threads, sparse):
    """`steps` Gibbs steps one at a time, the emulated kernel restarted from the oracle's state before each: every letter
    and every hidden unit that differs from the oracle's must sit on a p == u tie (|p - u| < 1e-6: the only place where
    the kernel's float32 and the oracle's float64 may decide differently).  Returns the number of ties met."""
    ds = o.doublestranded
    B, K, _, Lf = o.fantasy_h.shape
    Lv = Lf + o.motif_length - 1
    idx = np.arange(B) + o.seq_offset
    ties = 0
    for _ in range(steps):
        t = o.gibbs_step
        gh, ghp, gv, _, _ = run_gibbs(cid, o, tables, S, 1, grid, threads, sparse)
        uv = visible_uniforms(o.seed, t, idx, Lv, KIND_CHAIN_V)
        Pv, v = o._computeVgivenH(o.fantasy_h, o.fantasy_h_prime if ds else None, uv)
        uh = hidden_uniforms(o.seed, t, idx, K, Lf, 0, KIND_CHAIN_H)
        P, h = o._computeHgivenV(v, False, uh)
        pairs = [(gh, h, P, uh)]
        if ds:
            uhp = hidden_uniforms(o.seed, t, idx, K, Lf, 1, KIND_CHAIN_H)
            Pp, hp = o._computeHgivenV(v, True, uhp)
            pairs.append((ghp, hp, Pp, uhp))
        badv = (gv != v).any(axis=2)[:, 0]
        if badv.any():
            cum = np.cumsum(Pv[:, 0], axis=1)[:, :3]
            gap = np.min(np.abs(cum - uv[:, None, :]), axis=1)
            assert np.all(gap[badv] < TIE), "visible sample differs away from a tie"
            ties += int(badv.sum())
        clean = ~badv.any(axis=1)
        for got, want, prob, u in pairs:
            bad = (got != want) & clean[:, None, None, None]
            if bad.any():
                assert np.all(np.abs(prob - u)[bad] < TIE), "hidden sample differs away from a tie"
                ties += int(bad.sum())
        o.fantasy_h, o.fantasy_h_prime = h, (hp if ds else o.fantasy_h_prime)
        o.last_v_model = v
        o.gibbs_step += 1
    return ties


def test_gibbs(cases=None):
    for cid in (cases or GIBBS_CASES):
        info = case_info(cid)
        for sparse in ((1, 0) if info["DENSE"] else (1,)):     # both top-down variants where both exist
            for (B, Lf, S, steps, grid, threads) in ((5, 21, 2, 3, 2, 128), (3, 40, 4, 1, 1, 64)):
                info, o = oracle_for(cid, seed=11, batch=B, Lf=Lf, wscale=1.5)
                ds = info["ds"]
                o.seq_offset = 6
                rng = np.random.default_rng(5)
                o.fantasy_h = rng.binomial(1, 0.1, size=o.fantasy_h.shape).astype(np.float64)
                if ds:
                    o.fantasy_h_prime = rng.binomial(1, 0.1, size=o.fantasy_h.shape).astype(np.float64)
                tables = build_tables(cid, o)
                ones = np.zeros(grid * (threads // 64), dtype=np.uint32)
                start = (o.fantasy_h.copy(), o.fantasy_h_prime.copy() if ds else None, o.gibbs_step)
                h, hp, v, _, _ = run_gibbs(cid, o, tables, S, steps, grid, threads, sparse, ones)
                o.gibbs_steps(steps)
                mism = int((h != o.fantasy_h).sum()) + (int((hp != o.fantasy_h_prime).sum()) if ds else 0)
                mism_v = int((v != o.last_v_model).sum())
                if mism or mism_v:
                    # a chain may leave the oracle's only through a p == u tie: replay step by step from identical states
                    o.fantasy_h, o.gibbs_step = start[0], start[2]
                    if ds:
                        o.fantasy_h_prime = start[1]
                    ties = replay_chain_ties(cid, o, tables, S, steps, grid, threads, sparse)
                    assert 1 <= ties <= 3, ("gibbs mismatch without a tie", cid, sparse, mism, mism_v, ties)
                    print("gibbs: %d tie(s) in case %d" % (ties, cid))
                assert o.fantasy_h.sum() > 0
                assert int(ones.sum()) == int(h.sum()) + (int(hp.sum()) if ds else 0), "activity monitor"
        print("gibbs ok", cid, info)


def test_train_step(cases=None):
    """One PCD-k update through the kernels of the product path: data half = stats_mfma_body (SP),
    model half = the fused tail of the Gibbs kernel where the model has one (else Gibbs kernel +
    stand-alone stats_mfma_body on its visible sample), then apply_update; all against the oracle."""
    for cid in (cases or TRAIN_CASES):
        info = case_info(cid)
        K, M, ds, NW = info["K"], info["M"], info["ds"], info["NW"]
        B, Lf, n, L = 4, 18 if cid != 1 else 40, 5, M + 20
        o = make_oracle(K, M, ds, seed=3, batch=B, Lf=Lf, cd_k=2, rho=0.05)
        o.vW = f32(np.random.default_rng(1).standard_normal(o.W.shape) * 0.01).astype(np.float64)
        D = synthetic_onehot(n, L, seed=21)
        W, b, c = model_arrays(o)
        vW, vb, vc = f32(o.vW.reshape(K, 4, M)), f32(o.vb.ravel()), f32(o.vc.ravel())
        lay = (ctypes.c_int * 7)()
        lib.emu_sums_layout(K, M, lay)
        data_off, n_d, model_off, n_m, count, skipb, skipl = list(lay)
        sums = np.zeros(count, dtype=np.float32)
        letters, _ = encode(D)
        threads = 128
        row = 3 * K * 4 * M + 3 * K + 4
        partials = np.zeros(32 * row, dtype=np.float32)
        tables = build_tables(cid, o)
        r = lib.emu_stats_mfma(cid, fp(tables), up(letters), n, L, lib.emu_letter_words(L), 1, 0, 3,
                               fp(partials), partials.size, fp(sums[data_off:]), -1, 0)
        assert r == row, r
        Lv = Lf + M - 1
        # fused: Gibbs steps + model statistics in one launch
        hm, _ = pack_hidden(f32(o.fantasy_h), NW)
        hmp = pack_hidden(f32(o.fantasy_h_prime), NW)[0] if ds else np.zeros_like(hm)
        lws = lib.emu_gibbs(cid, fp(tables), up(hm), up(hmp), None, B, Lf, 2, o.cd_k,
                            ctypes.c_uint64(o.seed), o.gibbs_step, o.seq_offset, 2, threads, 1, None)
        vout = np.zeros((B, lws), dtype=np.uint32)
        r = lib.emu_gibbs_stats(cid, fp(tables), up(hm), up(hmp), up(vout), B, Lf, 2, o.cd_k,
                                ctypes.c_uint64(o.seed), o.gibbs_step, o.seq_offset, 2, threads,
                                fp(partials), partials.size, fp(sums[model_off:]), skipb, skipl)
        fused = r != -3
        if fused:
            assert r == row, r
            # ... and the whole local phase in ONE launch (chain + model half and data half interleaved)
            # must give the same chains and the same sums as the two launches above
            hm1, _ = pack_hidden(f32(o.fantasy_h), NW)
            hmp1 = pack_hidden(f32(o.fantasy_h_prime), NW)[0] if ds else np.zeros_like(hm1)
            vout1 = np.zeros((B, lws), dtype=np.uint32)
            sums1 = np.zeros(count, dtype=np.float32)
            pm, pd = np.zeros(32 * row, dtype=np.float32), np.zeros(32 * row, dtype=np.float32)
            r1 = lib.emu_train_local(cid, fp(tables), up(hm1), up(hmp1), up(vout1), B, Lf, 2, o.cd_k,
                                     ctypes.c_uint64(o.seed), o.gibbs_step, o.seq_offset, 2, 3, threads,
                                     up(letters), n, L, lib.emu_letter_words(L), fp(pm), fp(pd), pm.size,
                                     fp(sums1[data_off:]), fp(sums1[model_off:]), skipb, skipl)
            assert r1 == row, r1
            assert np.array_equal(hm1, hm) and np.array_equal(hmp1, hmp) and np.array_equal(vout1, vout)
            np.testing.assert_allclose(sums1[data_off:n_d], sums[data_off:n_d], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(sums1[model_off:n_m], sums[model_off:n_m], rtol=1e-5, atol=1e-6)
            assert sums1[n_d] == n and sums1[n_m] == B
            v = np.zeros((B, 1, 4, Lv), dtype=np.float32)
            lib.emu_decode(up(vout), fp(v), B, Lv, lws, 2)
            h_after = unpack_hidden(hm, K)
        else:
            h_after, _, v, vout, lws = run_gibbs(cid, o, tables, 2, o.cd_k, 2, threads)
            r = lib.emu_stats_mfma(cid, fp(tables), up(vout), B, Lv, lws, 0, 0, 2,
                                   fp(partials), partials.size, fp(sums[model_off:]), skipb, skipl)
            assert r == row, r
        assert sums[n_d] == n and sums[n_m] == B
        # raw sums against the oracle's
        P_m, P_mp, v_m = o.gibbs_steps(o.cd_k)
        assert np.array_equal(v, v_m)
        assert np.array_equal(h_after, o.fantasy_h)
        o.gibbs_step -= o.cd_k
        s = o.local_sums(D, P_m, P_mp, v_m)
        KAM = K * 4 * M
        close_vh(sums[0:KAM], s['vh_d'])
        np.testing.assert_allclose(sums[2 * KAM:2 * KAM + K], s['h_d'], rtol=2e-5)
        close_vh(sums[2 * KAM + 2 * K:3 * KAM + 2 * K], s['sw'])
        np.testing.assert_allclose(sums[3 * KAM + 2 * K:3 * KAM + 3 * K], s['sb'], rtol=2e-5)
        np.testing.assert_allclose(sums[3 * KAM + 3 * K:3 * KAM + 3 * K + 4], s['v_d'])
        close_vh(sums[model_off:model_off + KAM], s['vh_m'])
        np.testing.assert_allclose(sums[model_off + 2 * KAM:model_off + 2 * KAM + K], s['h_m'], rtol=2e-5)
        np.testing.assert_allclose(sums[n_m - 4:n_m], s['v_m'])
        if ds:
            close_vh(sums[KAM:2 * KAM], s['vh_dp'])
            close_vh(sums[model_off + KAM:model_off + 2 * KAM], s['vh_mp'])
            np.testing.assert_allclose(sums[model_off + 2 * KAM + K:model_off + 2 * KAM + 2 * K], s['h_mp'], rtol=2e-5)
        new_tables = np.zeros(info["TABLES"], dtype=np.float32)
        old = [x.copy() for x in (W, b, c, vW, vb, vc)]
        nxt = [np.full_like(x, np.nan) for x in old]
        lib.emu_update_tables(cid, fp(sums), *[fp(x) for x in old], *[fp(x) for x in nxt], L, Lf,
                              ctypes.c_float(o.learning_rate), ctypes.c_float(o.momentum), ctypes.c_float(o.rho),
                              ctypes.c_float(o.lambda_rate), fp(new_tables), 3, 128)
        for x, keep in zip(old, (W, b, c, vW, vb, vc)):
            assert np.array_equal(x, keep)               # the blocks still reading the old set see it untouched
        W, b, c, vW, vb, vc = nxt
        # the table images the fused launch leaves are those of the NEW parameters
        ref_tables = np.zeros(info["TABLES"], dtype=np.float32)
        assert lib.emu_tables(cid, fp(W), fp(b), fp(c), fp(ref_tables)) == 0
        np.testing.assert_array_equal(new_tables, ref_tables)
        o.finalize_from_sums(s, L, Lf)
        np.testing.assert_allclose(W.reshape(o.W.shape), o.W, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(b, o.b.ravel(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(c, o.c.ravel(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(vW.reshape(o.W.shape), o.vW, rtol=1e-4, atol=1e-7)
        print("train step ok", cid, (K, M, ds), "fused" if fused else "split")


def rank_step(cid, o, tables, D_rows, L, chain_lo, chain_hi, threads=128):
    """The local phase of one rank of a data-parallel step through the product kernels: packed raw
    sums (include/crbm_amd.h layout) of its data rows and of its chains, which it advances by cd_k
    Gibbs steps.  Returns (sums, hm, hmp) with the new chain masks."""
    info = case_info(cid)
    K, M, ds, NW = info["K"], info["M"], info["ds"], info["NW"]
    lay = (ctypes.c_int * 7)()
    lib.emu_sums_layout(K, M, lay)
    data_off, n_d, model_off, n_m, count, skipb, skipl = list(lay)
    sums = np.zeros(count, dtype=np.float32)
    row = 3 * K * 4 * M + 3 * K + 4
    partials = np.zeros(16 * row, dtype=np.float32)
    n = D_rows.shape[0]
    if n:
        letters, _ = encode(D_rows)
        r = lib.emu_stats_mfma(cid, fp(tables), up(letters), n, L, lib.emu_letter_words(L), 1, 0, 2,
                               fp(partials), partials.size, fp(sums[data_off:]), -1, 0)
        assert r == row
    sums[n_d] = n                                       # a rank may own no rows: zeros and n_d = 0
    B = chain_hi - chain_lo
    Lf = o.fantasy_h.shape[3]
    hm, _ = pack_hidden(f32(o.fantasy_h[chain_lo:chain_hi]), NW)
    hmp = pack_hidden(f32(o.fantasy_h_prime[chain_lo:chain_hi]), NW)[0] if ds else np.zeros_like(hm)
    lws = lib.emu_gibbs(cid, fp(tables), up(hm), up(hmp), None, B, Lf, 2, o.cd_k,
                        ctypes.c_uint64(o.seed), o.gibbs_step, chain_lo, 2, threads, 1, None)
    vout = np.zeros((B, lws), dtype=np.uint32)
    r = lib.emu_gibbs_stats(cid, fp(tables), up(hm), up(hmp), up(vout), B, Lf, 2, o.cd_k,
                            ctypes.c_uint64(o.seed), o.gibbs_step, chain_lo, 2, threads,
                            fp(partials), partials.size, fp(sums[model_off:]), skipb, skipl)
    assert r == row, r
    return sums, hm, hmp


def test_two_ranks():
    """SURVEY 8(e) in the GPU-less container with the PRODUCT kernels in the kernel seat: two ranks
    shard the rows of each mini-batch (crbm_amd.dist.shard_range, the arithmetic of
    crbm_train_epoch_resident) and the chains (Philox counters keyed by the global chain index), sum
    their packed buffers (what ncclAllReduce does) and apply the same update; the result must equal
    the one-rank step on the same rows: identical chains, parameters to reduction-order tolerance.
    The second mini-batch has a single row, so rank 0 owns none of it."""
    from crbm_amd.dist import shard_range
    cid = 1                                             # (10, 15, doublestranded): has the fused Gibbs variant
    info = case_info(cid)
    K, M, ds, NW = info["K"], info["M"], info["ds"], info["NW"]
    B, Lf, L = 4, 40, M + 20
    D = synthetic_onehot(6, L, seed=33)
    batches = [(0, 5), (5, 6)]

    def run(world, ipc=False):
        o = make_oracle(K, M, ds, seed=5, batch=B, Lf=Lf, cd_k=2, rho=0.05)
        step_no = 0
        W, b, c = model_arrays(o)
        vW, vb, vc = np.zeros_like(W), np.zeros_like(b), np.zeros_like(c)
        fh, fhp = o.fantasy_h.copy(), o.fantasy_h_prime.copy()
        for (s0, s1) in batches:
            o.W, o.b, o.c = W.reshape(o.W.shape).astype(np.float64), b.reshape(o.b.shape).astype(np.float64), c.reshape(o.c.shape).astype(np.float64)
            o.fantasy_h, o.fantasy_h_prime = fh, fhp
            tables = build_tables(cid, o)
            total = None
            new_h, new_hp = [], []
            per_rank = []
            for r in range(world):
                lo, hi = shard_range(s1 - s0, r, world)
                clo, chi = shard_range(B, r, world)
                sums, hm, hmp = rank_step(cid, o, tables, D[s0 + lo:s0 + hi], L, clo, chi)
                per_rank.append(sums)
                total = sums if total is None else (total + sums).astype(np.float32)
                new_h.append(unpack_hidden(hm, K))
                new_hp.append(unpack_hidden(hmp, K))
            fh, fhp = np.concatenate(new_h).astype(np.float64), np.concatenate(new_hp).astype(np.float64)
            o.gibbs_step += o.cd_k
            old = [W, b, c, vW, vb, vc]
            nxt = [np.zeros_like(x) for x in old]
            new_tables = np.zeros(info["TABLES"], dtype=np.float32)
            if ipc:
                # the all-reduce through mapped buffers: every rank publishes, the update launch adds the copies in rank order
                step_no += 1
                count = per_rank[0].size
                stride = (count + 31) & ~31
                bufs = np.full(world * stride, np.nan, dtype=np.float32)
                flags, status = np.zeros(world, dtype=np.uint32), np.zeros(1, dtype=np.uint32)
                rc = lib.emu_ipc_update(cid, world, fp(np.concatenate(per_rank)), count, fp(bufs), stride, up(flags), up(status), step_no,
                                        *[fp(x) for x in old], *[fp(x) for x in nxt], L, Lf,
                                        ctypes.c_float(o.learning_rate), ctypes.c_float(o.momentum), ctypes.c_float(o.rho),
                                        ctypes.c_float(o.lambda_rate), fp(new_tables), 2, 128, -1)
                assert rc == 0 and status[0] == 0 and np.all(flags == step_no)
                if step_no == 1:
                    # a peer that never publishes: the wait runs out, the sticky status word says so, nothing hangs
                    flags2, status2 = np.zeros(world, dtype=np.uint32), np.zeros(1, dtype=np.uint32)
                    scratch = [np.zeros_like(x) for x in old]
                    rc = lib.emu_ipc_update(cid, world, fp(np.concatenate(per_rank)), count, fp(bufs.copy()), stride, up(flags2), up(status2),
                                            step_no, *[fp(x) for x in old], *[fp(x) for x in scratch], L, Lf,
                                            ctypes.c_float(o.learning_rate), ctypes.c_float(o.momentum), ctypes.c_float(o.rho),
                                            ctypes.c_float(o.lambda_rate), fp(np.zeros_like(new_tables)), 1, 64, 1)
                    assert rc == 0 and status2[0] == 1 and flags2[1] == 0
                    for x, y in zip(old, scratch):        # ... and no update is applied: the state is carried over unchanged
                        np.testing.assert_array_equal(x, y)
                    # the status word is sticky: a later launch whose peers all delivered still applies nothing
                    scratch = [np.full_like(x, 7.0) for x in old]
                    rc = lib.emu_ipc_update(cid, world, fp(np.concatenate(per_rank)), count, fp(bufs.copy()), stride, up(flags2), up(status2),
                                            step_no, *[fp(x) for x in old], *[fp(x) for x in scratch], L, Lf,
                                            ctypes.c_float(o.learning_rate), ctypes.c_float(o.momentum), ctypes.c_float(o.rho),
                                            ctypes.c_float(o.lambda_rate), fp(np.zeros_like(new_tables)), 2, 128, -1)
                    assert rc == 0 and status2[0] == 1
                    for x, y in zip(old, scratch):
                        np.testing.assert_array_equal(x, y)
            else:
                lib.emu_update_tables(cid, fp(total), *[fp(x) for x in old], *[fp(x) for x in nxt], L, Lf,
                                      ctypes.c_float(o.learning_rate), ctypes.c_float(o.momentum), ctypes.c_float(o.rho),
                                      ctypes.c_float(o.lambda_rate), fp(new_tables), 2, 128)
            W, b, c, vW, vb, vc = nxt
        return W, b, c, vW, fh, fhp

    one, two = run(1), run(2)
    two_ipc = run(2, ipc=True)                          # ... and through publish + update_tables_ipc_body: bit for bit the same
    for x, y in zip(two, two_ipc):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(one[4], two[4])       # identical samples
    np.testing.assert_array_equal(one[5], two[5])
    for x, y in zip(one[:4], two[:4]):
        np.testing.assert_allclose(x, y, rtol=1e-5, atol=1e-7)
    assert not np.array_equal(one[0], model_arrays(make_oracle(K, M, ds, seed=5, batch=B, Lf=Lf, cd_k=2, rho=0.05))[0])
    print("two ranks ok")


def stats_sums(cid, tables, letters, n, L, want_sparsity, threads, gx, skip=(-1, 0)):
    """raw sums of one half through stats_mfma_body (+ the host column reduce of the harness)"""
    info = case_info(cid)
    K, M = info["K"], info["M"]
    row = 3 * K * 4 * M + 3 * K + 4
    partials = np.zeros(max(gx, 1) * row, dtype=np.float32)
    sums = np.zeros(row + 1, dtype=np.float32)
    r = lib.emu_stats_mfma(cid, fp(tables), up(letters), n, L, lib.emu_letter_words(L), want_sparsity, threads, gx,
                           fp(partials), partials.size, fp(sums), skip[0], skip[1])
    assert r == row, r
    return sums


def test_stats_mfma(cases=None):
    """The MFMA statistics kernel against the oracle's raw sums: units that span chains, chains
    shorter than a group, a ragged last group, an odd number of groups, several blocks, one to
    four waves per role."""
    cases = cases or ((ALL_CASES if FULL else [1, 3, 5]) + [8])
    for cid in cases:
        info = case_info(cid)
        K, M, ds = info["K"], info["M"], info["ds"]
        KAM = K * 4 * M
        # (n, L, waves per role, blocks); NR roles -> 64 * NR * waves threads
        shapes = ((5, M + 20, 2, 3), (3, M + 69, 4, 2), (3, M + 3, 1, 1))
        if cid == 8 or cid in LARGE_CASES:
            shapes = ((2, M + 40, 0, 2),)              # default geometry
        for (n, L, wpr, gx) in shapes:
            o = make_oracle(K, M, ds, seed=3, batch=2, Lf=10, rho=0.05)
            D = synthetic_onehot(n, L, seed=21 + n)
            letters, _ = encode(D)
            tables = build_tables(cid, o)
            Lh = L - M + 1
            # oracle raw sums of the data half
            P = o._bottomUpProbability(o._bottomUpActivity(D))
            Pp = o._bottomUpProbability(o._bottomUpActivity(D, True)) if ds else None
            Dv = D[:, 0].astype(np.float64)                       # (n,4,L)
            def vh(Pk):
                out = np.zeros((K, 4, M))
                for j in range(M):
                    out[:, :, j] = np.einsum('nks,nas->ka', Pk[:, :, 0, :], Dv[:, :, j:j + Lh])
                return out
            for want in (1, 0):
                NT, JT = -(-K // 16), -(-M // 16)
                kinds = 1 + int(ds) + want
                ntw = NT
                while ntw > 1 and (4 * JT * kinds * ntw > 32 or NT % ntw):
                    ntw -= 1
                threads = 64 * (NT // ntw) * wpr
                s = stats_sums(cid, tables, letters, n, L, want, threads, gx)
                close_vh(s[0:KAM], vh(P))
                np.testing.assert_allclose(s[2 * KAM:2 * KAM + K], P.sum(axis=(0, 2, 3)), rtol=2e-5)
                if ds:
                    close_vh(s[KAM:2 * KAM], vh(Pp))
                    np.testing.assert_allclose(s[2 * KAM + K:2 * KAM + 2 * K], Pp.sum(axis=(0, 2, 3)), rtol=2e-5)
                if want:
                    Q = P * (1 - P)
                    close_vh(s[2 * KAM + 2 * K:3 * KAM + 2 * K], vh(Q))
                    np.testing.assert_allclose(s[3 * KAM + 2 * K:3 * KAM + 3 * K], Q.sum(axis=(0, 2, 3)), rtol=2e-5)
                np.testing.assert_array_equal(s[3 * KAM + 3 * K:3 * KAM + 3 * K + 4], Dv.sum(axis=(0, 2)))
                assert s[3 * KAM + 3 * K + 4] == n
        print("stats mfma ok", cid, (K, M, ds))


def check_pooled_samples(name, got, prob, u, pool):
    """a pooled sample may differ from the oracle's only where the group's uniform sits on a cumulative-sum tie"""
    N, K, one, Lh = prob.shape
    p = prob.reshape(N, K, one, Lh // pool, pool)
    ug = u.reshape(N, K, one, Lh // pool, pool)[..., :1]
    cum = np.cumsum(p, axis=4)
    want = ((cum > ug) & (np.concatenate([np.zeros_like(cum[..., :1]), cum[..., :-1]], axis=4) <= ug)).astype(np.float32)
    g = got.reshape(N, K, one, Lh // pool, pool)
    bad = (g != want).any(axis=4)
    if bad.any():
        gap = np.min(np.abs(cum - ug), axis=4)
        assert np.all(gap[bad] < 1e-6), "%s: pooled sample differs away from a tie" % name
    assert np.all(g.sum(axis=4) <= 1)
    return int(bad.sum())


def test_pooling():
    """pooling > 1 (convRBM.py:245-267, :586-599, :664-665): probabilities exp(x)/(pool + sum exp), one
    multinomial draw per pooling group, free energy log(1 + sum exp), hit summaries, a Gibbs chain and
    the raw statistic sums with the pooled sparsity slope -- every kernel against the oracle."""
    for cid in POOLED_CASES:
        info, o = oracle_for(cid, seed=7, batch=4, Lf=24, cd_k=2, rho=0.05)
        K, M, ds, NW, pool = info["K"], info["M"], info["ds"], info["NW"], info["POOL"]
        tables = build_tables(cid, o)
        n, Lh = 3, 30
        L = Lh + M - 1
        d = synthetic_onehot(n, L, seed=K + 2)
        letters, _ = encode(d)
        for mode in (0, 1, 2):
            prob = np.zeros((n, K, 1, Lh), dtype=np.float32)
            smp = np.zeros_like(prob)
            ones = ctypes.c_ulonglong(0)
            assert lib.emu_hgv(cid, fp(tables), up(letters), n, L, mode, None, fp(prob), fp(smp), ctypes.byref(ones),
                               ctypes.c_uint64(77), 5, 3, KIND_API_H, 2, 2, 128) == 0
            act = o._bottomUpActivity(d) + o._bottomUpActivity(d, True) if mode == 2 else o._bottomUpActivity(d, mode == 1)
            ref = o._bottomUpProbability(act)
            np.testing.assert_allclose(prob, ref, rtol=2e-5, atol=1e-7)
            u = hidden_uniforms(77, 5, np.arange(n) + 3, K, Lh, 1 if mode == 1 else 0, KIND_API_H)
            check_pooled_samples("hgv", smp, ref, u, pool)
            assert ones.value == int(smp.sum())
        fe = np.zeros(n, dtype=np.float32)
        fem = np.zeros((n, K), dtype=np.float32)
        lib.emu_free_energy(cid, fp(tables), up(letters), n, L, fp(fe), fp(fem), 2, 128)
        np.testing.assert_allclose(fe, o.freeEnergy(d), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(fem, o.freeEnergy(d, True), rtol=1e-5, atol=1e-5)
        P = o.motifHitProbs(d)
        hmax = np.zeros((n, K), dtype=np.float32)
        hsum = np.zeros((n, K), dtype=np.float32)
        pos = np.zeros((K, Lh), dtype=np.float32)
        lib.emu_hit_summary(cid, fp(tables), up(letters), n, L, fp(hmax), fp(hsum), fp(pos), 2, 128)
        np.testing.assert_allclose(hmax, P.max(axis=(2, 3)), rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(pos / n, P.mean(axis=(0, 2)), rtol=2e-5, atol=1e-7)
        # the chain: from a pooled-valid start (at most one unit on per group)
        rng = np.random.default_rng(3)
        def pooled_start():
            h = np.zeros(o.fantasy_h.shape)
            grp = h.reshape(h.shape[0], K, 1, -1, pool)
            pick = rng.integers(0, 3 * pool, size=grp.shape[:4])
            for i in range(pool):
                grp[..., i] = (pick == i)
            return grp.reshape(h.shape)
        o.fantasy_h = pooled_start()
        if ds:
            o.fantasy_h_prime = pooled_start()
        o.seq_offset = 2
        h, hp, v, vout, lws = run_gibbs(cid, o, tables, 2, 2, 2, 128)
        P_m, P_mp, v_m = o.gibbs_steps(2)
        assert np.array_equal(v, v_m) and np.array_equal(h, o.fantasy_h)
        assert np.all(h.reshape(h.shape[0], K, 1, -1, pool).sum(axis=4) <= 1)
        # raw sums of both halves (the model half through the stand-alone kernel: no fused variant with pooling)
        lay = (ctypes.c_int * 7)()
        lib.emu_sums_layout(K, M, lay)
        data_off, n_d, model_off, n_m, count, skipb, skipl = list(lay)
        sums = np.zeros(count, dtype=np.float32)
        row = 3 * K * 4 * M + 3 * K + 4
        partials = np.zeros(8 * row, dtype=np.float32)
        assert lib.emu_stats_mfma(cid, fp(tables), up(letters), n, L, lib.emu_letter_words(L), 1, 128, 2,
                                  fp(partials), partials.size, fp(sums[data_off:]), -1, 0) == row
        Lv = o.fantasy_h.shape[3] + M - 1
        assert lib.emu_stats_mfma(cid, fp(tables), up(vout), o.fantasy_h.shape[0], Lv, lws, 0, 128, 2,
                                  fp(partials), partials.size, fp(sums[model_off:]), skipb, skipl) == row
        s = o.local_sums(d, P_m, P_mp, v_m)
        KAM = K * 4 * M
        close_vh(sums[0:KAM], s['vh_d'])
        close_vh(sums[2 * KAM + 2 * K:3 * KAM + 2 * K], s['sw'], rtol=5e-5)
        np.testing.assert_allclose(sums[3 * KAM + 2 * K:3 * KAM + 3 * K], s['sb'], rtol=5e-5, atol=1e-6)
        close_vh(sums[model_off:model_off + KAM], s['vh_m'])
        if ds:
            close_vh(sums[model_off + KAM:model_off + 2 * KAM], s['vh_mp'])
        print("pooling ok", cid, (K, M, ds, pool))


def test_free_energy(cases=None):
    for cid in (cases or CASES):
        info, o = oracle_for(cid)
        K, M, ds = info["K"], info["M"], info["ds"]
        tables = build_tables(cid, o)
        n, L = 7, M + 70
        d = synthetic_onehot(n, L, seed=K + 1)
        letters, _ = encode(d)
        fe = np.zeros(n, dtype=np.float32)
        fem = np.zeros((n, K), dtype=np.float32)
        lib.emu_free_energy(cid, fp(tables), up(letters), n, L, fp(fe), fp(fem), 2, 128)
        np.testing.assert_allclose(fe, o.freeEnergy(d), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(fem, o.freeEnergy(d, True), rtol=1e-5, atol=1e-5)
        print("free energy ok", (K, M, ds))


def test_hit_summary():
    for cid in CASES:
        info, o = oracle_for(cid)
        K, M, ds = info["K"], info["M"], info["ds"]
        tables = build_tables(cid, o)
        for n, L in ((9, M + 149), (5, M + 299)):   # one or two chunks (plain stores / atomic combine) for KP <= 12
            Lh = L - M + 1
            d = synthetic_onehot(n, L, seed=K + 5)
            letters, _ = encode(d)
            P = o.motifHitProbs(d)
            hmax = np.zeros((n, K), dtype=np.float32)
            hsum = np.zeros((n, K), dtype=np.float32)
            pos = np.zeros((K, Lh), dtype=np.float32)
            lib.emu_hit_summary(cid, fp(tables), up(letters), n, L, fp(hmax), fp(hsum), fp(pos), 2, 128)
            np.testing.assert_allclose(hmax, P.max(axis=(2, 3)), rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(hsum / Lh, P.mean(axis=(2, 3)), rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(pos / n, P.mean(axis=(0, 2)), rtol=1e-5, atol=1e-7)
        print("hit summary ok", (K, M, ds))



def test_big():
    """The generic kernels of models beyond the LDS-resident ones (run-time K and M): dense h|v outputs on both
    strands and their sum, a Gibbs step from the masks (tie-aware), the raw sums of both halves, the update, free
    energy and hit summaries -- each against the oracle, under the sanitizers.  Shapes: a 70-letter motif (more than two
    letter-window words), 37 motifs on two strands (a slab that is not a whole mask word: KS = 8), one motif, and two
    pooled models (groups of 3 on two strands, groups of 2 on one); then alphabets other than DNA's (input_dims,
    convRBM.py:68-71: 3 letters as in the reference's own constructor test, 20 on one strand, 5 on two with pooling, 1)."""
    for (K, M, ds, KS, JS, pool, A) in ((37, 70, True, 8, 16, 1, 4), (3, 5, False, 32, 5, 1, 4), (1, 1, True, 32, 1, 1, 4),
                                        (11, 9, True, 8, 4, 3, 4), (5, 12, False, 4, 12, 2, 4),
                                        (6, 4, True, 32, 2, 1, 3), (9, 7, False, 8, 3, 1, 20), (4, 6, True, 4, 6, 2, 5),
                                        (2, 3, False, 32, 3, 1, 1)):
        o = make_oracle(K, M, ds, seed=K + M, batch=3, Lf=24, cd_k=2, wscale=0.8 if M > 20 else 1.0, rho=0.05, pooling=pool, A=A)
        W, b, c = model_arrays(o)
        n, L = 4, M - 1 + 24
        d = synthetic_onehot(n, L, seed=K, A=A)
        letters, flags = encode(d)
        assert flags == 0
        if A != 4:      # the byte-code form of the same rows, and a code beyond the alphabet
            codes = np.ascontiguousarray(d[:, 0].argmax(axis=1).astype(np.uint8))
            l2 = np.zeros_like(letters)
            fl = np.zeros(4, dtype=np.uint32)
            lib.emu_encode_any(None, codes.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), up(l2), up(fl), n, L, A, 2)
            assert fl[0] == 0 and np.array_equal(l2, letters)
            codes[1, 2] = A
            lib.emu_encode_any(None, codes.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), up(l2), up(fl), n, L, A, 2)
            assert fl[0] == 1
            bad = d.copy()
            bad[0, 0, :, 1] = 0.0                        # a column without a letter
            assert encode(bad)[1] == 1
            # dense v | h of this alphabet (the API pass) against the oracle
            hh = np.random.default_rng(3).binomial(1, 0.15, size=(n, K, 1, L - M + 1)).astype(np.float32)
            hp = np.random.default_rng(4).binomial(1, 0.15, size=hh.shape).astype(np.float32) if ds else None
            act = np.zeros((n, 1, A, L), dtype=np.float32)
            prob = np.zeros_like(act)
            smp = np.zeros_like(act)
            lib.emu_vgh_any(fp(W), fp(c), K, M, A, fp(hh), fp(hp) if ds else None, n, L - M + 1, fp(act), fp(prob), fp(smp),
                            ctypes.c_uint64(o.seed), 7, 2, 2, 64)
            np.testing.assert_allclose(act, o._topDownActivity(hh, hp), rtol=1e-5, atol=2e-5)
            uvv = visible_uniforms(o.seed, 7, np.arange(n) + 2, L, KIND_API_V)
            Pv, vv = o._computeVgivenH(hh, hp, uvv)
            np.testing.assert_allclose(prob, Pv, rtol=2e-5, atol=1e-7)
            badv = (smp != vv).any(axis=2)[:, 0]
            if badv.any():
                gap = np.min(np.abs(np.cumsum(Pv[:, 0], axis=1)[:, :max(A - 1, 1)] - uvv[:, None, :]), axis=1)
                assert np.all(gap[badv] < TIE), "dense v|h of a general alphabet: sample differs away from a tie"
        LW = lib.emu_letter_words_any(A, L)
        Lh = L - M + 1
        NW = (K + 31) // 32

        def sample_ties(name, got, prob, u):
            if pool == 1:
                return check_samples(name, got, prob, u)
            return check_pooled_samples(name, got, prob, u, pool)
        for mode in (0, 1, 2):
            act = np.zeros((n, K, 1, Lh), dtype=np.float32)
            prob = np.zeros_like(act)
            smp = np.zeros_like(act)
            ones = ctypes.c_ulonglong(0)
            assert lib.emu_big_hgv(fp(W), fp(b), fp(c), K, M, int(ds), up(letters), n, L, mode, fp(act), fp(prob), fp(smp),
                                   ctypes.byref(ones), None, ctypes.c_uint64(o.seed), 4, 6, KIND_API_H, 2, KS, 2, 64, pool, A) == 0
            if mode == 2:
                ref = o._bottomUpActivity(d) + o._bottomUpActivity(d, True)
            else:
                ref = o._bottomUpActivity(d, mode == 1)
            np.testing.assert_allclose(act, ref, rtol=1e-5, atol=2e-5)
            pref = o._bottomUpProbability(ref)
            np.testing.assert_allclose(prob, pref, rtol=2e-5, atol=1e-7)
            u = hidden_uniforms(o.seed, 4, np.arange(n) + 6, K, Lh, 1 if mode == 1 else 0, KIND_API_H)
            sample_ties("big hgv", smp, pref, u)
            assert ones.value == int(smp.sum())
        # one Gibbs step at a time from the oracle's state (tie-aware)
        rng = np.random.default_rng(9)

        def start():
            h = rng.binomial(1, 0.1, size=o.fantasy_h.shape).astype(np.float64)
            if pool > 1:                                        # a valid pooled state: at most one unit on per group
                g = h.reshape(h.shape[0], K, 1, -1, pool)
                g[..., 1:] = 0.0
            return h
        o.fantasy_h = start()
        if ds:
            o.fantasy_h_prime = start()
        o.seq_offset = 5
        B, Lf = o.fantasy_h.shape[0], o.fantasy_h.shape[3]
        Lv = Lf + M - 1
        idx = np.arange(B) + o.seq_offset
        for _ in range(2):
            t = o.gibbs_step
            hm, _f = pack_hidden(f32(o.fantasy_h), NW)
            hmp = pack_hidden(f32(o.fantasy_h_prime), NW)[0] if ds else np.zeros_like(hm)
            lws = lib.emu_big_gibbs_step(fp(W), fp(b), fp(c), K, M, int(ds), up(hm), up(hmp), None, B, Lf, ctypes.c_uint64(o.seed), t,
                                         o.seq_offset, JS, KS, 2, 64, pool, A)
            vout = np.zeros((B, lws), dtype=np.uint32)
            assert lib.emu_big_gibbs_step(fp(W), fp(b), fp(c), K, M, int(ds), up(hm), up(hmp), up(vout), B, Lf, ctypes.c_uint64(o.seed), t,
                                          o.seq_offset, JS, KS, 2, 64, pool, A) == lws
            gv = np.zeros((B, 1, A, Lv), dtype=np.float32)
            lib.emu_decode_any(up(vout), fp(gv), B, Lv, lws, A, 2)
            uv = visible_uniforms(o.seed, t, idx, Lv, KIND_CHAIN_V)
            Pv, v = o._computeVgivenH(o.fantasy_h, o.fantasy_h_prime if ds else None, uv)
            np.testing.assert_array_equal(gv.sum(axis=2), 1.0)
            badv = (gv != v).any(axis=2)[:, 0]
            if badv.any():
                gap = np.min(np.abs(np.cumsum(Pv[:, 0], axis=1)[:, :max(A - 1, 1)] - uv[:, None, :]), axis=1)
                assert np.all(gap[badv] < TIE), "big v|h: sample differs away from a tie"
            clean = ~badv.any(axis=1)
            uh = hidden_uniforms(o.seed, t, idx, K, Lf, 0, KIND_CHAIN_H)
            P, h = o._computeHgivenV(v, False, uh)
            pairs = [(unpack_hidden(hm, K), h, P, uh)]
            if ds:
                uhp = hidden_uniforms(o.seed, t, idx, K, Lf, 1, KIND_CHAIN_H)
                Pp, hp = o._computeHgivenV(v, True, uhp)
                pairs.append((unpack_hidden(hmp, K), hp, Pp, uhp))
            for got, want, prob, u in pairs:
                if clean.all():
                    sample_ties("big chain h|v", got, prob, u)
                else:
                    assert (got != want)[clean].sum() == 0
            o.fantasy_h, o.fantasy_h_prime = h, (hp if ds else o.fantasy_h_prime)
            o.last_v_model = v
            o.gibbs_step += 1
        assert o.fantasy_h.sum() > 0
        # raw sums of both halves against the oracle's, the update against its finalisation
        lay = (ctypes.c_int * 7)()
        lib.emu_sums_layout_any(K, M, A, lay)
        data_off, n_d, model_off, n_m, count, skip_b, skip_l = list(lay)
        KAM = K * A * M
        row = 3 * KAM + 3 * K + A
        sums = np.zeros(count, dtype=np.float32)
        half = np.zeros(row + 1, dtype=np.float32)
        CH = 16 if pool == 1 else 4 * pool
        assert lib.emu_big_stats(fp(W), fp(b), fp(c), K, M, int(ds), up(letters), n, L, LW, 1, 2 if K < 8 else 1, CH, 64, fp(half), -1, 0, pool, A) == row
        sums[data_off:data_off + row] = half[:row]
        sums[n_d] = n
        P_m, P_mp, v_m = o.gibbs_steps(1)
        vl, _ = encode(v_m)
        vlw = lib.emu_letter_words_any(A, Lv)
        assert lib.emu_big_stats(fp(W), fp(b), fp(c), K, M, int(ds), up(vl), B, Lv, vlw, 0, 3 if K < 8 else 1, 8 if pool == 1 else 2 * pool, 64, fp(half), skip_b, skip_l, pool, A) == row
        sums[model_off:model_off + row - skip_l] = half[:row - skip_l]
        sums[n_m] = B
        ref = o.local_sums(d, P_m, P_mp, v_m)
        got = {"vh_d": sums[data_off:data_off + KAM], "h_d": sums[data_off + 2 * KAM:data_off + 2 * KAM + K],
               "sw": sums[data_off + 2 * KAM + 2 * K:data_off + 3 * KAM + 2 * K], "sb": sums[data_off + 3 * KAM + 2 * K:data_off + 3 * KAM + 3 * K],
               "v_d": sums[data_off + 3 * KAM + 3 * K:data_off + 3 * KAM + 3 * K + A],
               "vh_m": sums[model_off:model_off + KAM], "h_m": sums[model_off + 2 * KAM:model_off + 2 * KAM + K],
               "v_m": sums[model_off + 2 * KAM + 2 * K:model_off + 2 * KAM + 2 * K + A]}
        if ds:
            got.update({"vh_dp": sums[data_off + KAM:data_off + 2 * KAM], "h_dp": sums[data_off + 2 * KAM + K:data_off + 2 * KAM + 2 * K],
                        "vh_mp": sums[model_off + KAM:model_off + 2 * KAM], "h_mp": sums[model_off + 2 * KAM + K:model_off + 2 * KAM + 2 * K]})
        for key, val in got.items():
            np.testing.assert_allclose(val, np.ravel(ref[key]), rtol=2e-5, atol=2e-5, err_msg=key)
        Wn, bn, cn = W.copy(), b.copy(), c.copy()
        vW, vb, vc = (f32(x).copy() for x in (o.vW, o.vb, o.vc))
        lib.emu_big_update(fp(sums), fp(Wn), fp(bn), fp(cn), fp(vW), fp(vb), fp(vc), K, M, int(ds), L, Lf,
                           ctypes.c_float(o.learning_rate), ctypes.c_float(o.momentum), ctypes.c_float(o.rho),
                           ctypes.c_float(o.lambda_rate), 2, 64, A)
        o.finalize_from_sums(ref, L, Lf)
        np.testing.assert_allclose(Wn.reshape(o.W.shape), o.W, rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(bn.reshape(o.b.shape), o.b, rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(cn.reshape(o.c.shape), o.c, rtol=1e-4, atol=2e-6)
        # free energy and hit summaries (the model is the updated oracle now; use fresh arrays)
        W2, b2, c2 = model_arrays(o)
        fe = np.zeros(n, dtype=np.float32)
        fem = np.zeros((n, K), dtype=np.float32)
        lib.emu_big_eval(fp(W2), fp(b2), fp(c2), K, M, int(ds), up(letters), n, L, 0, fp(fe), fp(fem), None, None, None, 2, 64, pool, A)
        np.testing.assert_allclose(fe, o.freeEnergy(d), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(fem, o.freeEnergy(d, True), rtol=1e-4, atol=1e-5)
        hmax = np.zeros((n, K), dtype=np.float32)
        hmean = np.zeros((n, K), dtype=np.float32)
        pos = np.zeros((K, Lh), dtype=np.float32)
        lib.emu_big_eval(fp(W2), fp(b2), fp(c2), K, M, int(ds), up(letters), n, L, 1, None, None, fp(hmax), fp(hmean), fp(pos), 2, 64, pool, A)
        Ph = o.motifHitProbs(d)
        np.testing.assert_allclose(hmax, Ph.max(axis=(2, 3)), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(hmean, Ph.mean(axis=(2, 3)), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(pos / n, Ph.mean(axis=(0, 2)), rtol=1e-5, atol=1e-7)
        print("big kernels ok", (K, M, ds, pool, A))


def test_slabs():
    """A model larger than a compiled case as a row of slabs of that case (crbm_api.hip, slab_launch_hgv / slab_launch_stats):
    h|v of every slab into the model's mask rows -- units on the counters they have in the whole model, the last slab moved
    to a multiple of ten (it overlaps its neighbour and reaches past motif K into zero padding) -- against the oracle of the
    WHOLE model, sample for sample; and the column reduction of a slab's partial rows into the model's sums."""
    for cid, K, pool in ((4, 47, 1), (0, 25, 1), (10, 20, 3)):
        info = case_info(cid)
        Ks, M, ds = info["K"], info["M"], info["ds"]
        Lf = 12 * pool
        o = make_oracle(K, M, ds, seed=K, Lf=Lf, pooling=pool) if pool > 1 else make_oracle(K, M, ds, seed=K, Lf=Lf)
        W, b, c = model_arrays(o)
        Wp = np.concatenate([W, np.zeros((10, 4, M), np.float32)])
        bp = np.concatenate([b, np.zeros(10, np.float32)])
        n, L = 4, Lf + M - 1
        d = synthetic_onehot(n, L, seed=K + 1)
        letters, _ = encode(d)
        Lh, NW = L - M + 1, (K + 31) // 32
        nslab = (K + Ks - 1) // Ks
        origins = [i * Ks if (i + 1) * Ks <= K else (K - Ks + 9) // 10 * 10 for i in range(nslab)]
        tabs = np.zeros(nslab * info["TABLES"], dtype=np.float32)
        assert lib.emu_slab_tables(cid, fp(Wp), fp(bp), fp(c), fp(tabs), K, origins[-1], nslab) == 0
        for i, k0 in enumerate(origins):                 # the image of every slab is the image of that slab as a model
            t = np.zeros(info["TABLES"], dtype=np.float32)
            assert lib.emu_tables(cid, fp(np.ascontiguousarray(Wp[k0:k0 + Ks])), fp(np.ascontiguousarray(bp[k0:k0 + Ks])), fp(c), fp(t)) == 0
            np.testing.assert_array_equal(tabs[i * info["TABLES"]:(i + 1) * info["TABLES"]], t)
        # free energies slab by slab (per sequence and per motif) against the oracle of the whole model
        scratch = np.zeros(nslab * n * Ks, dtype=np.float32)
        fe = np.zeros(n, dtype=np.float32)
        fem = np.zeros((n, K), dtype=np.float32)
        assert lib.emu_slab_fe(cid, fp(tabs), up(letters), n, L, fp(scratch), K, origins[-1], nslab, fp(fe), fp(fem), 2, 128) == 0
        np.testing.assert_allclose(fe, o.freeEnergy(d), rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(fem, o.freeEnergy(d, True), rtol=2e-5, atol=2e-5)
        for strand in range(2 if ds else 1):
            masks = np.zeros((n, Lh, NW), dtype=np.uint32)
            ones = ctypes.c_ulonglong(0)
            assert lib.emu_slab_hgv(cid, fp(tabs), up(letters), n, L, strand, up(masks), NW, K, origins[-1], nslab,
                                    ctypes.byref(ones), ctypes.c_uint64(91), 4, 2, KIND_CHAIN_H, 2, 2, 128) == 0
            smp = unpack_hidden(masks, K).astype(np.float32).reshape(n, K, 1, Lh)
            assert ones.value == int(smp.sum()) and smp.sum() > 0
            x = o._bottomUpActivity(d, strand == 1)
            if pool > 1:
                prob = o._bottomUpProbability(x)
                u = hidden_uniforms(91, 4, np.arange(n) + 2, K, Lh, strand, KIND_CHAIN_H)
                check_pooled_samples("slab hgv", smp, prob, u, pool)
            else:
                u = hidden_uniforms(91, 4, np.arange(n) + 2, K, Lh, strand, KIND_CHAIN_H)
                check_samples("slab hgv", smp, 1 / (1 + np.exp(-x)), u)
        print("slab hgv ok", cid, (K, Ks, M, ds, pool), origins)
    # slab_reduce_kernel, all slabs in one launch: data half (everything carried) and model half (sw, sb dropped); the last slab
    # overlaps its neighbour (the same sums twice, as the kernels produce them) and reaches past motif K (dropped)
    rng = np.random.default_rng(5)
    K, Ks, M = 47, 20, 3
    origins, nslab = [0, 20, 30], 3
    M4, KAM, KAMs = 4 * M, K * 4 * M, Ks * 4 * M
    row, full = 3 * KAMs + 3 * Ks + 4, 3 * KAM + 3 * K + 4
    nrows = 11
    part = rng.standard_normal((nslab, nrows, row)).astype(np.float32)
    classes = [(0, 0, M4), (KAM, KAMs, M4), (2 * KAM, 2 * KAMs, 1), (2 * KAM + K, 2 * KAMs + Ks, 1),
               (2 * KAM + 2 * K, 2 * KAMs + 2 * Ks, M4), (3 * KAM + 2 * K, 3 * KAMs + 2 * Ks, 1)]     # (column in the full row, in the slab row, per motif)
    for dst, src, per in classes:                         # motifs 30..39: slab 2's columns are slab 1's
        part[2, :, src:src + 10 * per] = part[1, :, src + 10 * per:src + 20 * per]
    part[2, :, 3 * KAMs + 3 * Ks:] = part[1, :, 3 * KAMs + 3 * Ks:] = part[0, :, 3 * KAMs + 3 * Ks:]      # letter counts: the same in every slab
    col = part.astype(np.float64).sum(axis=1)
    for dsf, sp, skip_begin, skip_len in ((1, 1, full, 0), (0, 0, 2 * KAM + 2 * K, KAM + K), (1, 0, 2 * KAM + 2 * K, KAM + K)):
        sums = np.full(full + 1 - skip_len, -7.0, dtype=np.float32)
        assert lib.emu_slab_reduce(fp(part), fp(sums), nrows, row, Ks, origins[-1], nslab, K, M4, dsf, sp, skip_begin, skip_len, ctypes.c_float(13.0), 64) == 0
        want = np.full(full + 1, -7.0)
        valid = [True, dsf, True, dsf, sp, sp]
        for (dst, src, per), ok in zip(classes, valid):
            for i, k0 in enumerate(origins):
                kk = min(Ks, K - k0)                      # motifs of the slab that exist
                want[dst + k0 * per:dst + (k0 + kk) * per] = col[i, src:src + kk * per] if ok else 0.0
        want[3 * KAM + 3 * K:3 * KAM + 3 * K + 4] = col[0, 3 * KAMs + 3 * Ks:]
        want[full] = 13.0
        want = np.delete(want, np.arange(skip_begin, skip_begin + skip_len)) if skip_len else want
        np.testing.assert_allclose(sums, want, rtol=1e-5, atol=1e-5)
    print("slab reduce ok")


def test_large_models():
    """Models beyond 64 motifs (masks of more than two words) and beyond 32-letter motifs (two-word letter
    windows, 128-bit statistics windows): every kernel of a training step, the free energy and the chain."""
    test_hgv([12, 13])
    test_gibbs([11, 13])
    test_stats_mfma([12, 13])
    test_train_step([11, 12])
    test_free_energy([13])


if __name__ == "__main__":
    which = sys.argv[1:] or ["encode_pack", "hgv", "vgh", "gibbs", "stats_mfma", "train_step", "two_ranks", "pooling",
                             "free_energy", "hit_summary", "big", "slabs"]
    for w in which:
        globals()["test_" + w]()
    print("EMU ALL OK")
