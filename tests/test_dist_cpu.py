"""world_size-2 CPU tests of the data-parallel path (gloo; no GPU).

What is rank-dependent in the HIP path is host logic: which rows and chains a
rank owns, the Philox sequence offset, the packed raw-sum layout, and that the
update is formed from GLOBAL sums.  Here each rank computes its local raw sums
with the oracle (standing in for the kernels), gloo all-reduces them, and the
result must equal the single-process step on the whole batch."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

from oracle.crbm_oracle import OracleCRBM, synthetic_onehot


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(ds, B):
    return OracleCRBM(6, 7, doublestranded=ds, batchsize=B, cd_k=2, fantasy_hidden_len=25, seed=9,
                      W=np.random.default_rng(3).standard_normal((6, 1, 4, 7)))


def _rank_main(rank, world, port, ds, q):
    import torch
    import torch.distributed as dist
    from crbm_amd.dist import shard_range
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    B, n, L = 8, 11, 40
    D = synthetic_onehot(n, L, seed=4)
    o = _make(ds, B)
    for b in o._iterateBatchIndices(n, 8):                 # two mini-batches, the second short (3 rows)
        lo, hi = shard_range(b[1] - b[0], rank, world)
        rows = D[b[0] + lo:b[0] + hi]
        clo, chi = shard_range(B, rank, world)
        local = _make(ds, chi - clo)
        local.W, local.b, local.c = o.W, o.b, o.c
        local.fantasy_h = o.fantasy_h[clo:chi]
        local.fantasy_h_prime = o.fantasy_h_prime[clo:chi] if ds else None
        local.seq_offset, local.gibbs_step = clo, o.gibbs_step
        P_m, P_mp, v_m = local.gibbs_steps(local.cd_k)
        if len(rows):
            s = local.local_sums(rows, P_m, P_mp, v_m)
        else:                                              # a rank may own no data rows
            s = local.local_sums(D[:1], P_m, P_mp, v_m)
            for k in ("vh_d", "h_d", "sw", "sb", "v_d", "vh_dp", "h_dp"):
                if k in s:
                    s[k] = np.zeros_like(s[k])
            s["n_d"] = 0.0
        keys = sorted(s)
        flat = torch.from_numpy(np.concatenate([np.ravel(np.asarray(s[k], dtype=np.float64)) for k in keys]))
        dist.all_reduce(flat)                              # sum
        pos = 0
        for k in keys:
            size = int(np.size(s[k]))
            s[k] = flat[pos:pos + size].numpy().reshape(np.shape(s[k])) if np.ndim(s[k]) else float(flat[pos])
            pos += size
        o.finalize_from_sums(s, L, 25)
        # gather the chains back so that the next step starts from the global state
        hs = [torch.zeros(1)] * world
        parts = [None] * world
        dist.all_gather_object(parts, (local.fantasy_h, local.fantasy_h_prime))
        o.fantasy_h = np.concatenate([p[0] for p in parts])
        if ds:
            o.fantasy_h_prime = np.concatenate([p[1] for p in parts])
        o.gibbs_step += o.cd_k
    if rank == 0:
        q.put((o.W, o.b, o.c, o.fantasy_h))
    dist.destroy_process_group()


@pytest.mark.parametrize("ds", [False, True])
def test_two_rank_step_equals_single_process(ds):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, ds, q)) for r in range(2)]
    for p in procs:
        p.start()
    W, b, c, fh = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _make(ds, 8)
    D = synthetic_onehot(11, 40, seed=4)
    for lo, hi in ref._iterateBatchIndices(11, 8):
        ref.train_step(D[lo:hi])
    np.testing.assert_allclose(W, ref.W, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(b, ref.b, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(c, ref.c, rtol=1e-10, atol=1e-13)
    np.testing.assert_array_equal(fh, ref.fantasy_h)


def _id_rank(rank, port, q):
    from crbm_amd import dist
    uid = dist.exchange_unique_id(rank, 2, addr="127.0.0.1", port=port,
                                  make_id=lambda: bytes(range(128)))
    q.put((rank, uid))


def test_unique_id_exchange_over_tcp():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_id_rank, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
    assert got[0] == got[1] == bytes(range(128))


def test_shard_ranges_and_model_sharding():
    from crbm_amd import CRBM
    from crbm_amd.dist import shard_range
    assert [shard_range(11, r, 4) for r in range(4)] == [(0, 2), (2, 5), (5, 8), (8, 11)]
    assert [shard_range(3, r, 8) for r in range(8)].count((0, 0)) >= 1      # empty shards exist
    m = CRBM(4, 5, batchsize=16)
    m.rank, m.world_size = 3, 8
    assert m._shard_rows(40, 48) == (43, 44)
    covered = []
    for r in range(8):
        m.rank = r
        lo, hi = m._shard_rows(100, 120)
        covered += list(range(lo, hi))
    assert covered == list(range(100, 120))


# ---- the product's own control plane (crbm_amd.dist.ControlPlane: no torch) -------------------
def _cp_rank(rank, world, port, q):
    from crbm_amd.dist import ControlPlane
    cp = ControlPlane(rank, world, addr="127.0.0.1", port=port)
    got = {"bcast": cp.broadcast({"x": 41 + rank} if rank == 0 else None),
           "gather": cp.gather(rank * 10),
           "max": cp.allreduce_max([rank + 0.5, -rank]),
           "sum": cp.allreduce_sum([1.0, rank]).tolist()}
    cp.barrier()
    cp.close()
    q.put((rank, got))


def test_control_plane_collectives():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 3
    procs = [ctx.Process(target=_cp_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r] == {"bcast": {"x": 41}, "gather": [0, 10, 20], "max": [2.5, 0.0], "sum": [3.0, 3.0]}


def _replica_rank(rank, port, q):
    # every rank builds the model the way a user would: default (unseeded) filters, wall-clock seed
    import numpy as np
    from crbm_amd import CRBM
    from crbm_amd.dist import ControlPlane, sync_replicas
    np.random.seed(100 + rank)
    m = CRBM(4, 6, batchsize=8, seed=1000 + rank)
    before = m.motifs.get_value().copy()
    cp = ControlPlane(rank, 2, addr="127.0.0.1", port=port)
    sync_replicas(m, cp)
    cp.close()
    q.put((rank, before, m.motifs.get_value(), m.bias.get_value(), m.seed))


def test_sync_replicas_makes_default_initialised_ranks_identical():
    """CRBM() draws W from the process-global generator and seeds the sampler from
    the clock; attach() must leave every rank with rank 0's parameters and seed."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replica_rank, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: rest for r, *rest in (q.get(timeout=60) for _ in range(2))}
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert not np.array_equal(got[0][0], got[1][0])           # they really started different
    np.testing.assert_array_equal(got[0][1], got[0][0])       # rank 0 keeps its own
    np.testing.assert_array_equal(got[1][1], got[0][0])       # rank 1 adopts them
    np.testing.assert_array_equal(got[1][2], got[0][2])
    assert got[0][3] == got[1][3] == 1000


def test_shard_rows_cover_every_slice_once():
    from crbm_amd.dist import shard_range, shard_rows
    total, bs, world = 37, 8, 3
    parts = [shard_rows(total, bs, r, world) for r in range(world)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(total))
    # rank r's rows of slice [start, end) are the contiguous range the library computes (crbm_api.hip)
    for r in range(world):
        want = []
        for start in range(0, total, bs):
            n = min(total, start + bs) - start
            lo, hi = shard_range(n, r, world)
            want += list(range(start + lo, start + hi))
        assert parts[r].tolist() == want
    assert shard_rows(2, 8, 2, 3).size in (0, 1)              # a rank may own nothing of a short set


def test_wire_format_round_trip_and_rejects_garbage():
    """The control plane carries plain data in a tagged binary encoding of its own (no pickle):
    everything the job sends round-trips, anything else is refused on either side."""
    from crbm_amd.dist import encode, decode
    state = {"motifs": np.arange(24, dtype=np.float32).reshape(2, 1, 4, 3), "seed": 2**63 + 5, "pending": None,
             "bits": (np.packbits(np.ones(17, np.uint8)), None), "rng": (7, 1, 2), "list": [1.5, -2, True, "x", b"\x00\xff"],
             "empty": np.zeros((0, 3), np.int64)}
    back = decode(encode(state))
    assert set(back) == set(state) and back["seed"] == state["seed"] and back["rng"] == (7, 1, 2)
    np.testing.assert_array_equal(back["motifs"], state["motifs"])
    assert back["motifs"].dtype == np.float32 and back["motifs"].flags.writeable
    np.testing.assert_array_equal(back["bits"][0], state["bits"][0])
    assert back["bits"][1] is None and back["list"] == state["list"] and back["empty"].shape == (0, 3)
    with pytest.raises(TypeError):
        encode({"f": lambda: 0})
    with pytest.raises(TypeError):
        encode(np.array(["a"], dtype=object))
    import pickle
    for junk in (pickle.dumps({"x": 1}), b"", b"a\x02\x01<f", encode([1, 2]) + b"N", b"l" + b"\xff" * 8):
        with pytest.raises((ValueError, Exception)):
            decode(junk)


def _cp_pair_rank(rank, port, q):
    from crbm_amd.dist import ControlPlane
    cp = ControlPlane(rank, 2, addr="127.0.0.1", port=port, timeout=60.0)
    q.put((rank, cp.gather(rank + 1)))
    cp.close()


def test_control_plane_survives_strangers(monkeypatch):
    """a silent connection, a peer with the wrong key and a peer that claims a rank slot
    must neither take a slot nor stall the rendezvous: the real rank 1 still gets in."""
    import hashlib
    import hmac as _hmac
    import struct
    import threading
    import time
    from crbm_amd import dist
    monkeypatch.setenv("CRBM_JOB_SECRET", "s3cret-of-this-test")
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p0 = ctx.Process(target=_cp_pair_rank, args=(0, port, q))
    p0.start()

    def connect():
        deadline = time.time() + 30
        while True:
            try:
                return socket.create_connection(("127.0.0.1", port), timeout=2.0)
            except OSError:
                if time.time() > deadline:
                    raise
                time.sleep(0.05)

    silent = connect()                                   # never says a word
    wrong = connect()                                    # speaks the protocol, but with another job's key
    hello = dist._MAGIC + struct.pack("<II", 2, 1) + b"n" * 16
    badkey = hashlib.sha256(b"other job").digest()
    wrong.sendall(struct.pack("<Q", len(hello)) + _hmac.new(badkey, hello, hashlib.sha256).digest() + hello)
    t0 = time.time()
    p1 = ctx.Process(target=_cp_pair_rank, args=(1, port, q))
    p1.start()
    got = dict(q.get(timeout=60) for _ in range(2))
    took = time.time() - t0
    for p in (p0, p1):
        p.join(timeout=30)
        assert p.exitcode == 0
    assert got[0] == got[1] == [1, 2]
    assert took < 40, took                               # bounded by the handshake timeouts, not by 120 s each
    assert wrong.recv(64) == b""                         # the impostor was dropped without an answer
    silent.close(); wrong.close()


def test_control_plane_frames_cannot_be_replayed_and_real_interfaces_need_a_secret(monkeypatch):
    """the MAC covers a per-connection, per-direction sequence number (a recorded frame does not verify a
    second time), the frame a peer may send before it has authenticated is capped at 64 bytes and must arrive within
    one overall deadline however slowly it drips, and a rendezvous on a real interface requires CRBM_JOB_SECRET."""
    import struct
    import time
    from crbm_amd import dist
    a, b = socket.socketpair()
    key = dist.job_secret(2, 12345)
    try:
        dist._send_msg(a, b"first", key)
        dist._send_msg(a, b"second", key)
        raw = b.recv(4096, socket.MSG_PEEK)
        assert dist._recv_msg(b, key) == b"first" and dist._recv_msg(b, key) == b"second"
        a.sendall(raw[:8 + 32 + 5])                      # the recorded first frame once more
        with pytest.raises(ConnectionError, match="authentication"):
            dist._recv_msg(b, key)
        # a frame that claims to be large is refused before its body is read; a dripping peer runs into the deadline
        a.sendall(struct.pack("<Q", 1 << 20))
        with pytest.raises(ConnectionError, match="oversized"):
            dist._recv_msg(b, key, dist._MAX_HELLO)
        a.sendall(struct.pack("<Q", 32) + b"x" * 10)
        t0 = time.time()
        with pytest.raises(socket.timeout):
            dist._recv_msg(b, key, dist._MAX_HELLO, time.time() + 0.3)
        assert time.time() - t0 < 2.0
    finally:
        dist._forget(a); dist._forget(b)
        a.close(); b.close()
    monkeypatch.delenv("CRBM_JOB_SECRET", raising=False)
    with pytest.raises(Exception, match="CRBM_JOB_SECRET"):
        dist.job_secret(2, 29517, "10.1.2.3")
    monkeypatch.setenv("CRBM_JOB_SECRET", "x")
    assert len(dist.job_secret(2, 29517, "10.1.2.3")) == 32


def test_empty_shards_are_known_to_every_rank():
    from crbm_amd.dist import empty_shards, shard_rows
    assert empty_shards(1, 20, 2) == [0]                 # the lone row goes to rank 1: rank 0 owns nothing
    assert empty_shards(40, 20, 2) == []
    for total, bs, world in ((3, 8, 8), (17, 4, 3), (2, 2, 4)):
        assert empty_shards(total, bs, world) == [r for r in range(world) if shard_rows(total, bs, r, world).size == 0]
