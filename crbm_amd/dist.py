"""Data-parallel plumbing: one process per GPU, RCCL inside the HIP library.

The reference is single-device (no collective anywhere); this is
the new part.  Mini-batches and fantasy chains are sharded over ranks, every
rank runs the same kernels on its rows, and the library all-reduces one packed
buffer of raw statistic sums per training step (include/crbm_amd.h, "data
parallel").

Nothing here needs torch.  The host side of a job is a tiny control plane
(`ControlPlane`): every rank keeps one TCP connection to rank 0, over which
travel the 128-byte RCCL id, the initial replica state (so that all ranks
start from rank 0's parameters and sampler seed), barriers and the handful of
scalars a benchmark reduces.  The data path -- the per-step all-reduce -- is
RCCL over xGMI inside the library.
"""
import ctypes
import os
import pickle
import socket
import struct
import time

import numpy as np

from . import _lib

_ID_PORT_OFFSET = 17
_PORT_TRIES = 8            # rank 0 binds the first free port of MASTER_PORT + 17 + 101*i
_MAGIC = b"CRBMCTL1"


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(total, rank, world):
    """Contiguous, balanced rows [lo, hi) of `total` owned by `rank`."""
    return (total * rank) // world, (total * (rank + 1)) // world


def shard_rows(total, batchsize, rank, world):
    """Row indices of a `total`-row data set that `rank` owns when every
    sequential slice of `batchsize` rows (convRBM.py:722-726) is split
    contiguously over `world` ranks -- the order crbm_train_epoch_sharded
    expects the resident rows in."""
    rows = []
    for start in range(0, total, batchsize):
        n = min(total, start + batchsize) - start
        lo, hi = shard_range(n, rank, world)
        rows.append(np.arange(start + lo, start + hi, dtype=np.int64))
    return np.concatenate(rows) if rows else np.zeros(0, dtype=np.int64)


def make_unique_id():
    """A fresh 128-byte RCCL id (crbm_comm_unique_id); call on one rank and ship it to the others."""
    lib = _lib.load()
    buf = (ctypes.c_uint8 * _lib.UNIQUE_ID_BYTES)()
    rc = lib.crbm_comm_unique_id(buf)
    if rc != 0:
        raise Exception("crbm_comm_unique_id failed (%d): %s" % (rc, lib.crbm_last_error(None).decode()))
    return bytes(buf)


# ---------------------------------------------------------------- control plane
def _send_msg(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    chunks = []
    while n > 0:
        part = sock.recv(min(n, 1 << 20))
        if not part:
            raise ConnectionError("control-plane peer closed the connection")
        chunks.append(part)
        n -= len(part)
    return b"".join(chunks)


def _recv_msg(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class ControlPlane(object):
    """Star-shaped host channel of a one-node job: rank 0 listens, ranks 1..R-1
    connect once and stay connected.  Collectives are tiny and synchronous:
    `broadcast(obj)`, `gather(obj)`, `barrier()`, `allreduce_max(values)`,
    `allreduce_sum(values)`.  World size 1 needs no sockets."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0):
        if rank is None or world is None:
            rank, world = env_rank_world()
        self.rank, self.world = int(rank), int(world)
        self.peers = []          # rank 0: sockets of ranks 1..R-1, in rank order
        self.sock = None         # other ranks: socket to rank 0
        if self.world == 1:
            return
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(port or int(os.environ.get("MASTER_PORT", "29500")) + _ID_PORT_OFFSET)
        ports = [base + 101 * i for i in range(_PORT_TRIES)]
        hello = _MAGIC + struct.pack("<II", self.world, 0)
        if self.rank == 0:
            srv = None
            for p in ports:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr, p))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise OSError("control plane: no free port among %s" % ports)
            srv.listen(self.world)
            srv.settimeout(timeout)
            slots = {}
            try:
                while len(slots) < self.world - 1:
                    conn, _peer = srv.accept()
                    conn.settimeout(timeout)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    head = _recv_exact(conn, len(hello))
                    w, r = struct.unpack("<II", head[len(_MAGIC):])
                    if head[:len(_MAGIC)] != _MAGIC or w != self.world or not (0 < r < self.world) or r in slots:
                        conn.close()          # a stranger, or a rank of another job
                        continue
                    conn.sendall(_MAGIC)
                    slots[r] = conn
            finally:
                srv.close()
            self.peers = [slots[r] for r in range(1, self.world)]
        else:
            deadline = time.time() + timeout
            mine = _MAGIC + struct.pack("<II", self.world, self.rank)
            while self.sock is None:
                for p in ports:
                    try:
                        s = socket.create_connection((addr, p), timeout=5.0)
                        s.settimeout(timeout)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        s.sendall(mine)
                        if _recv_exact(s, len(_MAGIC)) == _MAGIC:
                            self.sock = s
                            break
                        s.close()
                    except (OSError, ConnectionError):
                        pass
                if self.sock is None:
                    if time.time() > deadline:
                        raise TimeoutError("control plane: rank 0 not reachable at %s ports %s" % (addr, ports))
                    time.sleep(0.05)

    # rank 0 gathers one object per rank, applies `combine` and sends the result to everyone
    def _collect(self, obj, combine):
        if self.world == 1:
            return combine([obj])
        if self.rank == 0:
            items = [obj] + [pickle.loads(_recv_msg(p)) for p in self.peers]
            out = combine(items)
            blob = pickle.dumps(out, protocol=4)
            for p in self.peers:
                _send_msg(p, blob)
            return out
        _send_msg(self.sock, pickle.dumps(obj, protocol=4))
        return pickle.loads(_recv_msg(self.sock))

    def broadcast(self, obj):
        """rank 0's `obj` on every rank."""
        return self._collect(obj if self.rank == 0 else None, lambda items: items[0])

    def gather(self, obj):
        """list of every rank's `obj` (rank order) on every rank."""
        return self._collect(obj, lambda items: items)

    def barrier(self):
        self._collect(None, lambda items: None)

    def allreduce_max(self, values):
        return self._collect([float(v) for v in values], lambda items: [max(col) for col in zip(*items)])

    def allreduce_sum(self, values):
        return self._collect(np.asarray(values, dtype=np.float64),
                             lambda items: np.sum(np.stack(items), axis=0))

    def close(self):
        for p in self.peers:
            try:
                p.close()
            except OSError:
                pass
        if self.sock is not None:
            try:
                self.sock.close()
            except OSError:
                pass
        self.peers, self.sock = [], None


def exchange_unique_id(rank, world, addr=None, port=None, timeout=120.0, make_id=None, control=None):
    """Rank 0 creates the id (crbm_comm_unique_id) and every rank receives it.
    `make_id` is injectable for tests without a GPU."""
    own = control is None
    cp = ControlPlane(rank, world, addr=addr, port=port, timeout=timeout) if own else control
    try:
        uid = None
        if rank == 0:
            uid = make_unique_id() if make_id is None else make_id()
            assert len(uid) == _lib.UNIQUE_ID_BYTES
        return cp.broadcast(uid)
    finally:
        if own:
            cp.close()


def sync_replicas(model, control):
    """Every rank adopts rank 0's initial parameters and sampler seed.  The
    reference draws W from the process-global NumPy generator and seeds its
    sampler from the wall clock (convRBM.py:127-131, :155): two processes never
    agree on their own, and after this point only statistic sums are shared.
    Host-only (the model has not touched the GPU yet)."""
    state = None
    if control.rank == 0:
        state = {"motifs": model._host["motifs"], "bias": model._host["bias"], "c": model._host["c"],
                 "seed": model.seed, "pending": model._pending_state}
    state = control.broadcast(state)
    model._host = {"motifs": np.array(state["motifs"], dtype=np.float32),
                   "bias": np.array(state["bias"], dtype=np.float32),
                   "c": np.array(state["c"], dtype=np.float32)}
    model.seed = int(state["seed"])
    model._pending_state = state["pending"]
    return model


def replica_checksum(model):
    """sum of the parameter bit patterns: equal on every rank of a healthy job."""
    tot = 0
    for name in ("motifs", "bias", "c"):
        tot += int(model._get_param(name).view(np.uint32).astype(np.uint64).sum())
    return tot & 0xFFFFFFFFFFFFFFFF


def attach(model, rank=None, world=None, uid=None, control=None):
    """Make `model` (a CRBM) one rank of a data-parallel job.  Must be called
    before the model touches the GPU.  batchsize stays the GLOBAL number of
    persistent chains; each rank owns batchsize/world of them.  All ranks leave
    with rank 0's parameters, velocities and seed."""
    if rank is None or world is None:
        rank, world = env_rank_world()
    if model._handle is not None:
        raise Exception("attach() must be called before the first GPU call")
    model.rank, model.world_size = rank, world
    if world == 1:
        return model
    own = control is None
    cp = ControlPlane(rank, world) if own else control
    model._control = cp
    sync_replicas(model, cp)
    if uid is None:
        uid = exchange_unique_id(rank, world, control=cp)
    h = model._h()
    buf = (ctypes.c_uint8 * _lib.UNIQUE_ID_BYTES).from_buffer_copy(uid)
    model._check(model._lib.crbm_comm_init(h, buf, world, rank))
    # belt and braces on the device side: rank 0's W, b, c and velocities over RCCL
    model._check(model._lib.crbm_comm_broadcast_state(h, 0))
    sums = cp.gather(replica_checksum(model))
    if len(set(sums)) != 1:
        raise Exception("data-parallel replicas differ after attach(): %s" % (sums,))
    return model
