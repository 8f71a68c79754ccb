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
import hashlib
import hmac
import os
import socket
import struct
import time

import numpy as np

from . import _lib

_ID_PORT_OFFSET = 17
_PORT_TRIES = 8            # rank 0 binds the first free port of MASTER_PORT + 17 + 101*i
_MAGIC = b"CRBMCTL2"
_HANDSHAKE_TIMEOUT = 5.0   # seconds a fresh connection gets to identify itself, in ALL (a silent or slow-dripping stranger must not stall the accept loop)
_MAX_HELLO = 64            # largest frame accepted before a peer has authenticated
_MAX_MSG = 1 << 32         # largest frame accepted afterwards (checkpoint gathers are tens of MB)


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


def empty_shards(total, batchsize, world):
    """Ranks that own no row at all of a `total`-row data set (every rank can compute this: fit()
    must fail on ALL ranks, not only on the empty one, or the others hang in the all-reduce)."""
    return [r for r in range(world)
            if all(shard_range(min(total, s + batchsize) - s, r, world)[0] ==
                   shard_range(min(total, s + batchsize) - s, r, world)[1] for s in range(0, total, batchsize))]


def make_unique_id():
    """A fresh 128-byte RCCL id (crbm_comm_unique_id); call on one rank and ship it to the others."""
    lib = _lib.load()
    buf = (ctypes.c_uint8 * _lib.UNIQUE_ID_BYTES)()
    rc = lib.crbm_comm_unique_id(buf)
    if rc != 0:
        raise Exception("crbm_comm_unique_id failed (%d): %s" % (rc, lib.crbm_last_error(None).decode()))
    return bytes(buf)


# ---------------------------------------------------------------- wire format
# Only plain data travels: None, bool, int, float, bytes, str, list/tuple, dict with str keys and
# NumPy arrays of numeric dtype -- a tagged binary encoding decoded by this module alone.  Nothing a
# peer sends is ever executed or unpickled (pickle.loads on bytes from a TCP peer).
_NUMERIC_KINDS = "biuf"


def _enc(obj, out):
    if obj is None:
        out.append(b"N")
    elif isinstance(obj, (bool, np.bool_)):
        out.append(b"T" if obj else b"F")
    elif isinstance(obj, (int, np.integer)):
        v = int(obj)
        if -(1 << 63) <= v < (1 << 63):
            out.append(b"i" + struct.pack("<q", v))
        else:                                            # 64-bit checksums, seeds
            raw = v.to_bytes((v.bit_length() + 8) // 8, "little", signed=True)
            out.append(b"I" + struct.pack("<I", len(raw)) + raw)
    elif isinstance(obj, (float, np.floating)):
        out.append(b"f" + struct.pack("<d", float(obj)))
    elif isinstance(obj, (bytes, bytearray)):
        out.append(b"b" + struct.pack("<Q", len(obj)) + bytes(obj))
    elif isinstance(obj, str):
        raw = obj.encode("utf-8")
        out.append(b"s" + struct.pack("<Q", len(raw)) + raw)
    elif isinstance(obj, (list, tuple)):
        out.append((b"l" if isinstance(obj, list) else b"t") + struct.pack("<Q", len(obj)))
        for item in obj:
            _enc(item, out)
    elif isinstance(obj, dict):
        out.append(b"d" + struct.pack("<Q", len(obj)))
        for key, val in obj.items():
            if not isinstance(key, str):
                raise TypeError("control plane: dict keys must be str, got %r" % type(key))
            _enc(key, out)
            _enc(val, out)
    elif isinstance(obj, np.ndarray):
        if obj.dtype.kind not in _NUMERIC_KINDS:
            raise TypeError("control plane: arrays must be numeric, got dtype %s" % obj.dtype)
        arr = np.ascontiguousarray(obj)
        dt = arr.dtype.str.encode("ascii")
        out.append(b"a" + struct.pack("<BB", len(dt), arr.ndim) + dt + struct.pack("<%dQ" % arr.ndim, *arr.shape))
        out.append(arr.tobytes())
    else:
        raise TypeError("control plane: cannot send %r" % type(obj))


def encode(obj):
    out = []
    _enc(obj, out)
    return b"".join(out)


class _Reader(object):
    def __init__(self, buf):
        self.buf, self.pos = memoryview(buf), 0

    def take(self, n):
        if n < 0 or self.pos + n > len(self.buf):
            raise ValueError("control plane: truncated message")
        part = self.buf[self.pos:self.pos + n]
        self.pos += n
        return part

    def unpack(self, fmt):
        return struct.unpack(fmt, self.take(struct.calcsize(fmt)))


def _dec(r, depth=0):
    if depth > 32:
        raise ValueError("control plane: message nested too deeply")
    tag = bytes(r.take(1))
    if tag == b"N":
        return None
    if tag in (b"T", b"F"):
        return tag == b"T"
    if tag == b"i":
        return r.unpack("<q")[0]
    if tag == b"I":
        return int.from_bytes(bytes(r.take(r.unpack("<I")[0])), "little", signed=True)
    if tag == b"f":
        return r.unpack("<d")[0]
    if tag == b"b":
        return bytes(r.take(r.unpack("<Q")[0]))
    if tag == b"s":
        return bytes(r.take(r.unpack("<Q")[0])).decode("utf-8")
    if tag in (b"l", b"t"):
        items = [_dec(r, depth + 1) for _ in range(r.unpack("<Q")[0])]
        return items if tag == b"l" else tuple(items)
    if tag == b"d":
        out = {}
        for _ in range(r.unpack("<Q")[0]):
            key = _dec(r, depth + 1)
            if not isinstance(key, str):
                raise ValueError("control plane: bad dict key")
            out[key] = _dec(r, depth + 1)
        return out
    if tag == b"a":
        nd, ndim = r.unpack("<BB")
        dt = np.dtype(bytes(r.take(nd)).decode("ascii"))
        if dt.kind not in _NUMERIC_KINDS or ndim > 8:
            raise ValueError("control plane: bad array header")
        shape = r.unpack("<%dQ" % ndim)
        count = 1
        for d in shape:
            count *= d
        return np.frombuffer(bytes(r.take(count * dt.itemsize)), dtype=dt).reshape(shape).copy()
    raise ValueError("control plane: unknown tag %r" % tag)


def decode(buf):
    r = _Reader(buf)
    obj = _dec(r)
    if r.pos != len(r.buf):
        raise ValueError("control plane: trailing bytes")
    return obj


def job_secret(world, port, addr="127.0.0.1"):
    """Key of the per-message HMAC.  A launcher that wants real authentication passes a random
    CRBM_JOB_SECRET to every rank (bench.py's own spawner does); otherwise the key is derived from
    what all ranks of one job share (launcher run id, rendezvous port, world size), which keeps the
    ranks of two jobs on one host apart but is guessable -- good enough on loopback only: a
    rendezvous on a real interface REQUIRES the secret (every rank evaluates the same condition, so
    all of them refuse together)."""
    s = os.environ.get("CRBM_JOB_SECRET")
    if not s:
        if _bind_address(addr) != "127.0.0.1":
            raise Exception("control plane: MASTER_ADDR=%s is not loopback -- set CRBM_JOB_SECRET (the same random string on "
                            "every rank) so that strangers on that network cannot claim a rank" % addr)
        s = "|".join([os.environ.get("TORCHELASTIC_RUN_ID", ""), str(port), str(world)])
    return hashlib.sha256(b"crbm-control-plane:" + s.encode("utf-8")).digest()


def _recv_exact(sock, n, deadline=None):
    """n bytes; with `deadline` (time.time() value) the WHOLE read must finish by then, however slowly the bytes drip"""
    chunks = []
    while n > 0:
        if deadline is not None:
            left = deadline - time.time()
            if left <= 0:
                raise socket.timeout("control plane: peer too slow")
            sock.settimeout(left)
        part = sock.recv(min(n, 1 << 20))
        if not part:
            raise ConnectionError("control-plane peer closed the connection")
        chunks.append(part)
        n -= len(part)
    return b"".join(chunks)


# Every frame: length | HMAC-SHA256(key, direction | sequence number | payload) | payload.  The per-connection,
# per-direction sequence number makes a recorded frame useless later in the same connection (and the nonce of the
# handshake in another one).
_SEQ = {}


def _mac(key, sock, direction, payload, bump):
    k = (id(sock), direction)
    seq = _SEQ.get(k, 0)
    if bump:
        _SEQ[k] = seq + 1
    return hmac.new(key, direction + struct.pack("<Q", seq) + payload, hashlib.sha256).digest()


def _send_msg(sock, payload, key):
    sock.sendall(struct.pack("<Q", len(payload)) + _mac(key, sock, b"S", payload, True) + payload)


def _recv_msg(sock, key, max_len=_MAX_MSG, deadline=None):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8, deadline))
    if n > max_len:
        raise ConnectionError("control plane: oversized frame (%d bytes)" % n)
    mac = _recv_exact(sock, 32, deadline)
    payload = _recv_exact(sock, n, deadline)
    # the peer's "S" direction is what we verify: its send counter is our receive counter
    k = (id(sock), b"R")
    seq = _SEQ.get(k, 0)
    want = hmac.new(key, b"S" + struct.pack("<Q", seq) + payload, hashlib.sha256).digest()
    if not hmac.compare_digest(mac, want):
        raise ConnectionError("control plane: message authentication failed (a peer of another job, or a replayed frame)")
    _SEQ[k] = seq + 1
    return payload


def _forget(sock):
    _SEQ.pop((id(sock), b"S"), None)
    _SEQ.pop((id(sock), b"R"), None)


def _bind_address(addr):
    """One-node jobs listen on loopback only; a real interface is used only when MASTER_ADDR names one."""
    if not addr or addr in ("localhost", "127.0.0.1", "::1"):
        return "127.0.0.1"
    return addr


class ControlPlane(object):
    """Star-shaped host channel of a one-node job: rank 0 listens, ranks 1..R-1
    connect once and stay connected.  Collectives are tiny and synchronous:
    `broadcast(obj)`, `gather(obj)`, `barrier()`, `allreduce_max(values)`,
    `allreduce_sum(values)`.  World size 1 needs no sockets.

    `timeout` bounds the rendezvous; `collective_timeout` (CRBM_CONTROL_TIMEOUT, default 1800 s)
    bounds how long a rank waits inside a collective for its peers -- e.g. the other ranks of a job
    whose rank 0 evaluates a test set between two collectives."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0, collective_timeout=None):
        if rank is None or world is None:
            rank, world = env_rank_world()
        self.rank, self.world = int(rank), int(world)
        self.peers = []          # rank 0: sockets of ranks 1..R-1, in rank order
        self.sock = None         # other ranks: socket to rank 0
        if collective_timeout is None:
            collective_timeout = float(os.environ.get("CRBM_CONTROL_TIMEOUT", "1800"))
        self.collective_timeout = collective_timeout
        if self.world == 1:
            return
        addr = _bind_address(addr or os.environ.get("MASTER_ADDR", "127.0.0.1"))
        base = int(port or int(os.environ.get("MASTER_PORT", "29500")) + _ID_PORT_OFFSET)
        ports = [base + 101 * i for i in range(_PORT_TRIES)]
        self.key = job_secret(self.world, base, addr)
        # hello = magic | world | rank | nonce, authenticated like every later message; rank 0 answers with
        # an authenticated echo of the nonce, so neither side accepts a peer that lacks the job's key
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
            srv.listen(self.world + 8)
            deadline = time.time() + timeout
            slots = {}
            try:
                while len(slots) < self.world - 1:
                    left = deadline - time.time()
                    if left <= 0:
                        raise TimeoutError("control plane: %d of %d ranks connected within %.0f s"
                                           % (len(slots) + 1, self.world, timeout))
                    srv.settimeout(left)
                    try:
                        conn, _peer = srv.accept()
                    except socket.timeout:
                        continue
                    try:
                        _forget(conn)
                        conn.settimeout(_HANDSHAKE_TIMEOUT)
                        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        hello = _recv_msg(conn, self.key, _MAX_HELLO, time.time() + _HANDSHAKE_TIMEOUT)
                        magic, w, r, nonce = hello[:8], *struct.unpack("<II", hello[8:16]), hello[16:]
                        if magic != _MAGIC or w != self.world or not (0 < r < self.world) or r in slots or len(nonce) != 16:
                            raise ConnectionError("not a rank of this job")
                        _send_msg(conn, _MAGIC + nonce, self.key)
                        conn.settimeout(self.collective_timeout)
                        slots[r] = conn
                    except (OSError, ConnectionError, ValueError, struct.error):
                        _forget(conn)
                        conn.close()              # a stranger, a rank of another job, or a silent connection
            finally:
                srv.close()
            self.peers = [slots[r] for r in range(1, self.world)]
        else:
            deadline = time.time() + timeout
            nonce = os.urandom(16)
            mine = _MAGIC + struct.pack("<II", self.world, self.rank) + nonce
            while self.sock is None:
                for p in ports:
                    s = None
                    try:
                        s = socket.create_connection((addr, p), timeout=_HANDSHAKE_TIMEOUT)
                        _forget(s)
                        s.settimeout(_HANDSHAKE_TIMEOUT)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        _send_msg(s, mine, self.key)
                        if _recv_msg(s, self.key, _MAX_HELLO, time.time() + _HANDSHAKE_TIMEOUT) == _MAGIC + nonce:
                            s.settimeout(self.collective_timeout)
                            self.sock = s
                            break
                        _forget(s)
                        s.close()
                    except (OSError, ConnectionError):
                        if s is not None:
                            _forget(s)
                            s.close()
                if self.sock is None:
                    if time.time() > deadline:
                        raise TimeoutError("control plane: rank 0 not reachable at %s ports %s" % (addr, ports))
                    time.sleep(0.05)

    # rank 0 gathers one object per rank, applies `combine` and sends the result to everyone
    def _collect(self, obj, combine, what="collective"):
        if self.world == 1:
            return combine([obj])
        try:
            if self.rank == 0:
                items = [obj] + [decode(_recv_msg(p, self.key)) for p in self.peers]
                out = combine(items)
                blob = encode(out)
                for p in self.peers:
                    _send_msg(p, blob, self.key)
                return out
            _send_msg(self.sock, encode(obj), self.key)
            return decode(_recv_msg(self.sock, self.key))
        except socket.timeout:
            raise TimeoutError("control plane: %s on rank %d waited %.0f s for its peers -- every rank must make the "
                               "same collective calls in the same order (saveState, fit and attach are collective); "
                               "CRBM_CONTROL_TIMEOUT raises the limit" % (what, self.rank, self.collective_timeout))

    def broadcast(self, obj):
        """rank 0's `obj` on every rank."""
        return self._collect(obj if self.rank == 0 else None, lambda items: items[0], "broadcast")

    def gather(self, obj):
        """list of every rank's `obj` (rank order) on every rank."""
        return self._collect(obj, lambda items: items, "gather")

    def barrier(self):
        self._collect(None, lambda items: None, "barrier")

    def allreduce_max(self, values):
        return self._collect([float(v) for v in values], lambda items: [max(col) for col in zip(*items)], "allreduce_max")

    def allreduce_sum(self, values):
        return self._collect(np.asarray(values, dtype=np.float64),
                             lambda items: np.sum(np.stack(items), axis=0), "allreduce_sum")

    def close(self):
        for p in self.peers:
            _forget(p)
            try:
                p.close()
            except OSError:
                pass
        if self.sock is not None:
            _forget(self.sock)
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


def attach_ipc(model, control):
    """The all-reduce through mapped buffers (include/crbm_amd.h, crbm_ipc_*): every rank exports a handle of
    its sums buffer, the control plane carries the handles to everybody, every rank maps them.  Collective and
    all-or-nothing: if any rank cannot export or map, EVERY rank raises (after the same number of collective
    calls), so no rank is left waiting for a peer that gave up."""
    h = model._h()
    mine = (ctypes.c_uint8 * _lib.IPC_HANDLE_BYTES)()
    rc = model._lib.crbm_ipc_export(h, mine)
    err = None if rc == 0 else "rank %d: crbm_ipc_export failed (%d): %s" % (control.rank, rc, model._lib.crbm_last_error(h).decode())
    handles = control.gather(bytes(mine) if err is None else None)
    if err is None and all(x is not None for x in handles):
        blob = b"".join(handles)
        buf = (ctypes.c_uint8 * len(blob)).from_buffer_copy(blob)
        rc = model._lib.crbm_ipc_attach(h, buf, control.world, control.rank)
        if rc != 0:
            err = "rank %d: crbm_ipc_attach failed (%d): %s" % (control.rank, rc, model._lib.crbm_last_error(h).decode())
    elif err is None:
        err = "a peer could not export its buffer"
    errors = [e for e in control.gather(err) if e]       # also the barrier: nobody publishes before everybody has mapped
    if errors:
        model._lib.crbm_ipc_detach(h)
        raise Exception("mapped-buffer all-reduce not available: " + "; ".join(errors))


def ipc_timed_out(model):
    """True if a wait for a peer's sums ever ran out (crbm_ipc_status): the run is invalid from then on."""
    flag = ctypes.c_int32()
    model._check(model._lib.crbm_ipc_status(model._h(), ctypes.byref(flag)))
    return flag.value != 0


def attach(model, rank=None, world=None, uid=None, control=None, allreduce=None):
    """Make `model` (a CRBM) one rank of a data-parallel job.  Must be called
    before the model touches the GPU.  batchsize stays the GLOBAL number of
    persistent chains; each rank owns batchsize/world of them.  All ranks leave
    with rank 0's parameters, velocities and seed.

    allreduce: "rccl" (default; CRBM_ALLREDUCE overrides) -- ncclAllReduce of the packed sums per step; or
    "ipc" -- the ranks of ONE node map each other's sums buffers and the update launch adds them itself
    (no collective launch; at most 8 ranks)."""
    if rank is None or world is None:
        rank, world = env_rank_world()
    if model._handle is not None:
        raise Exception("attach() must be called before the first GPU call")
    model.rank, model.world_size = rank, world
    if world == 1:
        return model
    allreduce = (allreduce or os.environ.get("CRBM_ALLREDUCE", "rccl")).lower()
    if allreduce not in ("rccl", "ipc"):
        raise Exception("allreduce must be 'rccl' or 'ipc', got %r" % allreduce)
    own = control is None
    cp = ControlPlane(rank, world) if own else control
    model._control = cp
    sync_replicas(model, cp)
    model._allreduce = allreduce
    if allreduce == "ipc":
        attach_ipc(model, cp)
    else:
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
