"""Data-parallel plumbing: one process per GPU, RCCL inside the HIP library.

The reference is single-device (no collective anywhere); this is
the new part.  Mini-batches and fantasy chains are sharded over ranks, every
rank runs the same kernels on its rows, and the library all-reduces one packed
buffer of raw statistic sums per training step (include/crbm_amd.h, "data
parallel").  Only the 128-byte RCCL unique id crosses processes on the host;
it travels over a plain TCP socket to MASTER_ADDR (no torch needed).
"""
import ctypes
import os
import socket
import time

from . import _lib

_ID_PORT_OFFSET = 17


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(total, rank, world):
    """Contiguous, balanced rows [lo, hi) of `total` owned by `rank`."""
    return (total * rank) // world, (total * (rank + 1)) // world


def make_unique_id():
    """A fresh 128-byte RCCL id (crbm_comm_unique_id); call on one rank and ship it to the others."""
    lib = _lib.load()
    buf = (ctypes.c_uint8 * _lib.UNIQUE_ID_BYTES)()
    rc = lib.crbm_comm_unique_id(buf)
    if rc != 0:
        raise Exception("crbm_comm_unique_id failed (%d): %s" % (rc, lib.crbm_last_error(None).decode()))
    return bytes(buf)


def exchange_unique_id(rank, world, addr=None, port=None, timeout=120.0, make_id=None):
    """Rank 0 creates the id (crbm_comm_unique_id) and serves it to the other
    ranks over TCP.  `make_id` is injectable for tests without a GPU."""
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port or int(os.environ.get("MASTER_PORT", "29500")) + _ID_PORT_OFFSET)
    if rank == 0:
        uid = make_unique_id() if make_id is None else make_id()
        assert len(uid) == _lib.UNIQUE_ID_BYTES
        if world > 1:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(timeout)
            try:
                for _ in range(world - 1):
                    conn, _peer = srv.accept()
                    conn.sendall(uid)
                    conn.close()
            finally:
                srv.close()
        return uid
    deadline = time.time() + timeout
    while True:
        try:
            s = socket.create_connection((addr, port), timeout=5.0)
            break
        except OSError:
            if time.time() > deadline:
                raise
            time.sleep(0.05)
    chunks = b""
    while len(chunks) < _lib.UNIQUE_ID_BYTES:
        part = s.recv(_lib.UNIQUE_ID_BYTES - len(chunks))
        if not part:
            raise Exception("unique-id server closed the connection early")
        chunks += part
    s.close()
    return chunks


def attach(model, rank=None, world=None, uid=None):
    """Make `model` (a CRBM) one rank of a data-parallel job.  Must be called
    before the model touches the GPU.  batchsize stays the GLOBAL number of
    persistent chains; each rank owns batchsize/world of them."""
    if rank is None or world is None:
        rank, world = env_rank_world()
    if model._handle is not None:
        raise Exception("attach() must be called before the first GPU call")
    model.rank, model.world_size = rank, world
    if world == 1:
        return model
    if uid is None:
        uid = exchange_unique_id(rank, world)
    h = model._h()
    buf = (ctypes.c_uint8 * _lib.UNIQUE_ID_BYTES).from_buffer_copy(uid)
    model._check(model._lib.crbm_comm_init(h, buf, world, rank))
    return model
