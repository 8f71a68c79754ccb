"""TEST INFRASTRUCTURE: one rank of a data-parallel training run (tests/test_gpu_parity.py starts
two of these as child processes).  Every rank builds its model the way a user would -- default,
unseeded filters and its own sampler seed -- so the run only succeeds if crbm_amd.dist makes the
replicas identical.

    python tests/dp_worker.py <mode> <out.npz>      RANK / WORLD_SIZE / MASTER_* from the environment

mode "rccl": crbm_amd.dist.attach() + CRBM.fit() (sharded upload, RCCL all-reduce inside the library;
             needs one GPU per rank).
mode "ipc":  crbm_amd.dist.attach(allreduce="ipc") + CRBM.fit(): the ranks map each other's sums buffers
             (hipIpcOpenMemHandle) and the update launch adds them; all ranks may share GPU 0.
mode "ipc_skew":    the same with CRBM_IPC_TIMEOUT_MS far below the skew: rank 0 sleeps before fit() and inside its
             per-epoch evaluation -- the host barrier in front of every epoch keeps that skew out of the device-side wait.
mode "ipc_devwait": the C-ABI directly (no barrier): rank 0 enters every epoch 1.5 s late, so rank 1's update launch
             waits that long ON THE DEVICE for rank 0's sums -- inside the default 30 s bound.
mode "ipc_dead":    rank 1 exits after epoch 0; rank 0's next epoch must fail with CRBM_ERR_IPC_TIMEOUT within the bound
             (CRBM_IPC_TIMEOUT_MS), not hang, keep its parameters of the last complete step, and leave the GPU usable.
mode "host": all ranks share GPU 0; the packed sums of crbm_train_local are summed over the control
             plane and applied with crbm_train_apply (same kernels, same sharding, no communicator).
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

K, M, B, LF, N, L, BS, EPOCHS = 5, 6, 8, 33, 17, 40, 8, 2     # 17 rows: slices of 8, 8 and 1 (rank 0 owns none of the last)


def data():
    letters = np.random.default_rng(77).integers(0, 4, size=(N, L))
    out = np.zeros((N, 1, 4, L), dtype=np.float32)
    out[np.arange(N)[:, None], 0, letters, np.arange(L)[None, :]] = 1
    return out


def main():
    mode, outfile = sys.argv[1], sys.argv[2]
    from crbm_amd import CRBM, dist
    from crbm_amd._lib import fptr
    rank, world = dist.env_rank_world()
    np.random.seed(500 + rank)                         # different filters on every rank ...
    model = CRBM(K, M, epochs=EPOCHS, doublestranded=True, batchsize=B, cd_k=2, fantasy_hidden_len=LF,
                 seed=9000 + rank,                    # ... and a different sampler seed
                 device=rank if mode == "rccl" else 0)
    D = data()
    if mode in ("rccl", "ipc", "ipc_skew"):
        dist.attach(model, rank, world, allreduce="rccl" if mode == "rccl" else "ipc")
        W0, seed0 = model.motifs.get_value(), model.seed
        if mode == "ipc_skew" and rank == 0:
            time.sleep(1.5)                              # the other rank is already inside fit()
            evaluate = model._evaluateParams
            model._evaluateParams = lambda: (time.sleep(1.5), evaluate())[1]   # ... and comes back to the next epoch early
        model.fit(D)
        if mode != "rccl":
            assert not dist.ipc_timed_out(model), "a wait for a peer's sums timed out"
    elif mode in ("ipc_devwait", "ipc_dead"):
        dist.attach(model, rank, world, allreduce="ipc")
        W0, seed0 = model.motifs.get_value(), model.seed
        cp = model._control
        model._upload(D[dist.shard_rows(N, BS, rank, world)], 0)
        for epoch in range(EPOCHS):
            cp.barrier()
            if mode == "ipc_dead" and epoch == 1:
                if rank == 1:
                    os._exit(0)                          # dies with its handle, its mappings and its half of the sums
                Wk = model.motifs.get_value()
                t0 = time.time()
                try:
                    model._call("crbm_train_epoch_sharded", BS, N, L)
                    raise SystemExit("the epoch without rank 1 did not fail")
                except Exception as e:
                    took = time.time() - t0
                    assert "in vain" in str(e) and "(-7)" in str(e), str(e)
                    assert took < 20.0, took             # bound 0.5 s per step that waits; nothing hangs
                assert dist.ipc_timed_out(model)
                np.testing.assert_array_equal(model.motifs.get_value(), Wk)   # no update was applied after the time-out
                # the GPU is still usable: a fresh single-rank model trains and samples
                from crbm_amd import CRBM as Fresh
                f = Fresh(K, M, doublestranded=True, batchsize=B, cd_k=2, fantasy_hidden_len=LF, seed=3)
                f._trainingFct(D[:BS])
                f.gibbsSteps(2)
                assert np.isfinite(f.freeEnergy(D)).all() and f.get_fantasy()[0].sum() > 0
                np.savez(outfile, survived=1, took=took)
                print("survivor ok: time-out after %.2f s" % took)
                os._exit(0)                              # (no collective shutdown: the peer is gone)
            if mode == "ipc_devwait" and rank == 0:
                time.sleep(1.5)                          # rank 1's update launch is waiting on the device meanwhile
            model._call("crbm_train_epoch_sharded", BS, N, L)
        assert not dist.ipc_timed_out(model), "a wait for a peer's sums timed out"
    else:
        cp = dist.ControlPlane(rank, world)
        dist.sync_replicas(model, cp)
        model.rank, model.world_size, model._control = rank, world, cp
        W0, seed0 = model.motifs.get_value(), model.seed
        h = model._h()
        count = model._lib.crbm_sums_count(h)
        for _ in range(EPOCHS):
            for start in range(0, N, BS):
                n = min(N, start + BS) - start
                lo, hi = dist.shard_range(n, rank, world)
                rows = np.ascontiguousarray(D[start + lo:start + hi])
                sums = np.zeros(count, dtype=np.float32)
                model._call("crbm_train_local", fptr(rows) if hi > lo else None, hi - lo, L, fptr(sums))
                tot = cp.allreduce_sum(sums).astype(np.float32)
                model._call("crbm_train_apply", fptr(np.ascontiguousarray(tot)), L)
    fh, fhp = model.get_fantasy()
    vW, vb, vc = model.get_velocities()
    state = os.path.splitext(outfile)[0] + ".state"
    model.saveState(state)                             # collective: rank 0 writes the global chains
    np.savez(outfile, W0=W0, seed0=seed0, W=model.motifs.get_value(), b=model.bias.get_value(),
             c=model.c.get_value(), fh=fh, fhp=fhp, vW=vW, vb=vb, vc=vc, checksum=dist.replica_checksum(model))
    if model._control is not None:
        model._control.barrier()


if __name__ == "__main__":
    main()
