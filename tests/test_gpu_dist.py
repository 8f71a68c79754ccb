"""GPU tests of the data-parallel path with real processes: two ranks train the same
rows as one rank and must end with the same parameters (<= 1e-5 relative: reduction order) and
IDENTICAL chains.  Every rank builds its model with default, unseeded filters and its own sampler
seed, so the tests only pass if crbm_amd.dist makes the replicas identical.

  * host-reduced, 2 processes on ONE GPU: sharding, sync_replicas, the packed sums buffer,
    crbm_train_local / crbm_train_apply, saveState/loadState across world sizes -- runs on every box;
  * RCCL, 2 processes on TWO GPUs: dist.attach + CRBM.fit (sharded upload, crbm_train_epoch_sharded,
    ncclAllReduce, ncclBroadcast) -- skipped unless crbm_device_count() >= 2.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(mode, tmp_path, world=2, extra_env=None, load=True):
    port = _free_port()
    outs = [str(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), mode, outs[r]],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-3000:])
    if not load:
        return outs, logs
    return [np.load(o, allow_pickle=True) for o in outs], logs


def _single_rank_reference(W0, seed0):
    """the same training on one rank, started from rank 0's filters and seed"""
    import dp_worker as w
    from crbm_amd import CRBM
    m = CRBM(w.K, w.M, epochs=w.EPOCHS, doublestranded=True, batchsize=w.B, cd_k=2, fantasy_hidden_len=w.LF,
             seed=int(seed0))
    m.motifs.set_value(W0)
    D = w.data()
    for _ in range(w.EPOCHS):
        for start in range(0, w.N, w.BS):
            m._trainingFct(D[start:min(w.N, start + w.BS)])
    return m


def _check_against_single(res, tmp_path):
    import dp_worker as w
    from crbm_amd import CRBM
    r0, r1 = res
    assert int(r0["checksum"]) == int(r1["checksum"])                   # replicas bit-identical
    np.testing.assert_array_equal(r0["W"], r1["W"])
    np.testing.assert_array_equal(r0["W0"], r1["W0"])                   # rank 1 adopted rank 0's start
    assert int(r0["seed0"]) == int(r1["seed0"]) == 9000
    ref = _single_rank_reference(r0["W0"], r0["seed0"])
    np.testing.assert_allclose(r0["W"], ref.motifs.get_value(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r0["b"], ref.bias.get_value(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r0["c"], ref.c.get_value(), rtol=1e-5, atol=1e-7)
    vW, vb, vc = ref.get_velocities()
    np.testing.assert_allclose(r0["vW"], vW, rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(r0["vb"], vb, rtol=1e-4, atol=1e-7)
    fh, fhp = ref.get_fantasy()
    np.testing.assert_array_equal(np.concatenate([r0["fh"], r1["fh"]]), fh)       # identical samples
    np.testing.assert_array_equal(np.concatenate([r0["fhp"], r1["fhp"]]), fhp)
    # the checkpoint written by the 2-rank job resumes on ONE rank exactly where the job stood
    back = CRBM.loadState(str(tmp_path / "rank0.state"))
    bh, bhp = back.get_fantasy()
    np.testing.assert_array_equal(bh, fh)
    np.testing.assert_array_equal(bhp, fhp)
    D = w.data()
    back._trainingFct(D[:w.BS])
    ref._trainingFct(D[:w.BS])
    np.testing.assert_allclose(back.motifs.get_value(), ref.motifs.get_value(), rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(back.get_fantasy()[0], ref.get_fantasy()[0])


def test_two_processes_host_reduced_equal_single_rank(tmp_path):
    res, _ = _run_ranks("host", tmp_path)
    _check_against_single(res, tmp_path)


def test_two_processes_ipc_allreduce_equal_single_rank(tmp_path):
    """The all-reduce through mapped buffers (crbm_ipc_*: hipIpcGetMemHandle / hipIpcOpenMemHandle, publish kernel,
    update launch that waits for both flags and adds both copies in rank order) with two REAL processes --
    both on GPU 0, which RCCL cannot do: dist.attach(allreduce="ipc") + CRBM.fit (sharded upload,
    crbm_train_epoch_sharded, a last mini-batch of which rank 0 owns nothing) equals the one-rank run, the
    replicas are bit-identical, the chains identical, and the checkpoint resumes on one rank."""
    res, logs = _run_ranks("ipc", tmp_path)
    assert "Epoch 1: FE=" in logs[0] and "Epoch" not in logs[1]        # rank 0 alone evaluates and prints
    _check_against_single(res, tmp_path)


def test_ipc_allreduce_host_skew_stays_out_of_the_device_wait(tmp_path):
    """Rank 0 enters fit() 1.5 s late and spends 1.5 s in its per-epoch evaluation while rank 1 is already at the next
    epoch -- with a device-side bound of 0.3 s.  The host barrier in front of every epoch's first launch (crbm.fit)
    keeps that skew on the control plane: no time-out, result equal to the one-rank run."""
    res, logs = _run_ranks("ipc_skew", tmp_path, extra_env={"CRBM_IPC_TIMEOUT_MS": "300"})
    assert "Epoch 1: FE=" in logs[0]
    _check_against_single(res, tmp_path)


def test_ipc_allreduce_waits_seconds_on_the_device(tmp_path):
    """Through the C-ABI without any barrier: rank 0 starts every epoch 1.5 s after rank 1, whose update launch waits
    that long on the GPU for rank 0's flag.  The wait is bounded by the GPU's wall clock (default 30 s), not by an
    iteration count: no time-out, result equal to the one-rank run."""
    res, _ = _run_ranks("ipc_devwait", tmp_path)
    _check_against_single(res, tmp_path)


def test_ipc_allreduce_dead_rank_times_out_and_the_gpu_stays_usable(tmp_path):
    """Rank 1 exits after epoch 0.  Rank 0's next epoch returns CRBM_ERR_IPC_TIMEOUT within the bound (0.5 s per
    waiting launch), its parameters are those of the last complete step, and a fresh model trains on the same GPU."""
    outs, logs = _run_ranks("ipc_dead", tmp_path, extra_env={"CRBM_IPC_TIMEOUT_MS": "500"}, load=False)
    assert "survivor ok" in logs[0], logs[0][-2000:]
    r0 = np.load(outs[0])
    assert int(r0["survived"]) == 1 and float(r0["took"]) < 20.0
    assert not os.path.exists(outs[1])                                   # rank 1 really left early


def test_two_gpu_rccl_fit_equals_single_rank(tmp_path):
    from crbm_amd import _lib
    if _lib.load().crbm_device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    res, logs = _run_ranks("rccl", tmp_path)
    assert "Epoch 1: FE=" in logs[0] and "Epoch" not in logs[1]        # rank 0 alone evaluates and prints
    _check_against_single(res, tmp_path)


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher starts two ranks itself."""
    from crbm_amd import _lib
    two = _lib.load().crbm_device_count() >= 2
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    if not two:
        env["CRBM_BENCH_SHARE_GPU"] = "1"          # plumbing rehearsal on one GPU (no RCCL communicator: it refuses two ranks on one device)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
                        "--no-cpu-baseline"], env=env, capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    import json
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 20
    assert line["train"]["global_batch"] == 2 * 8192
    assert line["train"]["all_reduce"].startswith("rccl" if two else "ipc")
    if two:
        assert line["train"]["replicas_identical"] is True
    # the all-reduce through mapped buffers runs in both cases (beside RCCL on two GPUs, alone on one)
    ipc = line["train"]["ipc_all_reduce"]
    assert "error" not in ipc and ipc["ok"] is True and ipc["replicas_identical"] is True and ipc["timed_out"] is False, ipc
    assert line["secondary_ok"] is True
