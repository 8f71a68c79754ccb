#!/usr/bin/env python
"""Benchmark of the CRBM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ...`)

A *step* is one Gibbs step (v ~ P(v|h), then h ~ P(h|v)) of the persistent
chain applied to one batch of 8192 chains per GPU -- the PCD-1 Gibbs step of
BASELINE.json's metric, config #2 (10 motifs of length 15, visible 4x200,
single-stranded); chains are sharded over ranks with no data-path collective
(weak scaling), so `value` = N_gpus * K / T in "8192-chain batch Gibbs steps
per second".  Inputs (chain state, parameters) are resident in HBM when the
timed region starts.  torch is used only for the rendezvous, the barrier and
the max-over-ranks; the product path is Python -> ctypes -> HIP.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: K, M, L, doublestranded, chains per GPU, cd_k
    "cfg2": dict(K=10, M=15, L=200, ds=False, chains=8192, k=1,
                 desc="10 motifs len 15, batch 8192 x 4x200, PCD-1, single-stranded"),
    "cfg4": dict(K=50, M=25, L=1000, ds=False, chains=8192, k=5,
                 desc="50 motifs len 25, batch 8192 x 4x1000, PCD-5"),
    "cfg5": dict(K=20, M=15, L=500, ds=True, chains=8192, k=1,
                 desc="20 motifs len 15, doublestranded, batch 8192/GPU x 4x500, PCD-1"),
}
TRAIN_WATCHDOG_S = 180
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def algorithmic_bytes_per_seq(K, M, L, ds):
    """B_step = 8*(S*K*Lh + 4*L) bytes per sequence per Gibbs step
    (dense float32 h read + written, v written + read)."""
    S = 2 if ds else 1
    return 8 * (S * K * (L - M + 1) + 4 * L)


def synthetic_onehot(n, L, seed):
    """(n,1,4,L) float32 one-hot, letters i.i.d. uniform."""
    letters = np.random.default_rng(seed).integers(0, 4, size=(n, L))
    out = np.zeros((n, 1, 4, L), dtype=np.float32)
    out[np.arange(n)[:, None], 0, letters, np.arange(L)[None, :]] = 1
    return out


def build_model(cfg, world, rank):
    from crbm_amd import CRBM
    K, M, L = cfg["K"], cfg["M"], cfg["L"]
    model = CRBM(K, M, doublestranded=cfg["ds"], batchsize=cfg["chains"] * world, cd_k=cfg["k"],
                 fantasy_hidden_len=L - M + 1, seed=2026, device=int(os.environ.get("LOCAL_RANK", "0")))
    # W ~ N(0,1) from default_rng(42); b = norm.ppf(rho); c = 0
    W = np.random.default_rng(42).standard_normal((K, 1, 4, M)).astype(np.float32)
    model.motifs.set_value(W)
    model.rank, model.world_size = rank, world
    return model


def cpu_baseline(cfg, budget_s=12.0):
    """The oracle's C restatement (oracle/crbm_cpu.c) timed on this host's
    cores on the same workload: whole 8192-chain batch, as many steps as fit
    the budget (at least 2)."""
    from oracle import build_cpu
    lib = ctypes.CDLL(build_cpu.build())
    K, M, L, ds, n = cfg["K"], cfg["M"], cfg["L"], cfg["ds"], cfg["chains"]
    Lh = L - M + 1
    import scipy.stats
    W = np.random.default_rng(42).standard_normal((K, 4, M)).astype(np.float32)
    b = np.full(K, scipy.stats.norm.ppf(0.01, 0, np.sqrt(M)), dtype=np.float32)
    c = np.zeros(4, dtype=np.float32)
    h = np.zeros((n, K, Lh), dtype=np.float32)
    hp = np.zeros((n, K, Lh), dtype=np.float32)
    v = np.zeros((n, 4, L), dtype=np.float32)
    F = ctypes.POINTER(ctypes.c_float)
    P = lambda a: a.ctypes.data_as(F)
    cores = lib.crbm_cpu_max_threads()

    def step(t):
        lib.crbm_cpu_gibbs_step(P(W), P(b), P(c), K, M, int(ds), P(h), P(hp), P(v), None, None, n, Lh,
                                ctypes.c_uint64(2026), ctypes.c_uint32(t), ctypes.c_uint32(0), cores)
    step(0)                                  # warm-up (also burn-in of the chain)
    t0 = time.perf_counter()
    done = 0
    while done < 2 or (time.perf_counter() - t0 < budget_s and done < 1000):
        step(1 + done)
        done += 1
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "8192-chain Gibbs steps/s", "cores": int(cores), "kind": "port",
            "sample": "%d Gibbs steps of the full %d-chain batch (oracle/crbm_cpu.c, dense fp32, OpenMP, %d threads), %.1f s"
                      % (done, n, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--multi-step", action="store_true", help="also time 16 Gibbs steps per launch (informational)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    # CRBM_BENCH_BACKEND=gloo rehearses the N>1 plumbing on a box with ONE GPU
    # (all ranks share device 0, no RCCL communicator); the driver's runs use nccl.
    backend = os.environ.get("CRBM_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
        os.environ["LOCAL_RANK"] = "0"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    red_dev = "cpu" if backend == "gloo" else "cuda"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from crbm_amd import _lib
    from crbm_amd import dist as cdist
    from crbm_amd._lib import fptr
    model = build_model(cfg, world, rank)
    h = model._h()
    lib = model._lib
    k = cfg["k"]

    # chains start at h = 0 (convRBM.py:168); 10 burn-in steps, then warm-up
    model._call("crbm_gibbs_steps", 10)
    for _ in range(args.warmup):
        model._call("crbm_gibbs_steps_async", k)
    model._call("crbm_sync")

    # ---- timed region: exactly K steps, HIP events on the library's stream ----
    total_ms = ctypes.c_float()
    barrier()
    t0 = time.perf_counter()
    model._call("crbm_time_gibbs", k, args.steps, ctypes.byref(total_ms))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed, total_ms.value / 1e3], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_s = float(t[0]), float(t[1])
    else:
        kernel_s = total_ms.value / 1e3
    launches = args.steps
    steps_done = args.steps * k                  # Gibbs steps per rank in the timed region
    value = world * steps_done / elapsed

    # informational, opt-in (--multi-step): the same chain advanced 16 steps per launch (no
    # per-launch fixed cost).  Off by default so that every crbm_gibbs launch of a default
    # run is a one-step launch (the rocprofv3 per-kernel average then equals avg_launch_us).
    us_per_step_k16 = None
    if args.multi_step:
        multi_ms = ctypes.c_float()
        model._call("crbm_time_gibbs", 16, 5, ctypes.byref(multi_ms))
        model._call("crbm_time_gibbs", 16, 40, ctypes.byref(multi_ms))
        us_per_step_k16 = 1e3 * multi_ms.value / (40 * 16)

    # hidden-unit activity of the chain (workload descriptor, after timing)
    hf, _ = model.get_fantasy()
    activity = float(hf.mean())

    out = None
    if rank == 0:
        info = _lib.CrbmLaunchInfo()
        lib.crbm_get_launch_info(h, ctypes.byref(info))
        alg_bytes = algorithmic_bytes_per_seq(cfg["K"], cfg["M"], cfg["L"], cfg["ds"]) * cfg["chains"] * k
        avg_launch_s = kernel_s / launches
        achieved = alg_bytes / avg_launch_s / 1e9
        traffic = valu_insts = None
        tfile = os.path.join(ROOT, "profiles", "gibbs_traffic.json")
        if os.path.exists(tfile):
            try:
                pmc = json.load(open(tfile)).get(args.config, {})
                traffic = pmc.get("hbm_bytes_per_launch")
                valu_insts = pmc.get("valu_wave_insts_per_launch")
            except Exception:
                traffic = valu_insts = None
        out = {
            "metric": "Gibbs-steps/sec (PCD-1) at batch 8192x4x200, 10 motifs len 15" if args.config == "cfg2"
                      else "Gibbs-steps/sec, " + cfg["desc"],
            "value": value, "unit": "8192-chain batch Gibbs steps/s (summed over GPUs)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.config + ": " + cfg["desc"], "chains_per_gpu": cfg["chains"],
                       "gibbs_steps_per_launch": k, "visible": "4x%d" % cfg["L"],
                       "hidden": "%dx%d" % (cfg["K"], cfg["L"] - cfg["M"] + 1),
                       "parallelism": "chains sharded over %d GPU(s), no collective in the Gibbs step" % world,
                       "hidden_activity": activity},
            "chain_steps_per_s": value * cfg["chains"],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "crbm_gibbs_sparse" if info.gibbs_sparse else "crbm_gibbs",   # name in the rocprofv3 summaries
                         "avg_launch_us": 1e6 * avg_launch_s,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "state_bytes_per_launch": int(lib.crbm_gibbs_state_bytes(h)),
                         "us_per_step_at_16_steps_per_launch": us_per_step_k16,
                         # what actually binds the kernel (PMC, profiles/): VALU issue time of one launch
                         "valu_wave_insts_per_launch": valu_insts,
                         "valu_issue_us": (valu_insts * 4 / (256 * 4) / 2.4e3) if valu_insts else None,
                         "note": "achieved = algorithmic bytes of the dense fp32 layout (SURVEY 8d) / launch time; "
                                 "the kernel keeps chain state bit-packed, so real HBM traffic is state_bytes_per_launch"},
            "launch": {"grid": info.gibbs_grid, "block": info.gibbs_block, "chains_per_tile": info.gibbs_seqs_per_tile,
                       "lds_bytes": info.gibbs_lds_bytes, "table_group": info.group},
        }

    # The headline line is complete before the secondary section starts.  With N > 1 a
    # watchdog prints it anyway if the RCCL part of the training section should hang on
    # some rank (a hang must not cost the Gibbs measurement).
    done = {"printed": False}

    def emit(train_result):
        if done["printed"]:
            return
        done["printed"] = True
        if rank == 0:
            out["train"] = train_result
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(cfg)
            print(json.dumps(out), flush=True)

    watchdog = None
    if world > 1 and not args.no_train:
        import threading

        def bail():
            emit({"error": "training section did not finish within %d s" % TRAIN_WATCHDOG_S})
            os._exit(0)
        watchdog = threading.Timer(TRAIN_WATCHDOG_S, bail)
        watchdog.daemon = True
        watchdog.start()

    # ---- secondary: full PCD-k training steps (with the RCCL all-reduce when N > 1) ----
    train = None
    if not args.no_train:
        try:
            if world > 1 and backend != "gloo":
                # the 128-byte RCCL id travels over the process group that is already up
                box = [cdist.make_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                uid = box[0]
                buf = (ctypes.c_uint8 * _lib.UNIQUE_ID_BYTES).from_buffer_copy(uid)
                model._call("crbm_comm_init", buf, world, rank)
            n = cfg["chains"]
            D = synthetic_onehot(n, cfg["L"], seed=1234 + rank)
            model._call("crbm_dataset_upload", fptr(D), n, cfg["L"])
            tms = ctypes.c_float()
            model._call("crbm_time_train", 0, n, 5, ctypes.byref(tms))
            tsteps = max(10, min(200, args.steps // 10))
            barrier()
            t1 = time.perf_counter()
            model._call("crbm_time_train", 0, n, tsteps, ctypes.byref(tms))
            barrier()
            tel = time.perf_counter() - t1
            if dist is not None:
                tt = torch.tensor([tel], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                tel = float(tt[0])
            train = {"train_steps_per_s": tsteps / tel, "global_batch": n * world, "cd_k": k,
                     "all_reduce": "rccl" if (world > 1 and backend != "gloo") else "none", "ms_per_train_step": 1e3 * tel / tsteps}
        except Exception as e:                      # report, never hide: the headline is the Gibbs metric
            train = {"error": str(e)[:300]}

    if watchdog is not None:
        watchdog.cancel()
    emit(train)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
