#!/usr/bin/env python
"""Benchmark of the CRBM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: this process starts the N ranks itself (plain child
processes, created before anything here touches a GPU; RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set for each) and relays rank 0's line.  Under a launcher
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`)
the ranks read the same variables from the environment.  No torch anywhere:
rendezvous, barriers and the max-over-ranks ride on crbm_amd.dist.ControlPlane
(one TCP connection per rank), the data path is Python -> ctypes -> HIP, and
the per-step all-reduce of the training section is RCCL inside the library.

A *step* is one Gibbs step (v ~ P(v|h), then h ~ P(h|v)) of the persistent
chain applied to one batch of 8192 chains per GPU -- the PCD-1 Gibbs step of
BASELINE.json's metric, config #2 (10 motifs of length 15, visible 4x200,
single-stranded); chains are sharded over ranks with no data-path collective
(weak scaling), so `value` = N_gpus * K / T in "8192-chain batch Gibbs steps
per second", T = the HOST clock of each rank from the opening barrier (every
stream idle) until its own K launches have completed, max over ranks, then a
closing barrier (metric_version 3;
round 2 quoted the HIP-event time, which is still reported as
`device_ms_per_step` and is what the roofline object prices the kernel with).
Inputs (chain state, parameters) are resident in HBM when the timed region
starts.

With one GPU and `rocprofv3` on PATH the counters behind the roofline object
(VALU issue cycles, HBM bytes) are collected for THIS box by three short child
runs of tools/prof_gibbs.py under `rocprofv3 --pmc`, started before this process
touches the GPU; otherwise the tracked values of profiles/gibbs_traffic.json
are quoted and marked as not measured in this run.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import threading
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
TRAIN_WATCHDOG_S = 240
SPAWN_TIMEOUT_S = 900
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec; the measured copy rate is reported next to it


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


def build_model(cfg, world, rank, device):
    from crbm_amd import CRBM
    K, M, L = cfg["K"], cfg["M"], cfg["L"]
    model = CRBM(K, M, doublestranded=cfg["ds"], batchsize=cfg["chains"] * world, cd_k=cfg["k"],
                 fantasy_hidden_len=L - M + 1, seed=2026, device=device)
    # W ~ N(0,1) from default_rng(42); b = norm.ppf(rho); c = 0
    W = np.random.default_rng(42).standard_normal((K, 1, 4, M)).astype(np.float32)
    model.motifs.set_value(W)
    model.rank, model.world_size = rank, world
    return model


def cpu_share(hw_threads):
    """CPUs this process can really use: the affinity mask, capped by the cgroup's CPU quota (v2 cpu.max, v1 cfs quota)."""
    n = hw_threads
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(cfg, budget_s=10.0):
    """The oracle's C restatement (oracle/crbm_cpu.c) timed on this host's cores on the same
    workload: whole 8192-chain batch on all cores, a 1/16 batch on one core, plus the NumPy
    oracle's PCD-5 training step of BASELINE config #1 (the reference's own CPU-runnable case)."""
    from oracle import build_cpu
    lib = ctypes.CDLL(build_cpu.build())
    K, M, L, ds, n = cfg["K"], cfg["M"], cfg["L"], cfg["ds"], cfg["chains"]
    Lh = L - M + 1
    import scipy.stats
    W = np.random.default_rng(42).standard_normal((K, 4, M)).astype(np.float32)
    b = np.full(K, scipy.stats.norm.ppf(0.01, 0, np.sqrt(M)), dtype=np.float32)
    c = np.zeros(4, dtype=np.float32)
    F = ctypes.POINTER(ctypes.c_float)
    P = lambda a: a.ctypes.data_as(F)
    hw_threads = lib.crbm_cpu_max_threads()
    cores = cpu_share(hw_threads)          # what this process may actually use (affinity mask, cgroup quota)

    def run(nchains, threads, budget, max_steps):
        h = np.zeros((nchains, K, Lh), dtype=np.float32)
        hp = np.zeros((nchains, K, Lh), dtype=np.float32)
        v = np.zeros((nchains, 4, L), dtype=np.float32)

        def step(t):
            lib.crbm_cpu_gibbs_step(P(W), P(b), P(c), K, M, int(ds), P(h), P(hp), P(v), None, None, nchains, Lh,
                                    ctypes.c_uint64(2026), ctypes.c_uint32(t), ctypes.c_uint32(0), threads)
        step(0)                              # warm-up (also burn-in of the chain)
        t0 = time.perf_counter()
        done = 0
        while done < 2 or (time.perf_counter() - t0 < budget and done < max_steps):
            step(1 + done)
            done += 1
        return done, time.perf_counter() - t0

    done, dt = run(n, cores, budget_s / 2, 1000)
    if cores >= 4:                           # SMT siblings often lose: also half the hardware threads, keep the better
        d2, t2 = run(n, cores // 2, budget_s / 2, 1000)
        if d2 / t2 > done / dt:
            done, dt, cores = d2, t2, cores // 2
    sub = max(64, n // 16)
    done1, dt1 = run(sub, 1, 3.0, 50)
    out = {"value": done / dt, "unit": "8192-chain Gibbs steps/s", "cores": int(cores), "kind": "port",
           "hardware_threads_of_the_host": int(hw_threads),
           "cores_note": "threads used = the CPU share of this process (scheduler affinity mask and cgroup CPU quota), "
                         "not the host's hardware threads: a GPU box hands a job a fraction of its host",
           "sample": "%d Gibbs steps of the full %d-chain batch (oracle/crbm_cpu.c, dense fp32, OpenMP, %d threads), %.1f s"
                     % (done, n, cores, dt),
           "value_1thread": (done1 / dt1) * sub / n,
           "parallel_efficiency": (done / dt) / ((done1 / dt1) * sub / n * cores),
           "sample_1thread": "%d Gibbs steps of %d chains on 1 thread (%.1f s), scaled to the %d-chain batch"
                             % (done1, sub, dt1, n)}
    try:
        # BASELINE config #1 (1000 x 200 bp, 10 motifs len 15, reference defaults batchsize 20, cd_k 5,
        # doublestranded): float64 NumPy restatement of convRBM.py:373-438, a bounded number of its
        # 50 steps per epoch, extrapolated
        from oracle.crbm_oracle import OracleCRBM, synthetic_onehot as oracle_onehot
        D = oracle_onehot(1000, 200, seed=1234)
        o = OracleCRBM(10, 15, doublestranded=True, batchsize=20, cd_k=5, seed=2026,
                       W=np.random.default_rng(42).standard_normal((10, 1, 4, 15)))
        t0 = time.perf_counter()
        nst = 0
        while nst < 2 or (time.perf_counter() - t0 < 4.0 and nst < 50):
            o.train_step(D[20 * nst:20 * (nst + 1)])
            nst += 1
        per = (time.perf_counter() - t0) / nst
        out["cfg1_fit_seconds"] = per * 50
        out["cfg1_sample"] = "%d of the 50 PCD-5 steps of one epoch of config #1 (oracle/crbm_oracle.py, float64 NumPy, " \
                             "NumPy threads as the host configures them), %.2f s per step, extrapolated to the epoch" % (nst, per)
    except Exception as e:                    # the baseline is a report, never a reason to lose the line
        out["cfg1_fit_seconds"] = None
        out["cfg1_sample"] = "failed: %s" % str(e)[:200]
    return out


def roofline_object(name, cfg, info, step_s, counters, counters_src, copy_gbs, state_bytes, us_per_step_k16=None, clock_mhz=None):
    """The roofline object of the chain kernel at one configuration.  step_s: HIP-event time of one launch step (all
    chains advanced by cfg["k"] Gibbs steps) in seconds; counters: {name: mean per launch} from measure_pmc or None."""
    k = cfg["k"]
    parts = max(1, int(info.chain_parts))
    alg_bytes = algorithmic_bytes_per_seq(cfg["K"], cfg["M"], cfg["L"], cfg["ds"]) * cfg["chains"] * k
    alg_gbs = alg_bytes / step_s / 1e9
    # Counters of the chain kernel: measured on this box by the child runs, else the tracked summary of an earlier box
    # (profiles/gibbs_traffic.json), named as such
    if counters is None:
        tfile = os.path.join(ROOT, "profiles", "gibbs_traffic.json")
        try:
            t = json.load(open(tfile)).get(name, {})
            counters = t.get("counters")
            counters_src = "NOT measured in this run (%s); quoted from %s" % (counters_src, t.get("source"))
            if counters and int(t.get("chain_parts", 1)) != parts:
                counters, counters_src = None, counters_src + " -- dropped: collected with another partitioning of the launch"
        except Exception:
            counters = None
    roof = {"kernel": "crbm_gibbs_sparse" if info.gibbs_sparse else "crbm_gibbs",   # name in the rocprofv3 summaries
            "avg_launch_us": 1e6 * step_s, "counters_source": counters_src,
            "launches_per_step": parts,
            "launch_note": ("one step of the whole batch = %d launches of this kernel, one per partition of the chains, on streams of "
                            "their own; they overlap, so avg_launch_us (HIP events over the timed region / steps: what the step costs) "
                            "is not the duration of one launch -- kernel_avg_us is (rocprofv3 --kernel-trace --stats of the same "
                            "launches, measured in this run)" % parts) if parts > 1 else
                           "one step of the whole batch = one launch of this kernel",
            "measured_copy_gbs": copy_gbs,
            "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_gbs": alg_gbs, "algorithmic_frac": alg_gbs / HBM_PEAK_GBS,
            "algorithmic_note": "SURVEY 8(d) bytes of the reference's dense fp32 layout / step time / 8 TB/s: an "
                                "equivalent-dense rate, not a bandwidth utilisation (the chain state is bit-packed, so it may "
                                "exceed 1); north_star's '>= 50 % of HBM roofline' is stated in these units",
            "state_bytes_per_launch": state_bytes,
            "shader_clock_mhz_timed_run": clock_mhz if clock_mhz and clock_mhz > 100.0 else None,   # sampled by the timed launches themselves
            "us_per_step_at_16_steps_per_launch": us_per_step_k16}
    if counters:
        # All counters are means per LAUNCH of the kernel (one partition); a step is `parts` of them.
        # HBM bytes: 2 x FETCH_SIZE + WRITE_SIZE (KB) -- gfx950 tallies a wide streaming read at half its bytes
        # (MI355X_MICROARCH.md, HBM).  VALU issue: SQ_ACTIVE_INST_VALU counts quad-cycles in which a wave has a
        # vector instruction in issue; summed over waves / 1024 SIMDs = cycles a SIMD's vector issue is busy.
        # The step in shader cycles: its HIP-event time x the shader clock, the clock taken from the same counter pass
        # (SQ_BUSY_CYCLES / 32 shader engines over the duration of the serialised dispatches).
        traffic = parts * (2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0
        launch_cycles = counters["SQ_BUSY_CYCLES"] / N_SE
        serial_us = counters.get("SERIAL_NS", 0.0) / 1e3
        clock_counters = launch_cycles / serial_us if serial_us > 0 else None      # clock of the serialised (profiled) launches
        clock_timed = clock_mhz if clock_mhz and clock_mhz > 100.0 else None        # clock of the timed launches (probe)
        valu_cycles = parts * 4.0 * counters["SQ_ACTIVE_INST_VALU"] / N_SIMD
        if parts == 1:
            step_cycles = launch_cycles                  # one launch = one step: both counters of the same launches, no clock needed
            clock_mhz = clock_timed or step_cycles / (1e6 * step_s)
        else:
            clock_mhz = clock_timed or clock_counters
            step_cycles = 1e6 * step_s * clock_mhz
        roof.update({
            "bound": "valu_issue", "achieved": valu_cycles, "peak": step_cycles,
            "unit": "shader cycles per step: a SIMD's vector-issue-busy cycles, summed over the step's launches (achieved) vs the step (peak)",
            "frac": valu_cycles / step_cycles,
            "traffic": traffic, "hbm_actual_gbs": traffic / step_s / 1e9,
            "hbm_actual_frac": traffic / step_s / 1e9 / HBM_PEAK_GBS,
            "valu_wave_insts_per_launch": counters["SQ_INSTS_VALU"],
            "valu_wave_insts_per_step": parts * counters["SQ_INSTS_VALU"],
            "valu_trans_insts_per_launch": counters.get("SQ_INSTS_VALU_TRANS_F32"),
            "valu_cycles_per_wave_inst": 4.0 * counters["SQ_ACTIVE_INST_VALU"] / counters["SQ_INSTS_VALU"],
            "valu_issue_us": valu_cycles / clock_mhz,
            "shader_clock_mhz_during_launch": clock_mhz,
            "shader_clock_mhz_timed_run": clock_timed, "shader_clock_mhz_counter_pass": clock_counters,
            "kernel_serialised_us": serial_us or None,
            "kernel_avg_us": (counters["TRACE_AVG_NS"] / 1e3) if counters.get("TRACE_AVG_NS") else None,
            "kernel_trace_calls": counters.get("TRACE_CALLS"),
            "lds_bank_conflict_share": (counters["SQ_LDS_BANK_CONFLICT"] / counters["SQ_LDS_IDX_ACTIVE"])
                                       if counters.get("SQ_LDS_IDX_ACTIVE") else None,
            "note": "the kernel is bound by vector-instruction issue (and the LDS gathers behind it), not by HBM: "
                    "frac = SIMD vector-issue-busy cycles / step cycles from PMC counters of the same kernel on this box "
                    "(counter collection serialises the launches; with partitioned launches the step's cycles are its HIP-event "
                    "time x the shader clock the timed launches themselves sampled); hbm_actual_frac prices the "
                    "measured HBM bytes against 8 TB/s"})
    else:
        roof.update({"bound": "valu_issue", "achieved": None, "peak": None, "unit": "shader cycles per step",
                     "frac": None, "traffic": None, "note": "no counters available: " + str(counters_src)})
    return roof


def launch_object(info):
    return {"grid": info.gibbs_grid, "block": info.gibbs_block, "chains_per_tile": info.gibbs_seqs_per_tile,
            "lds_bytes": info.gibbs_lds_bytes, "table_group": info.group, "chain_parts": max(1, int(info.chain_parts)),
            "note": "geometry of ONE launch of the chain kernel; chain_parts of them (one per partition of the chains, each on a "
                    "stream of its own) make a step of the whole batch"}


def train_bound_note(name):
    what = {"cfg2": "the whole local phase is one launch, crbm_train_local (chain + model statistics + data statistics), which keeps a "
                    "SIMD's vector issue busy for about 0.77 of its time and moves about 21 MB of HBM per step",
            "cfg4": "chain launch (5 Gibbs steps), data-half and model-half statistics are three launches; the step moves a few hundred "
                    "MB of HBM (packed chains and letters)",
            "cfg5": "chain launch, data-half and model-half statistics are three launches of similar length; the step moves about "
                    "0.15 GB of HBM (packed chains and letters)"}.get(name, "")
    return ("equivalent-dense rate like roofline.algorithmic_frac, not a bandwidth utilisation: the step is bound by vector-instruction "
            "issue, not by HBM -- " + what + " (rocprofv3 PMC summaries per config under profiles/)")


def other_config(name, pmc, pmc_note, copy_gbs, steps=40, warmup=10):
    """Short timed run of another BASELINE configuration on this GPU (world size 1): chain launches and training steps on
    the driver's clock, with the same roofline object as the headline (counters from the in-run rocprofv3 children)."""
    from crbm_amd import _lib
    from crbm_amd._lib import fptr
    cfg = CONFIGS[name]
    k = cfg["k"]
    model = build_model(cfg, 1, 0, 0)
    h = model._h()
    lib = model._lib
    burn = max(1, 300 // k)
    for _ in range(burn):
        model._call("crbm_gibbs_steps_async", k)
    model._call("crbm_sync")
    ms = ctypes.c_float()
    model._call("crbm_time_gibbs", k, warmup, ctypes.byref(ms))
    model._call("crbm_sync")
    t0 = time.perf_counter()
    model._call("crbm_time_gibbs", k, steps, ctypes.byref(ms))
    wall = time.perf_counter() - t0
    clock = ctypes.c_float()
    model._call("crbm_last_shader_clock", ctypes.byref(clock))
    info = _lib.CrbmLaunchInfo()
    lib.crbm_get_launch_info(h, ctypes.byref(info))
    step_s = ms.value / 1e3 / steps
    out = {"workload": name + ": " + cfg["desc"], "launches": steps, "warmup_launches": warmup + burn,
           "gibbs_steps_per_launch": k,
           "ms_per_launch": 1e3 * wall / steps, "device_ms_per_launch": 1e3 * step_s,
           "gibbs_steps_per_s": k * steps / wall,
           "roofline": roofline_object(name, cfg, info, step_s, pmc, pmc_note, copy_gbs, int(lib.crbm_gibbs_state_bytes(h)),
                                       clock_mhz=float(clock.value)),
           "launch": launch_object(info)}
    n = cfg["chains"]
    D = synthetic_onehot(n, cfg["L"], seed=1234)
    model._call("crbm_dataset_upload", fptr(D), n, cfg["L"])
    tsteps, twarm = (40, 20) if name == "cfg4" else (100, 50)
    model._call("crbm_time_train", 0, n, twarm, ctypes.byref(ms))
    model._call("crbm_sync")
    t1 = time.perf_counter()
    model._call("crbm_time_train", 0, n, tsteps, ctypes.byref(ms))
    twall = time.perf_counter() - t1
    tb = n * (4 * 4 * cfg["L"] + k * algorithmic_bytes_per_seq(cfg["K"], cfg["M"], cfg["L"], cfg["ds"]))
    out["train"] = {"ms_per_train_step": ms.value / tsteps, "wall_ms_per_train_step": 1e3 * twall / tsteps,
                    "steps": tsteps, "warmup_steps": twarm, "cd_k": k, "global_batch": n, "all_reduce": "none",
                    "algorithmic_bytes_per_step_per_gpu": tb,
                    "algorithmic_frac": tb / (ms.value / tsteps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "bound_note": train_bound_note(name)}
    del model
    return out


def cfg1_fit_gpu():
    """BASELINE config #1 on the GPU: CRBM(10, 15).fit on 1000 synthetic 200-bp sequences with the reference's defaults
    (batchsize 20, cd_k 5, doublestranded; tutorial.py:13-16 without the data file), ONE epoch = 50 PCD-5 updates plus
    the per-epoch evaluation and status line.  The handle exists before the clock starts, as the reference compiles its
    Theano functions in the constructor (convRBM.py:175)."""
    import contextlib
    import io
    from crbm_amd import CRBM
    D = synthetic_onehot(1000, 200, seed=1234)
    out = {}
    for rep in ("first", "second"):
        np.random.seed(42)
        model = CRBM(10, 15, epochs=1, seed=2026)
        t0 = time.perf_counter()
        model._h()
        out["cfg1_create_seconds_%s" % rep] = time.perf_counter() - t0
        buf = io.StringIO()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(buf):
            model.fit(D)
        out["cfg1_fit_seconds_gpu" if rep == "second" else "cfg1_fit_seconds_gpu_first_call"] = time.perf_counter() - t0
        out["cfg1_status_line"] = [l for l in buf.getvalue().splitlines() if l.startswith("Epoch")][-1]
        del model
    out["cfg1_gpu_sample"] = ("one epoch of CRBM(10, 15).fit on 1000 x 200 bp, reference defaults (50 PCD-5 updates, evaluation of the "
                              "1000 rows in 50 batches, status line), float one-hot input as the reference takes it; the second of two "
                              "calls (the first pays one-off allocations and kernel loads)")
    return out


def generic_model_timing(K=300, M=10, ds=False, chains=4096, L=200, steps=20, warmup=5):
    """A model beyond the LDS-resident kernels (not a BASELINE configuration: reported beside them because the reference takes
    any model size, convRBM.py:72-108): Gibbs steps and PCD-1 training steps of a DNA model whose statistics and h|v run on
    the specialised kernels a slab of motifs at a time, v|h on the generic kernel (DESIGN.md section 5 "Slabs")."""
    from crbm_amd._lib import fptr
    cfg = dict(K=K, M=M, L=L, ds=ds, chains=chains, k=1)
    model = build_model(cfg, 1, 0, 0)
    model._h()
    D = synthetic_onehot(chains, L, seed=1234)
    model._call("crbm_dataset_upload", fptr(D), chains, L)
    model._call("crbm_train_step_resident", 0, chains)
    model.gibbsSteps(warmup)
    model._call("crbm_sync")
    t0 = time.perf_counter()
    model.gibbsSteps(steps)
    model._call("crbm_sync")
    gibbs_ms = 1e3 * (time.perf_counter() - t0) / steps
    for _ in range(warmup):
        model._call("crbm_train_step_resident", 0, chains)
    model._call("crbm_sync")
    t0 = time.perf_counter()
    for _ in range(steps):
        model._call("crbm_train_step_resident", 0, chains)
    model._call("crbm_sync")
    train_ms = 1e3 * (time.perf_counter() - t0) / steps
    del model
    return {"workload": "%d motifs len %d%s, %d chains x 4x%d, PCD-1 (beyond the LDS-resident kernels)" % (K, M, ", doublestranded" if ds else "", chains, L),
            "ms_per_gibbs_step": gibbs_ms, "ms_per_train_step": train_ms, "steps": steps, "warmup_steps": warmup, "clock": "host",
            "kernels": "statistics and h|v: the specialised kernels per slab of <= 60 motifs; v|h: generic kernel"}


PMC_PASSES = (("valu", "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"),
              ("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"))
N_SIMD = 256 * 4          # MI355X: 256 CUs x 4 SIMDs
N_SE = 32                 # SQ_BUSY_CYCLES is summed over the 32 shader engines


def pmc_of_csv(path, kernel_prefix):
    """mean per dispatch of every counter of the kernels whose name starts with `kernel_prefix`
    (rocprofv3 counter_collection.csv: one row per counter instance; the first quarter of the dispatches is dropped)"""
    import csv
    from collections import defaultdict
    acc = defaultdict(lambda: defaultdict(float))
    dur = {}
    for r in csv.DictReader(open(path)):
        if r["Kernel_Name"].startswith(kernel_prefix):
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            if r.get("End_Timestamp") and r.get("Start_Timestamp"):
                dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = {}
    for name, per in acc.items():
        ids = sorted(per, key=int)
        ids = ids[len(ids) // 4:]
        out[name] = sum(per[i] for i in ids) / len(ids)
        if name == "SQ_BUSY_CYCLES" and all(i in dur for i in ids):
            # duration of the same dispatches (counter collection serialises them): with SQ_BUSY_CYCLES the shader clock
            out["SERIAL_NS"] = sum(dur[i] for i in ids) / len(ids)
    return out


def measure_pmc(config, launches=24, timeout_s=150):
    """The chain kernel's counters on THIS box: one child `rocprofv3 --kernel-trace --pmc ...` per pass
    (FETCH_SIZE and WRITE_SIZE cannot share one, MI355X_MICROARCH.md), the program directly behind `--`.
    Must run before this process initialises the GPU.  Returns ({counter: mean per launch}, note)."""
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    vals, trace = {}, {}
    tmp = tempfile.mkdtemp(prefix="crbm_pmc_")
    try:
        # kernel trace alone (no counters: dispatches are not serialised, partitioned launches overlap as in the timed
        # run): average duration of ONE launch of the chain kernel, as `rocprofv3 --kernel-trace --stats` reports it
        d = os.path.join(tmp, "trace")
        cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "t", "--",
               sys.executable, os.path.join(ROOT, "tools", "prof_gibbs.py"), config, str(8 * launches)]
        try:
            r = subprocess.run(cmd, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               timeout=timeout_s)
            import csv
            for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Name"].startswith("crbm_gibbs"):
                        trace = {"TRACE_AVG_NS": float(row["AverageNs"]), "TRACE_CALLS": float(row["Calls"])}
        except Exception:
            trace = {}
        for tag, counters in PMC_PASSES:
            d = os.path.join(tmp, tag)
            cmd = [exe, "--kernel-trace", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "tools", "prof_gibbs.py"), config, str(launches)]
            try:
                r = subprocess.run(cmd, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), stdout=subprocess.PIPE,
                                   stderr=subprocess.STDOUT, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 pass '%s' timed out" % tag
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 pass '%s' failed (rc %d)" % (tag, r.returncode)
            for f in files:
                vals.update(pmc_of_csv(f, "crbm_gibbs"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    need = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES")
    if any(k not in vals for k in need):
        return None, "rocprofv3 output lacks %s" % [k for k in need if k not in vals]
    vals.update(trace)
    return vals, "measured in this run: rocprofv3 --kernel-trace --pmc (passes: %s) of tools/prof_gibbs.py %s %d" % (
        " | ".join(c for _, c in PMC_PASSES), config, launches)


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv):
    """Start n ranks of this script (nothing in this process has touched a GPU), relay rank 0's
    stdout, fail if any rank fails."""
    port = free_port()
    secret = os.urandom(16).hex()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("CRBM_JOB_SECRET", secret)        # keys the control plane's message authentication
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out = b""
    deadline = time.time() + SPAWN_TIMEOUT_S
    try:
        out, _ = procs[0].communicate(timeout=SPAWN_TIMEOUT_S)
        for p in procs[1:]:
            p.wait(timeout=max(1.0, deadline - time.time()))
    except subprocess.TimeoutExpired:
        pass
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.stdout.write(out.decode(errors="replace"))
    sys.stdout.flush()
    codes = [p.returncode if p.returncode is not None else 124 for p in procs]
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect the chain kernel's PMC counters in child runs")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of configs #4, #5 and #1 behind the headline")
    ap.add_argument("--multi-step", action="store_true", help="also time 16 Gibbs steps per launch (informational)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    cfg = CONFIGS[args.config]
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # CRBM_BENCH_SHARE_GPU=1 rehearses the N > 1 plumbing on a box with ONE GPU: all ranks share
    # device 0 and the training section runs without the RCCL communicator (which refuses two
    # ranks on one device).  The driver's runs never set it.
    share_gpu = os.environ.get("CRBM_BENCH_SHARE_GPU", "0") == "1"
    device = 0 if share_gpu else local_rank
    if share_gpu and world > 1:
        # N processes on ONE GPU oversubscribe its hardware queues once every rank brings a partition stream of its own
        # (six ranks: 0.075 -> 0.40 ms per step, the mapped-buffer all-reduce 7 us -> 11 ms of time slicing): the rehearsal
        # keeps the chain launches whole.  Real multi-GPU runs (one GPU per rank) partition as a single rank does.
        os.environ.setdefault("CRBM_CHAIN_PARTS", "1")

    # counters of the chain kernel on this box (children under rocprofv3, before this process touches the GPU)
    pmc, pmc_note = None, "not collected (multi-rank run, or --no-pmc)"
    others = [c for c in ("cfg4", "cfg5") if world == 1 and args.config == "cfg2" and not args.no_other_configs]
    other_pmc = {}
    if world == 1 and not args.no_pmc and os.environ.get("CRBM_BENCH_PMC", "1") != "0":
        try:
            pmc, pmc_note = measure_pmc(args.config)
        except Exception as e:                      # a report, never a reason to lose the line
            pmc, pmc_note = None, "rocprofv3 child failed: %s" % str(e)[:200]
        for c in others:
            try:
                other_pmc[c] = measure_pmc(c, launches=8 if c == "cfg4" else 16)
            except Exception as e:
                other_pmc[c] = (None, "rocprofv3 child failed: %s" % str(e)[:200])

    from crbm_amd import _lib
    from crbm_amd import dist as cdist
    from crbm_amd._lib import fptr
    control = cdist.ControlPlane(rank, world)
    model = build_model(cfg, world, rank, device)
    h = model._h()
    lib = model._lib
    k = cfg["k"]

    def barrier():
        model._call("crbm_wait_idle")     # this rank's streams are idle ...
        control.barrier()                 # ... and so are everybody else's

    copy_gbs = ctypes.c_float()
    if os.environ.get("CRBM_BENCH_COPY_FIRST", "0") == "1":      # (A/B switch: the copy test in front of the timed region, as until round 3)
        model._call("crbm_copy_bandwidth", 1 << 30, 5, ctypes.byref(copy_gbs))
    # chains start at h = 0 (convRBM.py:168): 3000 burn-in steps (55 ms at config #2), then the W warm-up launches.
    # The burn-in is part of building the workload, not of the measurement: a persistent chain is never at
    # h = 0 in training -- and it is what brings the GPU to the clock it holds under load: the part steps between
    # ~2.14 and ~2.39 GHz with a few milliseconds of hysteresis (tools/ramp_probe.py prints the clock the launches
    # themselves sample: 2.14 GHz in 20-launch windows right behind 500 steps or 10 ms of idling, 2.39 GHz behind a
    # 2000-launch window), and the driver's timed region is 0.4 ms long.  With 3000 steps enqueued by the library's
    # own loop the 20-launch window runs at 2.35 GHz and 19.1-20.4 us per step (gpurun_out/r4_tests/burnin2.txt);
    # a window that short still carries ~2 us per step of start and tail (the first launches of the two partitions
    # have nothing to overlap with, the last one runs alone), the steady state of a 2000-launch run is 17.2 us.
    # warmup_effective in the line counts the burn-in; shader_clock_mhz_timed_run is the clock of the timed launches.
    # (as launches of k steps like the timed ones: every launch of the chain kernel in this process is then
    # the same work, and a profiler's per-kernel average is that of the timed launches)
    burn_launches = max(1, int(os.environ.get("CRBM_BENCH_BURNIN", "3000")) // k)
    burn_ms = ctypes.c_float()
    if os.environ.get("CRBM_BENCH_BURNIN_PY", "0") == "1":        # (A/B switch: one Python call per burn-in launch, as until round 3)
        for _ in range(burn_launches):
            model._call("crbm_gibbs_steps_async", k)
        model._call("crbm_sync")
    else:
        # enqueued by the library's own loop: a Python call per launch (10-15 us of host time) does not keep up with 17-us
        # steps, and a GPU that waits for its host between launches never settles at its sustained clock
        model._call("crbm_time_gibbs", k, burn_launches, ctypes.byref(burn_ms))
    if args.warmup > 0:
        model._call("crbm_time_gibbs", k, args.warmup, ctypes.byref(burn_ms))

    # ---- timed region: exactly K launches, HIP events on the library's stream ----
    total_ms = ctypes.c_float()
    barrier()
    t0 = time.perf_counter()
    model._call("crbm_time_gibbs", k, args.steps, ctypes.byref(total_ms))     # returns once its last launch has completed
    wall = time.perf_counter() - t0          # this rank's K steps, start barrier -> own stream idle; the closing barrier
    barrier()                                # (TCP round trips, a blocking read-back of the activity monitor) is not step time
    clock_mhz = ctypes.c_float()
    model._call("crbm_last_shader_clock", ctypes.byref(clock_mhz))     # sampled by the timed launches themselves
    wall, kernel_s = control.allreduce_max([wall, total_ms.value / 1e3])
    launches = args.steps
    steps_done = args.steps * k                  # Gibbs steps per rank in the timed region
    value = world * steps_done / wall            # host clock (barrier + synchronise on both sides), max over ranks
    value_device = world * steps_done / kernel_s

    # informational, opt-in (--multi-step): the same chain advanced 16 steps per launch (no
    # per-launch fixed cost).  Off by default so that every crbm_gibbs launch of a default
    # run is a one-step launch (the rocprofv3 per-kernel average then equals avg_launch_us).
    us_per_step_k16 = None
    if args.multi_step:
        multi_ms = ctypes.c_float()
        model._call("crbm_time_gibbs", 16, 5, ctypes.byref(multi_ms))
        model._call("crbm_time_gibbs", 16, 40, ctypes.byref(multi_ms))
        us_per_step_k16 = 1e3 * multi_ms.value / (40 * 16)

    # device-copy bandwidth of this GPU (the measured ceiling quoted beside the 8 TB/s spec) -- behind the timed region
    if copy_gbs.value == 0.0:
        model._call("crbm_copy_bandwidth", 1 << 30, 5, ctypes.byref(copy_gbs))

    # hidden-unit activity of the chain (workload descriptor, after timing)
    hf, _ = model.get_fantasy()
    activity = float(hf.mean())

    out = None
    if rank == 0:
        info = _lib.CrbmLaunchInfo()
        lib.crbm_get_launch_info(h, ctypes.byref(info))
        roof = roofline_object(args.config, cfg, info, kernel_s / launches, pmc, pmc_note, float(copy_gbs.value),
                               int(lib.crbm_gibbs_state_bytes(h)), us_per_step_k16, clock_mhz=float(clock_mhz.value))
        out = {
            "metric": "Gibbs-steps/sec (PCD-1) at batch 8192x4x200, 10 motifs len 15" if args.config == "cfg2"
                      else "Gibbs-steps/sec, " + cfg["desc"],
            "value": value, "unit": "8192-chain batch Gibbs steps/s (summed over GPUs)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * wall / args.steps,
            "device_ms_per_step": 1e3 * kernel_s / args.steps, "value_device": value_device,
            "metric_version": 3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "warmup_effective": args.warmup + burn_launches, "chain_burn_in_launches": burn_launches,
            "config": {"workload": args.config + ": " + cfg["desc"], "chains_per_gpu": cfg["chains"],
                       "gibbs_steps_per_launch": k, "visible": "4x%d" % cfg["L"],
                       "hidden": "%dx%d" % (cfg["K"], cfg["L"] - cfg["M"] + 1),
                       "parallelism": "chains sharded over %d GPU(s), no collective in the Gibbs step" % world,
                       "hidden_activity": activity},
            "timing": "value and ms_per_step: host clock of each rank from the opening barrier (all streams idle) until its own K "
                      "launches have completed (event synchronise), max over ranks, closing barrier behind it (metric_version 3; version 2 = round 2 quoted the HIP-event time, now device_ms_per_step / "
                      "value_device; version 1 = round 1, host clock).  warmup_effective counts the chain burn-in launches "
                      "(chains start at h = 0) together with the --warmup launches",
            "chain_steps_per_s": value * cfg["chains"],
            "roofline": roof,
            "launch": launch_object(info),
        }

    # The headline line is complete before the secondary section starts.  With N > 1 a
    # watchdog prints it anyway if the RCCL part of the training section should hang on
    # some rank (a hang must not cost the Gibbs measurement) -- and exits non-zero.
    done = {"printed": False}

    def emit(train_result):
        if done["printed"]:
            return
        done["printed"] = True
        if rank == 0:
            out["train"] = train_result
            if others:
                # every BASELINE configuration that fits one GPU on the same clock (each in its own try: a report)
                out["other_configs"] = {}
                for c in others:
                    try:
                        cp, cn = other_pmc.get(c, (None, "not collected (--no-pmc)"))
                        out["other_configs"][c] = other_config(c, cp, cn, float(copy_gbs.value))
                    except Exception as e:
                        out["other_configs"][c] = {"error": str(e)[:300]}
                try:
                    out["other_configs"]["cfg1"] = cfg1_fit_gpu()
                except Exception as e:
                    out["other_configs"]["cfg1"] = {"error": str(e)[:300]}
                try:
                    out["other_configs"]["generic_300x10"] = generic_model_timing()
                except Exception as e:
                    out["other_configs"]["generic_300x10"] = {"error": str(e)[:300]}
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(cfg)
            print(json.dumps(out), flush=True)

    watchdog = None
    progress = {}
    if world > 1 and not args.no_train:
        def bail():
            # what the section had measured when it stopped making progress is kept (e.g. the RCCL steps when the
            # comparison run of the mapped-buffer all-reduce behind them hangs)
            emit(dict(progress, error="training section did not finish within %d s" % TRAIN_WATCHDOG_S))
            # the primary measurements (Gibbs steps; RCCL training steps) stand if only the comparison behind them hung
            os._exit(0 if "all_reduce_us" in progress else 3)
        watchdog = threading.Timer(TRAIN_WATCHDOG_S, bail)
        watchdog.daemon = True
        watchdog.start()

    # ---- secondary: full PCD-k training steps (with the RCCL all-reduce when N > 1) ----
    train = None
    failed = False
    if not args.no_train:
        try:
            rccl = world > 1 and not share_gpu
            n = cfg["chains"]
            D = synthetic_onehot(n, cfg["L"], seed=1234 + rank)
            model._call("crbm_dataset_upload", fptr(D), n, cfg["L"])
            tms = ctypes.c_float()
            # 200 timed steps after 50: the hidden activity of the randomly initialised model settles within ~50 steps
            # (config #2: 79 us per step over the first ten, 70.5 over steps 40-50), and a timed region of a few
            # milliseconds still contains the GPU's ramp out of idle (tools/train_trajectory.py: 70 us per step in
            # 10-step regions against 65.5 in a 400-step one)
            tsteps, twarm = 200, 50

            def timed_steps():
                model._call("crbm_time_train", 0, n, twarm, ctypes.byref(tms))
                barrier()
                t1 = time.perf_counter()
                model._call("crbm_time_train", 0, n, tsteps, ctypes.byref(tms))
                dt = time.perf_counter() - t1
                barrier()
                return control.allreduce_max([dt, tms.value / 1e3])

            local_dev = rccl_init_s = allreduce_us = None
            if world > 1:
                # the same steps first WITHOUT any all-reduce (every rank for itself; the replicas drift apart and are
                # made one model again below): the difference to the steps with it is what the all-reduce adds to the
                # critical path of a step
                _, local_dev = timed_steps()
            if rccl:
                # the 128-byte RCCL id travels over the control plane
                t2 = time.perf_counter()
                uid = cdist.exchange_unique_id(rank, world, control=control)
                buf = (ctypes.c_uint8 * _lib.UNIQUE_ID_BYTES).from_buffer_copy(uid)
                model._call("crbm_comm_init", buf, world, rank)
                model._call("crbm_comm_broadcast_state", 0)
                rccl_init_s = control.allreduce_max([time.perf_counter() - t2])[0]
            stats_note = ("P enters the MFMA contraction as two f16 halves (22 significant bits; fp32 has 24), "
                          "accumulation in fp32; the chain itself computes in f32")
            if world == 1 or rccl:
                twall, tdev = timed_steps()
                train = {"train_steps_per_s": tsteps / tdev, "global_batch": n * world, "cd_k": k,
                         "all_reduce": "rccl" if rccl else "none", "ms_per_train_step": 1e3 * tdev / tsteps,
                         "wall_ms_per_train_step": 1e3 * twall / tsteps, "steps": tsteps, "warmup_steps": twarm,
                         "statistics_dtype": stats_note}
            if rccl:
                ams = ctypes.c_float()
                model._call("crbm_time_allreduce", 10, ctypes.byref(ams))
                barrier()
                model._call("crbm_time_allreduce", 100, ctypes.byref(ams))
                allreduce_us = control.allreduce_max([1e3 * ams.value / 100])[0]
                train.update({"ms_per_train_step_without_all_reduce": 1e3 * local_dev / tsteps,
                              "all_reduce_us": 1e6 * (tdev - local_dev) / tsteps,
                              "all_reduce_alone_us": allreduce_us, "rccl_init_s": rccl_init_s,
                              "all_reduce_note": "all_reduce_us = device time of a step with the communicator minus the same step "
                                                 "without it (what the collective adds to the critical path); all_reduce_alone_us = "
                                                 "back-to-back ncclAllReduce launches of the %d-float sums buffer, HIP events"
                                                 % int(lib.crbm_sums_count(h))})
                sums = control.gather(cdist.replica_checksum(model))
                train["replicas_identical"] = len(set(sums)) == 1
                failed = failed or not train["replicas_identical"]
                progress.update(train)
                model._call("crbm_comm_destroy")
            if world > 1 and os.environ.get("CRBM_BENCH_IPC", "1") != "0":
                # the all-reduce through mapped buffers (crbm_ipc_*): the ranks map each other's sums buffers and the update
                # launch adds them -- no collective launch.  Works with all ranks on one GPU too (the rehearsal), where RCCL
                # refuses; on a real node it is timed beside RCCL so that the two can be compared.
                ipc = {}
                try:
                    # identical replicas again: the filters every rank was built with, velocities at zero
                    model.motifs.set_value(np.random.default_rng(42).standard_normal((cfg["K"], 1, 4, cfg["M"])).astype(np.float32))
                    model.bias.set_value(model._host["bias"])
                    model.c.set_value(np.zeros((1, 4), dtype=np.float32))
                    model.set_velocities(np.zeros((cfg["K"], 1, 4, cfg["M"]), np.float32), np.zeros((1, cfg["K"]), np.float32),
                                         np.zeros((1, 4), np.float32))
                    t3 = time.perf_counter()
                    cdist.attach_ipc(model, control)
                    ipc["init_s"] = control.allreduce_max([time.perf_counter() - t3])[0]
                    iwall, idev = timed_steps()
                    ipc.update({"ms_per_train_step": 1e3 * idev / tsteps, "wall_ms_per_train_step": 1e3 * iwall / tsteps,
                                "all_reduce_us": 1e6 * (idev - local_dev) / tsteps,
                                "timed_out": bool(max(control.gather(int(cdist.ipc_timed_out(model)))))})
                    sums = control.gather(cdist.replica_checksum(model))
                    ipc["replicas_identical"] = len(set(sums)) == 1
                    ipc["ok"] = ipc["replicas_identical"] and not ipc["timed_out"]
                except Exception as e:
                    ipc["error"] = str(e)[:300]
                    ipc["ok"] = False
                # beside RCCL this form is a comparison, reported with its own flag; where it is the only all-reduce that
                # can run (all ranks on one GPU) its failure is the training section's
                failed = failed or (not rccl and not ipc["ok"])
                if train is None:               # all ranks on one GPU: the IPC form is the only all-reduce that can run
                    train = {"train_steps_per_s": (tsteps / idev) if "ms_per_train_step" in ipc else None, "global_batch": n * world,
                             "cd_k": k, "all_reduce": "ipc (mapped buffers, no collective launch)",
                             "ms_per_train_step": ipc.get("ms_per_train_step"), "wall_ms_per_train_step": ipc.get("wall_ms_per_train_step"),
                             "ms_per_train_step_without_all_reduce": 1e3 * local_dev / tsteps, "steps": tsteps,
                             "statistics_dtype": stats_note}
                train["ipc_all_reduce"] = ipc
            if train is None:
                train = {"error": "no all-reduce available (CRBM_BENCH_SHARE_GPU with CRBM_BENCH_IPC=0)", "global_batch": n * world}
            if train.get("ms_per_train_step"):
                # SURVEY 8(d), train-step bytes in the reference's dense fp32 layout: the data batch read once, k Gibbs
                # steps, the model statistics fused (no extra traffic) -- per GPU
                S = 2 if cfg["ds"] else 1
                tb = n * (4 * 4 * cfg["L"] + k * algorithmic_bytes_per_seq(cfg["K"], cfg["M"], cfg["L"], cfg["ds"]))
                train["algorithmic_bytes_per_step_per_gpu"] = tb
                train["algorithmic_frac"] = tb / (train["ms_per_train_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                train["bound_note"] = train_bound_note(args.config)
        except Exception as e:                      # report, never hide: the headline is the Gibbs metric
            train = {"error": str(e)[:300]}
            failed = True

    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        out["secondary_ok"] = not failed        # the training section is secondary: its failure is reported in the line
    emit(train)                                 # (train.error) and through the flag; the headline above stays valid
    control.close()
    # A failed training section (diverged replicas, a time-out of the mapped-buffer all-reduce, an exception) of a
    # multi-rank run is a failed run: the line above is complete and says what happened (secondary_ok, train.error),
    # the exit code says it to whoever only looks at that.  One-GPU runs keep exit 0 for the Gibbs headline unless
    # CRBM_BENCH_STRICT=1 (the training section is secondary there and has no collective to get wrong).
    if failed and (world > 1 or os.environ.get("CRBM_BENCH_STRICT", "0") == "1") and os.environ.get("CRBM_BENCH_STRICT", "") != "0":
        sys.exit(4)


if __name__ == "__main__":
    main()
