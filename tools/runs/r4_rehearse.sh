# round 4: the driver's multi-rank commands rehearsed on one GPU (all ranks share device 0; plumbing evidence, not scaling numbers)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_rehearse; mkdir -p $O
( time CRBM_BENCH_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 ) > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "2-rank rc=$?"; tail -2 $O/bench_2rank.err
( time CRBM_BENCH_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus 6 --steps 20 --warmup 5 ) > $O/bench_6rank.json 2> $O/bench_6rank.err; echo "6-rank rc=$?"; tail -2 $O/bench_6rank.err
( time CRBM_BENCH_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus 4 --config cfg5 --steps 20 --warmup 5 ) > $O/bench_cfg5_4rank.json 2> $O/bench_cfg5_4rank.err; echo "cfg5 4-rank rc=$?"; tail -2 $O/bench_cfg5_4rank.err
# the same through the launcher the driver uses (torch is only the launcher here)
( time CRBM_BENCH_SHARE_GPU=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 ) > $O/bench_2rank_torchrun.json 2> $O/bench_2rank_torchrun.err; echo "torchrun 2-rank rc=$?"; tail -2 $O/bench_2rank_torchrun.err
python - <<PY
import json
for f in ("bench_2rank", "bench_6rank", "bench_cfg5_4rank", "bench_2rank_torchrun"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        t = d["train"]
        print(f, "n_gpus", d["n_gpus"], "value %.0f" % d["value"], "ms/step %.4f" % d["ms_per_step"], "secondary_ok", d["secondary_ok"],
              "train", t.get("all_reduce"), t.get("ms_per_train_step"), "ipc", {k: t.get("ipc_all_reduce", {}).get(k) for k in ("ok", "timed_out", "replicas_identical", "all_reduce_us", "error")})
    except Exception as e:
        print(f, "ERR", e)
PY
