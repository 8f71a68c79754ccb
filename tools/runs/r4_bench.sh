# round 4: bench lines (driver style and default) after the partitioned launches
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_bench; mkdir -p $O
( time timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench driver-style rc=$?"; tail -3 $O/bench_driver.err
python - <<PY
import json
d = json.loads(open("$O/bench_driver.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms_per_step", d["ms_per_step"], "device", d["device_ms_per_step"])
print({k: r.get(k) for k in ("avg_launch_us", "kernel_avg_us", "kernel_serialised_us", "launches_per_step", "frac", "algorithmic_frac", "hbm_actual_frac", "valu_wave_insts_per_step", "shader_clock_mhz_during_launch", "counters_source")})
print("train", {k: d["train"].get(k) for k in ("ms_per_train_step", "algorithmic_frac", "error")})
for c, o in d.get("other_configs", {}).items():
    if "roofline" in o:
        print(c, o["device_ms_per_launch"], o["ms_per_launch"], {k: o["roofline"].get(k) for k in ("frac", "algorithmic_frac", "kernel_avg_us", "launches_per_step")}, o["train"]["ms_per_train_step"], o["train"]["algorithmic_frac"])
    else:
        print(c, o)
print("cpu", d.get("cpu_baseline"))
PY
( time timeout -k 10 500 python3 bench.py --no-other-configs ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench default rc=$?"; cut -c1-400 $O/bench_cfg2.json
for n in 20 50 200; do echo "steps=$n: $(timeout -k 10 200 python3 bench.py --steps $n --warmup 5 --no-pmc --no-cpu-baseline --no-train --no-other-configs | cut -c1-330)"; done
