cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_final; mkdir -p $O
( time timeout -k 10 500 python bench.py ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench rc=$?"; tail -4 $O/bench_cfg2.err
( time timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver-style rc=$?"; tail -4 $O/bench_driver.err
python - <<'PY'
import json
for f in ("bench_cfg2.json", "bench_driver.json"):
    d = json.loads(open("gpurun_out/r4_final/" + f).read().strip().splitlines()[-1])
    oc = d.get("other_configs", {})
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("train", {}).get("ms_per_step") if isinstance(d.get("train"), dict) else None,
          {k: (v.get("cfg1_fit_seconds_gpu") if k == "cfg1" else v.get("value")) for k, v in oc.items()} if isinstance(oc, dict) else oc)
PY
