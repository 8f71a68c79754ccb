"""Copies the evidence collected by tools/runs/evidence.sh (gpurun_out/<tag>/) into profiles/ under
round-prefixed names and regenerates profiles/gibbs_traffic.json (what bench.py quotes as
roofline.traffic) from the PMC passes of the Gibbs-only run.
usage: python tools/make_profiles.py gpurun_out/<tag> r02"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

src, rnd = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles")


def cp(a, b):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, "%s_%s" % (rnd, b)))
        print("profiles/%s_%s" % (rnd, b))


cp("bench_cfg2.json", "bench_cfg2.json")
cp("bench_cfg4.json", "bench_cfg4.json")
cp("bench_cfg5.json", "bench_cfg5.json")
cp("bench_cfg2_under_rocprof.json", "bench_cfg2_under_rocprof.json")
cp("prof_bench/bench_kernel_stats.csv", "bench_cfg2_kernel_stats.csv")
cp("gibbs_steps_per_launch_scan.txt", "gibbs_steps_per_launch_scan.txt")
cp("pmc_summary.txt", "pmc_summary.txt")
cp("pytest.log", "gpu_tests.log")

cp("bench_driver.json", "bench_cfg2_driver_style_20_steps.json")
cp("bench_6rank_one_gpu_rehearsal.json", "bench_cfg2_6rank_one_gpu_rehearsal.json")
cp("bench_cfg5_4rank_one_gpu_rehearsal.json", "bench_cfg5_4rank_one_gpu_rehearsal.json")
cp("floor_scan.txt", "gibbs_launch_floor_scan.txt")
cp("lds_conflict_phases.txt", "stats_lds_conflicts_by_phase.txt")
cp("trace_train.txt", "train_step_kernel_trace.txt")
# round 4
cp("ramp_probe.txt", "ramp_probe.txt")
cp("fit_cfg2.txt", "fit_cfg2.txt")
cp("sweep_1M_cfg2model.json", "sweep_1M_cfg2model.json")
cp("smoke.log", "smoke.log")
cp("timeline_20.txt", "chain_launch_timeline_20_steps.txt")
cp("bench_2rank.json", "bench_cfg2_2rank_one_gpu_rehearsal.json")
cp("bench_6rank.json", "bench_cfg2_6rank_one_gpu_rehearsal.json")
cp("bench_cfg5_4rank.json", "bench_cfg5_4rank_one_gpu_rehearsal.json")
cp("bench_2rank_torchrun.json", "bench_cfg2_2rank_torchrun_one_gpu_rehearsal.json")

# Counters of the chain kernel per config (what bench.py quotes when it cannot collect them itself): mean per launch
# of every counter of the passes of tools/prof_gibbs.py (FETCH_SIZE and WRITE_SIZE in passes of their own)
out = {}
for cfg in ("cfg2", "cfg4", "cfg5"):
    vals = {}
    for p in "abfw":
        for f in glob.glob(os.path.join(src, "pmc_gibbs_%s_%s" % (cfg, p), "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(float))
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"].startswith("crbm_gibbs"):
                    acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            for name in acc:
                ids = sorted(acc[name], key=int)
                ids = ids[len(ids) // 4:]
                vals[name] = sum(acc[name][i] for i in ids) / len(ids)
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        parts = 1
        try:      # partitions of a chain launch at this config: the counters are per LAUNCH (one partition)
            line = json.loads(open(os.path.join(src, "bench_%s.json" % cfg)).read().strip().splitlines()[-1])
            parts = int(line["launch"].get("chain_parts", 1))
        except Exception:
            pass
        out[cfg] = {
            "counters": vals,
            "chain_parts": parts,
            "hbm_bytes_per_launch": int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024),
            "correction": "gfx950 FETCH_SIZE reports 1/2 of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM): doubled",
            "source": "rocprofv3 --pmc passes of python3 tools/prof_gibbs.py %s on another box (tools/runs/evidence.sh; "
                      "profiles/%s_pmc_summary.txt)" % (cfg, rnd),
        }
json.dump(out, open(os.path.join(dst, "gibbs_traffic.json"), "w"), indent=1)
print("profiles/gibbs_traffic.json", sorted(out))
