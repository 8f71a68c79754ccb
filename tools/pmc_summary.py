"""Summarises the rocprofv3 --pmc passes that tools/runs/evidence.sh collects: mean counter value per
launch for every kernel of the training step and of the Gibbs-only run, per bench config.
usage: python tools/pmc_summary.py <evidence dir>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
KERNELS = ("crbm_train_local", "crbm_gibbs_sparse_stats", "crbm_gibbs_sparse", "crbm_gibbs", "crbm_stats_mfma_data", "crbm_stats_mfma_model",
           "crbm_update_tables", "crbm::reduce_partials_pair_kernel")
for cfg in ("cfg2", "cfg4", "cfg5"):
    for run in ("train", "gibbs"):
        acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))      # kernel -> counter -> dispatch -> value
        for p in "abfw":
            for f in glob.glob(os.path.join(root, "pmc_%s_%s_%s" % (run, cfg, p), "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    name = r["Kernel_Name"]
                    key = next((k for k in KERNELS if name == k or name.startswith(k + "(")), None)
                    if key:
                        acc[key][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        if not acc:
            continue
        print("== %s, %s run (tools/prof_%s.py %s)" % (cfg, run, run, cfg))
        for k in KERNELS:
            if k not in acc:
                continue
            print("  %s" % k)
            for cname in sorted(acc[k]):
                ids = sorted(acc[k][cname], key=int)
                ids = ids[len(ids) // 4:]                      # skip the warm-up launches
                vals = [acc[k][cname][i] for i in ids]
                print("     %-26s mean per launch = %.5g  (%d launches)" % (cname, sum(vals) / len(vals), len(vals)))
            c = acc[k]
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                f_ = sum(c["FETCH_SIZE"].values()) / len(c["FETCH_SIZE"])
                w_ = sum(c["WRITE_SIZE"].values()) / len(c["WRITE_SIZE"])
                print("     HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB = %d   (FETCH_SIZE doubled: gfx950 counts 128-B reads "
                      "at 64 B, MI355X_MICROARCH.md)" % int((2 * f_ + w_) * 1024))
