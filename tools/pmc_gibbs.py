"""Summarise rocprofv3 --pmc passes of the Gibbs kernel: mean counter value per launch.
usage: python tools/pmc_gibbs.py <dir with *counter_collection.csv> [kernel-name-prefix]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else "crbm_gibbs"
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    acc = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith(prefix):
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    print(f)
    for name in sorted(acc):
        ids = sorted(acc[name], key=int)[1:]          # the first launch is the 10-step burn-in of prof_gibbs.py
        vals = [acc[name][i] for i in ids]
        print("   %-26s mean per launch = %.5g  (%d launches)" % (name, sum(vals) / len(vals), len(vals)))
