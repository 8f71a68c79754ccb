cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_trace20; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-train --no-other-configs > $O/line.json 2> $O/err.txt
cut -c1-300 $O/line.json
python3 - <<PY
import csv, glob
f = glob.glob("$O/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("crbm_gibbs")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-60:]
t0 = int(last[0]["Start_Timestamp"])
print("last 60 chain launches: queue start_us end_us dur_us")
for r in last:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(r["Queue_Id"], "%.1f %.1f %.1f" % (s / 1e3, e / 1e3, (e - s) / 1e3))
PY
