cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_cfg1; mkdir -p $O
timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/prof_cfg1.py 10 2>&1 | tail -1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python3 $GRAFT_REPO_ROOT/tools/prof_cfg1.py 5 > $O/log.txt 2>&1
tail -1 $O/log.txt
f=$(find $O/t -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("   %-60s calls %5s avg %9.2f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
