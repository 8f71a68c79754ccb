cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_profbig; mkdir -p $O
IFS=";"; for shape in ${SHAPES:-300 10 0 256 200;120 40 1 256 200;300 10 0 4096 200}; do
  IFS=" "; set -- $shape
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$1_$4 -o t -- python3 $GRAFT_REPO_ROOT/tools/prof_big.py $shape 10 > $O/log_$1_$4.txt 2>&1
  cat $O/log_$1_$4.txt | tail -2
  f=$(find $O/t_$1_$4 -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("   %-46s calls %5s avg %9.2f us" % (r["Name"][:46], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
