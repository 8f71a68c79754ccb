# rocprofv3 per-kernel stats of N training steps, default tail vs CRBM_TAIL=split.  usage: bash tools/runs/trace_train.sh <tag> [cfg]
TAG=${1:-trace}; CFG=${2:-cfg2}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in default split; do
  if [ $v = split ]; then export CRBM_TAIL=split; else unset CRBM_TAIL; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${CFG}_$v -o t -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py $CFG 200 > $O/trace_${CFG}_$v.log 2>&1
  echo "== $v: $(tail -1 $O/trace_${CFG}_$v.log)"
  f=$(find $O/trace_${CFG}_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("   %-40s calls %6s avg %9.2f us  min %9.2f max %9.2f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
unset CRBM_TAIL
