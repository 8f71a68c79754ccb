# rocprofv3 per-kernel stats of N training steps of a bench config.  usage: bash tools/runs/trace_train.sh <tag> <cfg> [steps]
TAG=${1:-trace}; CFG=${2:-cfg2}; N=${3:-200}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${CFG} -o t -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py $CFG $N > $O/trace_${CFG}.log 2>&1
echo "== $CFG: $(grep 'us/train' $O/trace_${CFG}.log)"
f=$(find $O/trace_${CFG} -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("   %-40s calls %6s avg %9.2f us  min %9.2f max %9.2f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
