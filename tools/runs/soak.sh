# Longer randomised parity soak (tools/soak_parity.py <shapes> <seed>) on one MI355X
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/soak
mkdir -p $O
timeout -k 10 1100 python tools/soak_parity.py ${1:-240} ${2:-31} > $O/soak_${1:-240}_${2:-31}.txt 2>&1; echo "soak rc=$?"; grep -v ": ok" $O/soak_${1:-240}_${2:-31}.txt | tail -12; grep -c ": ok" $O/soak_${1:-240}_${2:-31}.txt
