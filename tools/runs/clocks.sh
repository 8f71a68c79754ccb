# Samples the engine clock while the chain kernel runs back to back (is the kernel power- or clock-limited?)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/clocks
mkdir -p $O
rocm-smi --showclocks > $O/idle.txt 2>&1
( KS=$(python3 -c "print(','.join(['16']*300))") python tools/gibbs_k_scan.py cfg2 > $O/scan.txt 2>&1 ) &
BG=$!
sleep 6
for i in 1 2 3 4 5 6 7 8 9 10; do rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power\|mclk" | head -6 >> $O/busy.txt; echo "--" >> $O/busy.txt; sleep 1; done
wait $BG
cat $O/idle.txt | grep -i "sclk\|mclk" | head -4; echo ==; grep -i 'sclk\|Power (W)' $O/busy.txt | head -24; tail -3 $O/scan.txt
