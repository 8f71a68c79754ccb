cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_enq; mkdir -p $O
{
for p in 0 1; do echo "== CRBM_CHAIN_PARTS=$p"; CRBM_CHAIN_PARTS=$p timeout -k 10 200 python3 tools/ramp_probe.py cfg2; done
echo "== parts=2 without stagger"; CRBM_PART_STAGGER=0 timeout -k 10 200 python3 tools/ramp_probe.py cfg2
} 2>&1 | tee $O/ramp.txt
