cd $GRAFT_REPO_ROOT
for dbg in 0 1 2 4 7; do
  echo "gibbs debug=$dbg"; CRBM_GIBBS_DEBUG=$dbg KS=0,1 python tools/gibbs_k_scan.py cfg2 2>&1 | tail -2
done
for g in 512 768 2048; do
  echo "grid=$g"; CRBM_GIBBS_GRID=$g KS=1 python tools/gibbs_k_scan.py cfg2 2>&1 | tail -1
done
for S in 4 16; do
  echo "S=$S"; CRBM_GIBBS_S=$S KS=0,1 python tools/gibbs_k_scan.py cfg2 2>&1 | tail -2
done
