cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_geom; mkdir -p $O
{
for rep in 1 2; do
for st in "0 0" "2 128" "2 256" "4 128" "4 256" "8 256" "8 512" "16 512" "16 1024" "32 1024"; do
  set -- $st
  echo "cfg2 S=$1 T=$2: $(CRBM_GIBBS_S=$1 CRBM_GIBBS_THREADS=$2 timeout -k 10 120 python tools/prof_gibbs.py cfg2 3000 2>&1 | tail -1)"
done
done
} 2>&1 | tee $O/geom.txt
