# slab table budget / slab size A/B on one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4s2_slab_ab; mkdir -p $O
for kv in "CRBM_SLAB_TABLE_BUDGET=26624" "CRBM_SLAB_TABLE_BUDGET=49152" "CRBM_SLAB_TABLE_BUDGET=81920" "CRBM_SLAB_MOTIFS=40" "CRBM_SLAB_MOTIFS=80 CRBM_SLAB_TABLE_BUDGET=49152" "CRBM_SLAB_HGV=0"; do
  for shape in "300 10 0 256 200" "120 40 1 256 200" "300 10 0 4096 200" "256 4 1 4096 200" "120 40 1 2048 200"; do
    echo "[$kv] $(env $kv timeout -k 10 200 python tools/prof_big.py $shape 10 2>&1 | tr '\n' ' ')"
  done
done | tee $O/timing.txt
