cd $GRAFT_REPO_ROOT
export CRBM_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
for rep in 1 2; do
for d in "" "-DCRBM_AB_OLD_COPY" "-DCRBM_AB_COPY_FIRST" "-DCRBM_AB_COPY_FIRST -DCRBM_AB_OLD_COPY" "-DCRBM_AB_LDS_STORE" "-DCRBM_AB_LDS_STORE -DCRBM_AB_COPY_FIRST -DCRBM_AB_OLD_COPY"; do
  echo "== defines: $d"
  CRBM_JIT_DEFINES="$d" KS=0,1 timeout -k 10 200 python tools/gibbs_k_scan.py cfg2 || exit 1
done
done
