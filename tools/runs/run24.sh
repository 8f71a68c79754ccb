cd $GRAFT_REPO_ROOT
export CRBM_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
for d in "" "-DCRBM_EXP_NO_PHILOX_H" "-DCRBM_EXP_ONE_GROUP" "-DCRBM_EXP_NO_PHILOX_H -DCRBM_EXP_ONE_GROUP" "-DCRBM_EXP_NO_HV" "-DCRBM_EXP_NO_VH" "-DCRBM_EXP_NO_VH -DCRBM_EXP_NO_HV"; do
  echo "== defines: $d"
  CRBM_JIT_DEFINES="$d" KS=1,16 timeout -k 10 200 python tools/gibbs_k_scan.py cfg2 || exit 1
done
