cd $GRAFT_REPO_ROOT
export CRBM_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
B="-DCRBM_EXP_NO_VH -DCRBM_EXP_RANDOM_LETTERS -DCRBM_EXP_NO_PIPE"
for d in "" "-DCRBM_EXP_NO_PHILOX_H" "-DCRBM_EXP_NO_DECIDE" "-DCRBM_EXP_NO_PHILOX_H -DCRBM_EXP_NO_DECIDE" "-DCRBM_EXP_NO_PHILOX_H -DCRBM_EXP_NO_DECIDE -DCRBM_EXP_ONE_GROUP"; do
  echo "== defines: $d"
  CRBM_JIT_DEFINES="$B $d" KS=16 timeout -k 10 200 python tools/gibbs_k_scan.py cfg2 || exit 1
done
