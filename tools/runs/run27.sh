cd $GRAFT_REPO_ROOT
export CRBM_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
B="-DCRBM_EXP_NO_VH"
for r in "" "-DCRBM_EXP_RANDOM_LETTERS"; do
for d in "-DCRBM_EXP_NO_PIPE" "-DCRBM_EXP_NO_PIPE -DCRBM_EXP_NO_PHILOX_H" "-DCRBM_EXP_NO_PIPE -DCRBM_EXP_ONE_GROUP" "-DCRBM_EXP_NO_PIPE -DCRBM_EXP_NO_PHILOX_H -DCRBM_EXP_ONE_GROUP" ""; do
  echo "== defines: $B $r $d"
  CRBM_JIT_DEFINES="$B $r $d" KS=16 timeout -k 10 200 python tools/gibbs_k_scan.py cfg2 || exit 1
done
done
echo "== full kernel, no pipe"; CRBM_JIT_DEFINES="-DCRBM_EXP_NO_PIPE" KS=1,16 timeout -k 10 200 python tools/gibbs_k_scan.py cfg2
