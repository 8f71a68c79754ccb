cd $GRAFT_REPO_ROOT
for g in 1024 768 512; do
  echo "gibbs grid=$g"; CRBM_GIBBS_GRID=$g python tools/prof_train.py cfg2 300 | tail -1; CRBM_GIBBS_GRID=$g python tools/prof_train.py cfg2 300 | tail -1
done
