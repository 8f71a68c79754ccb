cd $GRAFT_REPO_ROOT
for rows in 640 704 768 832 896 960; do
  echo "stats rows=$rows"; CRBM_STATS_ROWS=$rows python tools/prof_train.py cfg2 300 | tail -1; CRBM_STATS_ROWS=$rows python tools/prof_train.py cfg2 300 | tail -1
done
