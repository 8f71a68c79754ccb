cd $GRAFT_REPO_ROOT
KS=0,1,16 timeout -k 10 120 python tools/gibbs_k_scan.py cfg2
