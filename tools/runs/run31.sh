cd $GRAFT_REPO_ROOT
python tools/train_wall.py cfg2 50 200 800 200
