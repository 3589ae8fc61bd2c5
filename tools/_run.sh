cd $GRAFT_REPO_ROOT
python bench.py --live depth --steps 60 2>&1 | tail -1
