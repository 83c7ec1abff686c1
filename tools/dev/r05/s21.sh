cd $GRAFT_REPO_ROOT
python tools/dev/r05/determinism.py 16 8 4 2>&1 | tail -8
