cd $GRAFT_REPO_ROOT/_bisect/r4
DET_ONLY=production python tools/dev/r05/determinism.py 24 8 4 2>&1 | tail -3
DET_ONLY="overlap off" python tools/dev/r05/determinism.py 24 8 4 2>&1 | tail -3
