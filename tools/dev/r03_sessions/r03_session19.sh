cd $GRAFT_REPO_ROOT
export REPS=12
timeout 600 python tools/dev/determinism_layer.py 2>&1 | tail -8
timeout 300 python tools/dev/determinism_c5.py - 3 2>&1 | tail -6
