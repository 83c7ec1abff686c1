cd $GRAFT_REPO_ROOT
DET_ONLY=production python tools/dev/r05/determinism.py 40 8 4 2>&1 | tail -2
DET_NEW=1 python tools/dev/r05/det_phase.py 60 8 2>&1 | tail -4
python -m pytest tests -m gpu -q 2>&1 | tail -12
