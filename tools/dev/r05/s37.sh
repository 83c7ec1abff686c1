cd $GRAFT_REPO_ROOT
DET_ONLY=production DET_SIZE=s python tools/dev/r05/determinism.py 10 32 3 2>&1 | tail -2
DET_ONLY=production DET_SIZE=x python tools/dev/r05/determinism.py 4 4 2 2>&1 | tail -2
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | head
