cd $GRAFT_REPO_ROOT
echo "== new model per trial, first step"; DET_NEW=1 python tools/dev/r05/det_phase.py 40 8 2>&1 | tail -40
echo "== new model per trial, one AdamW step before"; DET_NEW=1 DET_PRE=1 python tools/dev/r05/det_phase.py 40 8 2>&1 | tail -40
