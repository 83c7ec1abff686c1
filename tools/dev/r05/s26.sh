cd $GRAFT_REPO_ROOT
python tools/dev/r05/gemm_tl_mid.py 2>&1 | tail -80
