cd $GRAFT_REPO_ROOT
S="16,320,80,80,320,3,1;16,160,160,160,160,3,1;16,320,160,160,320,3,1;32,256,80,80,256,3,1;64,64,80,80,144,3,1"
for f in 1 2; do echo "== HALO_MR4=$f"; YS_HALO_MR4=$f YS_LB_SHAPES="$S" python tools/dev/r05/layer_bench.py 2>&1 | tail -5; done
