cd $GRAFT_REPO_ROOT
L=build/libyolosharp_hip_abl.so
for dbg in 24 25 26 27 56; do
  echo "== YS_GEMM_DBG=$dbg"; YS_GEMM_DBG=$dbg timeout 300 python tools/dev/r05/layer_bench.py $L 2>&1 | tail -3
done
