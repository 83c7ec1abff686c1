cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv.py -x -q -m gpu -k "reruns or halo or bf16" 2>&1 | tail -3
echo "== product"; timeout 300 python tools/dev/r05/layer_bench.py 2>&1 | tail -3
L=build/libyolosharp_hip_abl.so
for dbg in 0 8 24; do
  echo "== YS_GEMM_DBG=$dbg"; YS_GEMM_DBG=$dbg timeout 300 python tools/dev/r05/layer_bench.py $L 2>&1 | tail -3
done
python tools/dev/r05/halo_timeline.py 2>&1 | tail -16
