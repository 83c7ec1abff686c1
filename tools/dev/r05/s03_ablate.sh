# Round-5 session 3: halo kernel: rerun-determinism test, then ablations of the K-step on three layers (triage build)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_s03; mkdir -p $O
timeout 900 python -m pytest tests/test_conv.py -x -q -m gpu -k "reruns or halo" 2>&1 | tail -3
L=build/libyolosharp_hip_abl.so
for dbg in 0 8 12 4 16 24 88 1 2 3 32; do
  echo "== YS_GEMM_DBG=$dbg"; YS_GEMM_DBG=$dbg timeout 300 python tools/dev/r05/layer_bench.py $L 2>&1 | tail -3
done
echo "== old kernel"; YS_GEMM_HALO=0 timeout 300 python tools/dev/r05/layer_bench.py $L 2>&1 | tail -3
