cd $GRAFT_REPO_ROOT
export YS_GEMM_HALO=1
timeout 900 python -m pytest tests/test_conv.py -x -q -m gpu -k "reruns or halo or bf16" 2>&1 | tail -3
echo "== halo"; timeout 300 python tools/dev/r05/layer_bench.py 2>&1 | tail -3
echo "== old"; YS_GEMM_HALO=0 timeout 300 python tools/dev/r05/layer_bench.py 2>&1 | tail -3
bash tools/dev/r05/layer_pmc.sh 2>&1 | tail -6
