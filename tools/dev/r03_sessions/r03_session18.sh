# Round-3 GPU session 18: bisect the step-0 non-determinism of the config-5 graph
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O
for t in default:X=1 stages2:YS_GEMM_STAGES=2 nooverlap:YS_OVERLAP=0 nobnred:YS_BNRED=0 nodefer:YS_WGRED_DEFER=0; do
  tag=${t%%:*}; ev=$(echo ${t#*:} | tr ',' ' ')
  echo "== $tag"; env $ev timeout 300 python tools/dev/determinism_c5.py - 3 2>&1 | tail -6
done
echo "== epi0 lib (before the zero-C / mask changes)"; timeout 300 python tools/dev/determinism_c5.py build/libyolosharp_hip_epi0.so 3 2>&1 | tail -6
