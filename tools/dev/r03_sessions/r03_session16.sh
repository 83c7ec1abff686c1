# Round-3 GPU session 16: VALU diet of conv_p2_kernel (zero-C first MFMA instead of cleared accumulators, channel masks only with bias) vs the previous build (build/libyolosharp_hip_epi0.so = staged epilogue before these changes)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03p; mkdir -p $O
timeout 600 python -m pytest tests/test_conv.py tests/test_bnred.py tests/test_model.py tests/test_fp8.py tests/test_heads.py -x -q -m gpu > $O/tests.txt 2>&1; tail -2 $O/tests.txt
line() { python -c "
import json,sys
j=json.loads(open('$1').read().strip().splitlines()[-1]); r=j['roofline']
print('%-20s %7.3f ms/step | %s' % ('$2', j['ms_per_step'], '  '.join('%s %.2f/%d' % (k.replace('conv_','').replace('_kernel',''), v['kernel_ms_per_step'], v['launches_per_step']) for k, v in list(r['kernels'].items())[:5])))
" 2>&1 | tail -1; }
B2="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
for L in new:X prev:--lib=build/libyolosharp_hip_epi0.so newb:X prevb:--lib=build/libyolosharp_hip_epi0.so; do
  tag=${L%%:*}; la=${L#*:}; [ "$la" = X ] && la=""
  timeout 300 $B2 $la > $O/c2_$tag.json 2> $O/c2_$tag.err; line $O/c2_$tag.json c2_$tag
done
