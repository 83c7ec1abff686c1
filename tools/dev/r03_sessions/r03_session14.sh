# Round-3 GPU session 14: conv_gemm_kernel fragment read-ahead (both K-steps requested before the first MFMA), A/B against the old order
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03n; mkdir -p $O
timeout 600 python -m pytest tests/test_conv.py -x -q -m gpu > $O/tests.txt 2>&1; tail -1 $O/tests.txt
line() { python -c "
import json,sys
j=json.loads(open('$1').read().strip().splitlines()[-1]); r=j['roofline']
print('%-20s %7.3f ms/step | %s' % ('$2', j['ms_per_step'], '  '.join('%s %.2f/%d' % (k.replace('conv_','').replace('_kernel',''), v['kernel_ms_per_step'], v['launches_per_step']) for k, v in list(r['kernels'].items())[:5])))
" 2>&1 | tail -1; }
B5="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-infer --size x --imgsz 1280 --batch 16"
B4="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-nms --no-infer --family 11 --size m --task segment --batch 32"
B2="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
for L in ahead:X noahead:--lib=build/libyolosharp_hip_noahead.so; do
  tag=${L%%:*}; la=${L#*:}; [ "$la" = X ] && la=""
  timeout 300 $B5 $la > $O/c5_$tag.json 2> $O/c5_$tag.err; line $O/c5_$tag.json c5_$tag
  timeout 300 $B5 --dtype fp8 $la > $O/c5f8_$tag.json 2> $O/c5f8_$tag.err; line $O/c5f8_$tag.json c5f8_$tag
  timeout 300 $B4 $la > $O/c4_$tag.json 2> $O/c4_$tag.err; line $O/c4_$tag.json c4_$tag
  timeout 300 $B2 $la > $O/c2_$tag.json 2> $O/c2_$tag.err; line $O/c2_$tag.json c2_$tag
done
