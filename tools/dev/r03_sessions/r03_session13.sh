# Round-3 GPU session 13: multi-stage operand pipeline of conv_gemm_kernel + vectorised batched wgrad reduce
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03m; mkdir -p $O
timeout 600 python -m pytest tests/test_conv.py tests/test_bnred.py tests/test_model.py tests/test_fp8.py -x -q -m gpu > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for st in 3 4; do YS_GEMM_STAGES=$st timeout 600 python -m pytest tests/test_conv.py tests/test_fp8.py -x -q -m gpu > $O/tests_st$st.txt 2>&1; tail -1 $O/tests_st$st.txt; done
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
line() { python -c "
import json,sys
j=json.loads(open('$1').read().strip().splitlines()[-1]); r=j['roofline']
print('%-20s %7.3f ms/step | %s' % ('$2', j['ms_per_step'], '  '.join('%s %.2f/%d' % (k.replace('conv_','').replace('_kernel',''), v['kernel_ms_per_step'], v['launches_per_step']) for k, v in list(r['kernels'].items())[:5])))
" 2>&1 | tail -1; }
for t in c2:X=1 c2_st2:YS_GEMM_STAGES=2 c2b:X=1; do
  tag=${t%%:*}; ev=$(echo ${t#*:} | tr ',' ' ')
  env $ev timeout 200 $B > $O/ab_$tag.json 2> $O/ab_$tag.err; line $O/ab_$tag.json $tag
done
B5="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-infer --size x --imgsz 1280 --batch 16"
for t in c5_st2:YS_GEMM_STAGES=2 c5_st3:YS_GEMM_STAGES=3 c5_st4:YS_GEMM_STAGES=4; do
  tag=${t%%:*}; ev=$(echo ${t#*:} | tr ',' ' ')
  env $ev timeout 300 $B5 > $O/ab_$tag.json 2> $O/ab_$tag.err; line $O/ab_$tag.json $tag
done
for t in c5f8_st2:YS_GEMM_STAGES=2 c5f8_st3:YS_GEMM_STAGES=3; do
  tag=${t%%:*}; ev=$(echo ${t#*:} | tr ',' ' ')
  env $ev timeout 300 $B5 --dtype fp8 > $O/ab_$tag.json 2> $O/ab_$tag.err; line $O/ab_$tag.json $tag
done
