# Round-3 GPU session 12: batched store loop of the staged epilogue (conv_epi.h) -- parity, step time, gemm timeline
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03l; mkdir -p $O
timeout 600 python -m pytest tests/test_conv.py tests/test_bnred.py tests/test_model.py -x -q -m gpu > $O/tests.txt 2>&1; tail -2 $O/tests.txt
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
for t in new:X=1 new2:X=1; do
  tag=${t%%:*}; ev=$(echo ${t#*:} | tr ',' ' ')
  env $ev timeout 200 $B > $O/ab_$tag.json 2> $O/ab_$tag.err
  python -c "
import json
j=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]); r=j['roofline']
print('%-20s %7.3f ms/step | %s' % ('$tag', j['ms_per_step'], '  '.join('%s %.2f/%d' % (k.replace('conv_','').replace('_kernel',''), v['kernel_ms_per_step'], v['launches_per_step']) for k, v in list(r['kernels'].items())[:4])))
" 2>&1 | tail -1
done
python -m yolosharp_amd.build timeline > /dev/null 2>&1
timeout 200 python tools/dev/gemm_timeline.py $O/gemm_timeline.txt > /dev/null 2> $O/gtl.err
timeout 200 python tools/dev/p2_timeline.py $O/p2_timeline.txt > /dev/null 2> $O/ptl.err
