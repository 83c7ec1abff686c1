# Round-3 GPU session 17: head lanes (P4 / P5 towers on side streams in the training forward) -- full GPU suite, then A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03q; mkdir -p $O
line() { python -c "
import json,sys
j=json.loads(open('$1').read().strip().splitlines()[-1]); r=j['roofline']
print('%-20s %7.3f ms/step | %s' % ('$2', j['ms_per_step'], '  '.join('%s %.2f/%d' % (k.replace('conv_','').replace('_kernel',''), v['kernel_ms_per_step'], v['launches_per_step']) for k, v in list(r['kernels'].items())[:5])))
" 2>&1 | tail -1; }
B2="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
for t in lanes:X=1 nolanes:YS_HEAD_LANES=0 lanesb:X=1 nolanesb:YS_HEAD_LANES=0; do
  tag=${t%%:*}; ev=$(echo ${t#*:} | tr ',' ' ')
  env $ev timeout 300 $B2 > $O/c2_$tag.json 2> $O/c2_$tag.err; line $O/c2_$tag.json c2_$tag
done
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
