cd $GRAFT_REPO_ROOT; O=gpurun_out/r03e; mkdir -p $O
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
for t in base:X=1 overlap:YS_OVERLAP=1 base2:X=1 overlap2:YS_OVERLAP=1; do
  tag=${t%%:*}; ev=${t#*:}
  env $ev timeout 200 $B > $O/ab_$tag.json 2> $O/ab_$tag.err
  python -c "
import json,sys
j=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]); r=j['roofline']
print('%-10s %7.3f ms/step %8.1f img/s | igemm %.3f wgrad %.3f' % ('$tag', j['ms_per_step'], j['value'], r['class_ms_per_step']['conv_igemm'], r['class_ms_per_step']['conv_wgrad']))
"
done
