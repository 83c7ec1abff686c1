cd $GRAFT_REPO_ROOT; O=gpurun_out/r03h; mkdir -p $O
timeout 600 python -m pytest tests/test_fp8.py tests/test_configs.py -x -q -m gpu -k "fp8 or c5" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
C5="--size x --imgsz 1280 --batch 16 --steps 10 --warmup 4 --no-cpu-baseline --no-nms --no-infer --dtype fp8"
for t in fused:X=1 nored:YS_BNRED=0 nooverlap:YS_OVERLAP=0; do
  tag=${t%%:*}; ev=${t#*:}
  env $ev timeout 300 python bench.py $C5 > $O/c5_$tag.json 2> $O/c5_$tag.err
  python -c "
import json
j=json.loads(open('$O/c5_$tag.json').read().strip().splitlines()[-1]); r=j['roofline']
print('%-10s %8.3f ms/step | %s' % ('$tag', j['ms_per_step'], '  '.join('%s %.2f' % (k, v['kernel_ms_per_step']) for k, v in list(r['kernels'].items())[:5])))
"
done
