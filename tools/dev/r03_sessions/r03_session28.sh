# Round-3 GPU session 28: last-arriver BatchNorm finalize inside conv_p2_kernel (YS_BN_TICKET=1) -- parity, launch rows, step time
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03y; mkdir -p $O
YS_BN_TICKET=2 timeout 900 python -m pytest tests/test_model.py tests/test_blocks.py -x -q -m gpu > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt | tail -1
for t in ticket:YS_BN_TICKET=1 uc:YS_BN_TICKET=2 plain:X=1 uc2:YS_BN_TICKET=2 plain2:X=1; do
  tag=${t%%:*}; ev=${t#*:}
  env $ev timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer > $O/$tag.json 2>/dev/null
  python -c "
import json; j=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=j['roofline']
print('%-8s %.3f ms/step  p2 %.2f ms  loss %s' % ('$tag', j['ms_per_step'], r['kernels']['conv_p2_kernel']['kernel_ms_per_step'], j['loss_items']))"
done
cd /tmp && export TMPDIR=/tmp
YS_BN_TICKET=2 YS_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-nms --no-infer > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/kernel_stats.py $(ls $O/st/*/s_kernel_stats.csv $O/st/s_kernel_stats.csv 2>/dev/null | head -1) 9 | head -12
rm -rf $O/st
