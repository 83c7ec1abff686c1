# Round-3 GPU session 11: NMS after the launch merges; occupancy / tile-cost knobs of conv_p2_kernel with the round-3 prologue
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03k; mkdir -p $O
timeout 300 python -m pytest tests/test_nms.py tests/test_obb.py tests/test_obb_pose.py -x -q -m gpu > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer > $O/nms.json 2> $O/nms.err
python -c "
import json
j=json.loads(open('$O/nms.json').read().strip().splitlines()[-1]); print('nms', j['nms'])"
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
for t in base:X=1 lds3_54k:YS_P2_LDS3=54600 lds3_54k_oldpitch:YS_P2_LDS3=54600,YS_P2_PITCH=0,YS_P2_ROWPAD=0 lds3_54k_norowpad:YS_P2_LDS3=54600,YS_P2_ROWPAD=0 tc0:YS_P2_TILECONST=0 tc1500:YS_P2_TILECONST=1500 tc6000:YS_P2_TILECONST=6000 tc12000:YS_P2_TILECONST=12000 wres30:YS_P2_WRESMAX=30000 wres60:YS_P2_WRESMAX=60000 base2:X=1; do
  tag=${t%%:*}; ev=$(echo ${t#*:} | tr ',' ' ')
  env $ev timeout 200 $B > $O/ab_$tag.json 2> $O/ab_$tag.err
  python -c "
import json
j=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]); r=j['roofline']
print('%-20s %7.3f ms/step | %s' % ('$tag', j['ms_per_step'], '  '.join('%s %.2f/%d' % (k.replace('conv_','').replace('_kernel',''), v['kernel_ms_per_step'], v['launches_per_step']) for k, v in list(r['kernels'].items())[:4])))
" 2>&1 | tail -1
done
