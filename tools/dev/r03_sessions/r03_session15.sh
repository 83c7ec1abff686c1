# Round-3 GPU session 15: direct epilogue with 16-lane row swap (16-byte stores from registers) vs the LDS-staged epilogue (build epi0)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03o; mkdir -p $O
timeout 600 python -m pytest tests/test_conv.py tests/test_bnred.py tests/test_model.py tests/test_fp8.py -x -q -m gpu > $O/tests.txt 2>&1; tail -2 $O/tests.txt
line() { python -c "
import json,sys
j=json.loads(open('$1').read().strip().splitlines()[-1]); r=j['roofline']
print('%-20s %7.3f ms/step | %s' % ('$2', j['ms_per_step'], '  '.join('%s %.2f/%d' % (k.replace('conv_','').replace('_kernel',''), v['kernel_ms_per_step'], v['launches_per_step']) for k, v in list(r['kernels'].items())[:5])))
" 2>&1 | tail -1; }
B2="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
B5="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-infer --size x --imgsz 1280 --batch 16"
B4="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-nms --no-infer --family 11 --size m --task segment --batch 32"
for L in swap16:X staged:--lib=build/libyolosharp_hip_epi0.so swap16b:X stagedb:--lib=build/libyolosharp_hip_epi0.so; do
  tag=${L%%:*}; la=${L#*:}; [ "$la" = X ] && la=""
  timeout 300 $B2 $la > $O/c2_$tag.json 2> $O/c2_$tag.err; line $O/c2_$tag.json c2_$tag
done
for L in swap16:X staged:--lib=build/libyolosharp_hip_epi0.so; do
  tag=${L%%:*}; la=${L#*:}; [ "$la" = X ] && la=""
  timeout 300 $B5 $la > $O/c5_$tag.json 2> $O/c5_$tag.err; line $O/c5_$tag.json c5_$tag
  timeout 300 $B4 $la > $O/c4_$tag.json 2> $O/c4_$tag.err; line $O/c4_$tag.json c4_$tag
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-nms > $O/inf_swap16.json 2>/dev/null; python -c "
import json; j=json.loads(open('$O/inf_swap16.json').read().strip().splitlines()[-1]); print('infer swap16', j.get('infer'))"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-nms --lib=build/libyolosharp_hip_epi0.so > $O/inf_staged.json 2>/dev/null; python -c "
import json; j=json.loads(open('$O/inf_staged.json').read().strip().splitlines()[-1]); print('infer staged', j.get('infer'))"
