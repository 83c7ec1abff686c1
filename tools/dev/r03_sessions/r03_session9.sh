# Round-3 GPU session 9: does the blocked-GEMM kernel beat conv_p2_kernel on the 64- / 80-channel 3x3 layers (gate sweeps)?
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03i; mkdir -p $O
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
for t in base:X=1 gemm80:YS_GEMM_MIN_CIN=80 gemm64:YS_GEMM_MIN_CIN=64 gemm32:YS_GEMM_MIN_CIN=32 wgemm80:YS_WGEMM_MIN_C=80 wgemm64:YS_WGEMM_MIN_C=64 both64:YS_GEMM_MIN_CIN=64,YS_WGEMM_MIN_C=64 base2:X=1; do
  tag=${t%%:*}; ev=$(echo ${t#*:} | tr ',' ' ')
  env $ev timeout 200 $B > $O/ab_$tag.json 2> $O/ab_$tag.err
  python -c "
import json
j=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]); r=j['roofline']
print('%-10s %7.3f ms/step | %s' % ('$tag', j['ms_per_step'], '  '.join('%s %.2f/%d' % (k.replace('conv_','').replace('_kernel',''), v['kernel_ms_per_step'], v['launches_per_step']) for k, v in list(r['kernels'].items())[:5])))
" 2>&1 | tail -1
done
