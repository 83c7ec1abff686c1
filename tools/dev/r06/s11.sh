cd $GRAFT_REPO_ROOT
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$1', d['ms_per_step'], ' '.join('%s %.2f/%d'%(n.replace('conv_','').replace('_kernel',''),v['kernel_ms_per_step'],v['launches_per_step']) for n,v in list(k.items())[:5]))"; }
for cfg in "c2:" "c3:--size s --batch 32" "c4:--family 11 --size m --task segment --batch 32"; do
  n=${cfg%%:*}; a=${cfg#*:}
  for g in 0 257 385 513; do
    YS_GEMM_MIN_CIN_K1=$g python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-infer $a 2>/dev/null | tail -1 | show ${n}_k1min$g
  done
done
