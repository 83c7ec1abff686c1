cd $GRAFT_REPO_ROOT
A="--size x --imgsz 1280 --batch 16 --dtype fp8 --steps 12 --warmup 4 --no-cpu-baseline --no-nms --no-infer"
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$1', d['ms_per_step'], d['loss_items'], ' '.join('%s %.2f/%d'%(n.replace('conv_','').replace('_kernel',''),v['kernel_ms_per_step'],v['launches_per_step']) for n,v in list(k.items())[:6]))"; }
python bench.py $A 2>/dev/null | tail -1 | show base
YS_F8_MIN_TAPS=1 python bench.py $A 2>/dev/null | tail -1 | show taps1
YS_F8_MIN_CIN=64 python bench.py $A 2>/dev/null | tail -1 | show cin64
YS_F8_MIN_CIN=64 YS_F8_MIN_TAPS=1 python bench.py $A 2>/dev/null | tail -1 | show cin64taps1
YS_GEMM_HALO=0 python bench.py $A 2>/dev/null | tail -1 | show nohalo
python bench.py $A 2>/dev/null | tail -1 | show base2
