cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_s10; mkdir -p $O
export YS_GEMM_HALO=1
timeout 1200 python -m pytest tests/test_conv.py tests/test_blocks.py tests/test_bnred.py -x -q -m gpu 2>&1 | tail -3
S="--steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-infer"
for halo in 1 0; do
  export YS_GEMM_HALO=$halo
  timeout 400 python bench.py $S --size x --imgsz 1280 --batch 16 --dump-launches $O/c5_h${halo}_launches.csv > $O/c5_h$halo.json 2> $O/c5_h$halo.err
  timeout 400 python bench.py $S --family 11 --size m --task segment --batch 32 --dump-launches $O/c4_h${halo}_launches.csv > $O/c4_h$halo.json 2> $O/c4_h$halo.err
done
python - <<'PY'
import json, csv, collections
for t in ('c5','c4'):
  for h in (1,0):
    try:
        j=json.loads(open('gpurun_out/r05_s10/%s_h%d.json'%(t,h)).read().strip().splitlines()[-1]); r=j['roofline']
        print(t, 'halo',h, j['dtype'], j['ms_per_step'], j['value'], r['kernel'], r['bound'], r['frac'])
    except Exception as e: print(t, h, 'ERR', e)
def agg(f):
    d=collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        a=d.setdefault(r['label'],[0,0.0]); a[0]+=1; a[1]+=float(r['us'])
    return d
for cfg in ('c5','c4'):
    h=agg('gpurun_out/r05_s10/%s_h1_launches.csv'%cfg); o=agg('gpurun_out/r05_s10/%s_h0_launches.csv'%cfg)
    for l,(n,us) in sorted(h.items(), key=lambda kv:-kv[1][1]):
        if not l.startswith('halo'): continue
        m=[(ol,v) for ol,v in o.items() if ol.startswith('gemm k33 s1') and ' '.join(ol.split()[4:8])==' '.join(l.split()[4:8])]
        old = m[0][1][1]/m[0][1][0] if m else float('nan')
        print('%-80s n=%2d avg %7.1f us  old %7.1f'%(l[:80],n,us/n, old))
PY
