cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_s09; mkdir -p $O
export YS_GEMM_HALO=1
timeout 900 python -m pytest tests/test_conv.py -x -q -m gpu -k "reruns or halo or bf16" 2>&1 | tail -3
echo "== halo"; timeout 300 python tools/dev/r05/layer_bench.py 2>&1 | tail -3
S="--steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-infer"
for halo in 1 0; do
  export YS_GEMM_HALO=$halo
  timeout 400 python bench.py $S --size x --imgsz 1280 --batch 16 --dump-launches $O/c5_h${halo}_launches.csv > $O/c5_h$halo.json 2> $O/c5_h$halo.err
  timeout 400 python bench.py $S --family 11 --size m --task segment --batch 32 --dump-launches $O/c4_h${halo}_launches.csv > $O/c4_h$halo.json 2> $O/c4_h$halo.err
  timeout 400 python bench.py $S --size s --batch 32 --dump-launches $O/c3_h${halo}_launches.csv > $O/c3_h$halo.json 2> $O/c3_h$halo.err
  timeout 300 python bench.py $S --dump-launches $O/c2_h${halo}_launches.csv > $O/c2_h$halo.json 2> $O/c2_h$halo.err
done
python - <<'PY'
import json
for t in ('c5','c4','c3','c2'):
  for h in (1,0):
    try:
        j=json.loads(open('gpurun_out/r05_s09/%s_h%d.json'%(t,h)).read().strip().splitlines()[-1]); r=j['roofline']
        print(t, 'halo',h, j['dtype'], j['ms_per_step'], j['value'], r['kernel'], r['bound'], r['frac'])
    except Exception as e: print(t, h, 'ERR', e)
PY
