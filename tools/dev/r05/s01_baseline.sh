# Round-5 session 1: baseline of the round-4 tree on today's box: C2 bench line, per-launch dumps of C2/C3/C4/C5 (bf16, fp8)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_s01; mkdir -p $O
S="--steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-infer"
timeout 300 python bench.py $S --dump-launches $O/c2_launches.csv > $O/c2.json 2> $O/c2.err; tail -c 600 $O/c2.json
timeout 400 python bench.py $S --size x --imgsz 1280 --batch 16 --dump-launches $O/c5_launches.csv > $O/c5.json 2> $O/c5.err
timeout 400 python bench.py $S --size x --imgsz 1280 --batch 16 --dtype fp8 --dump-launches $O/c5f8_launches.csv > $O/c5f8.json 2> $O/c5f8.err
timeout 400 python bench.py $S --family 11 --size m --task segment --batch 32 --dump-launches $O/c4_launches.csv > $O/c4.json 2> $O/c4.err
timeout 400 python bench.py $S --size s --batch 32 --dump-launches $O/c3_launches.csv > $O/c3.json 2> $O/c3.err
python - <<'PY'
import json
for t in ('c2','c5','c5f8','c4','c3'):
    try:
        j=json.loads(open('gpurun_out/r05_s01/%s.json'%t).read().strip().splitlines()[-1]); r=j['roofline']
        print(t, j['dtype'], j['ms_per_step'], j['value'], r['kernel'], r['bound'], r['frac'])
    except Exception as e: print(t, 'ERR', e)
PY
