# Round-4 final evidence: smoke, the other configurations, the default bench line (with the PMC traffic of this source sha), NMS kernel table
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
: > $O/other_configs.jsonl
for cfg in "--size x --imgsz 1280 --batch 16" "--size x --imgsz 1280 --batch 16 --dtype fp8" "--family 11 --size m --task segment --batch 32" "--size s --batch 32"; do
  timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-infer $cfg 2>/dev/null | tail -1 >> $O/other_configs.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r04_final/other_configs.jsonl'):
    if not l.startswith('{'): continue
    j=json.loads(l); r=j['roofline']
    print('%-60s %s %8.3f ms/step %8.1f %s | %s %s frac %.3f' % (j['config']['workload'][:60], j['dtype'], j['ms_per_step'], j['value'], j['unit'], r['kernel'], r['bound'], r['frac']))
PY
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=j['roofline']
print('BENCH', j['value'], j['unit'], j['ms_per_step'], 'frac', r['frac'], 'traffic', r.get('traffic'), r.get('traffic_note'), 'step_frac', r.get('step_frac'), 'nms', j['nms']['ms'], 'infer', j['infer']['images_per_s'], 'cpu', j.get('cpu_baseline'))"
bash tools/dev/r04_nms.sh > $O/nms.txt 2>&1; tail -6 $O/nms.txt
