#!/bin/bash
# usage: knob_sweep.sh "ENV1=.. ENV2=.." ...   -> per config: step ms + class ms + a few key layers
mkdir -p gpurun_out/r2/sweep
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-infer --dump-launches gpurun_out/r2/sweep/l_$i.csv > gpurun_out/r2/sweep/b_$i.json 2>/dev/null
  python - "$cfg" gpurun_out/r2/sweep/b_$i.json gpurun_out/r2/sweep/l_$i.csv <<'PY'
import sys, json, csv, collections, re
cfg, bj, lc = sys.argv[1:4]
d = json.loads(open(bj).read().strip().splitlines()[-1])
agg = collections.OrderedDict()
for r in csv.DictReader(open(lc)):
    m = re.search(r'(k\d+) s(\d) .*?cin(\d+) cout(\d+) M(\d+) acc(\d)', r['label'])
    key = (r['class'][5:], m.group(1), m.group(3), m.group(4), m.group(5)) if m else (r['class'], r['label'][:30])
    a = agg.setdefault(key, [0, 0.0, r['label']]); a[0] += 1; a[1] += float(r['us'])
want = [("igemm","k33","64","64","102400"),("igemm","k33","128","128","25600"),("igemm","k33","32","32","409600"),("igemm","k33","80","80","409600"),("igemm","k11","384","256","25600"),("igemm","k11","64","64","409600"),("igemm","k33","16","16","1638400")]
print(f"== {cfg or 'default'}: {d['ms_per_step']} ms/step, classes {d['roofline']['class_ms_per_step']}")
for k in want:
    if k in agg:
        n, us, lab = agg[k]
        mm = re.search(r'mr\d nr\d wres\d npu\d+ tile\S+ grid\S+', lab)
        print(f"   {k[1]} {k[2]}->{k[3]} M{k[4]}: {us/n:6.1f} us x{n//2}  [{mm.group(0) if mm else ''}]")
PY
done
