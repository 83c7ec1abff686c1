#!/bin/bash
# env_sweep.sh "<bench args>" "ENV=.. ENV2=.." ...   -> ms/step and class times per env set
args="$1"; shift
for cfg in "$@"; do
  env $cfg python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-nms --no-infer $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['dtype'], d['ms_per_step'], d['roofline']['class_ms_per_step'])"
done
