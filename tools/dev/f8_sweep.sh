#!/bin/bash
# usage: f8_sweep.sh "<bench args>" "ENV=.." ...   prints ms/step per dtype and env
args="$1"; shift
for cfg in "$@"; do
  for dt in bf16 fp8; do
    env $cfg python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-nms --no-infer $args --dtype $dt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['dtype'], d['value'], d['ms_per_step'], d['roofline']['class_ms_per_step'], d['loss_items'])"
  done
done
