#!/bin/bash
# ms/step of the BASELINE configurations on the current build: usage cfg_sweep.sh <outdir> ["ENV=.. ENV2=.."]
out=gpurun_out/$1; mkdir -p $out; envs="${2:-A=1}"
run() { # name, args
  env $envs python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-nms --no-infer $2 --dump-launches $out/l_$1.csv 2>$out/$1.err | tail -1 > $out/$1.json
  python -c "import sys,json; d=json.loads(open('$out/$1.json').read()); print('$1', d['dtype'], d['value'], d['ms_per_step'], d['roofline']['class_ms_per_step'], d['loss_items'])"
}
run c5_bf16 "--size x --imgsz 1280 --batch 16"
run c5_fp8 "--size x --imgsz 1280 --batch 16 --dtype fp8"
run c4 "--family 11 --size m --task segment --batch 32"
run c3 "--size s --batch 32"
run c3_fp8 "--size s --batch 32 --dtype fp8"
run c2 ""
