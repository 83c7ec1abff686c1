#!/bin/bash
# A/B of the blocked-GEMM convolution kernel (YS_NO_GEMM=1 = previous kernels) on the wide-layer configurations + the headline one
mkdir -p gpurun_out/gemm
run() { # name, env, args
  env $2 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-nms --no-infer $3 --dump-launches gpurun_out/gemm/l_$1.csv 2>gpurun_out/gemm/$1.err | tail -1 > gpurun_out/gemm/$1.json
  python -c "import sys,json; d=json.loads(open('gpurun_out/gemm/$1.json').read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['class_ms_per_step'], d['loss_items'])"
}
run c5_gemm "A=1" "--size x --imgsz 1280 --batch 16"
run c5_old "YS_NO_GEMM=1" "--size x --imgsz 1280 --batch 16"
run c4_gemm "A=1" "--family 11 --size m --task segment --batch 32"
run c4_old "YS_NO_GEMM=1" "--family 11 --size m --task segment --batch 32"
run c2_gemm "A=1" ""
run c2_old "YS_NO_GEMM=1" ""
run c3_gemm "A=1" "--size s --batch 32"
run c3_old "YS_NO_GEMM=1" "--size s --batch 32"
