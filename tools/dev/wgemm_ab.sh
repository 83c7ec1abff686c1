#!/bin/bash
# A/B of the blocked-GEMM wgrad kernel: YS_NO_WGEMM=1 = previous kernels; K-tile and workgroups-per-CU variants
mkdir -p gpurun_out/wgemm
run() { # name, env, args
  env $2 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-nms --no-infer $3 --dump-launches gpurun_out/wgemm/l_$1.csv 2>gpurun_out/wgemm/$1.err | tail -1 > gpurun_out/wgemm/$1.json
  python -c "import sys,json; d=json.loads(open('gpurun_out/wgemm/$1.json').read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['class_ms_per_step'], d['loss_items'])"
}
C5="--size x --imgsz 1280 --batch 16"
run c5_new "A=1" "$C5"
run c5_old "YS_NO_WGEMM=1" "$C5"
run c5_kt32 "YS_WGEMM_KT=32" "$C5"
run c5_wpc1 "YS_WGEMM_WPC=1" "$C5"
run c5_kt32_wpc3 "YS_WGEMM_KT=32 YS_WGEMM_WPC=3" "$C5"
run c4_new "A=1" "--family 11 --size m --task segment --batch 32"
run c4_old "YS_NO_WGEMM=1" "--family 11 --size m --task segment --batch 32"
run c2_new "A=1" ""
run c3_new "A=1" "--size s --batch 32"
run c3_old "YS_NO_WGEMM=1" "--size s --batch 32"
