#!/bin/bash
# launch dumps of the other BASELINE configurations (per-launch HIP-event records): C5 fp8 / bf16, C4, C3 per-GPU shape
cd $GRAFT_REPO_ROOT; O=gpurun_out/cfgs; mkdir -p $O
run() { tag=$1; shift; timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-nms --no-infer --dump-launches $O/launches_$tag.csv "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err; python -c "
import json; j=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag', j['ms_per_step'], {k:round(v['kernel_ms_per_step'],2) for k,v in j['roofline']['kernels'].items()})"; python tools/launch_report.py $O/launches_$tag.csv > $O/report_$tag.txt 2>/dev/null; }
run c5f8 --size x --imgsz 1280 --batch 16 --dtype fp8
run c5bf --size x --imgsz 1280 --batch 16
run c4 --family 11 --size m --task segment --batch 32
run c3 --size s --batch 32
