#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/s9; mkdir -p $O
YS_GROUP=0 YS_P2_VIA_GROUP=1 timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-nms --no-infer --dump-launches $O/launches_via.csv > $O/bench_via.json 2> $O/bench_via.err
grep "k33 s1 div1 cin80 cout80\|k33 s1 div1 cin64 cout64 M409600\|k33 s1 div1 cin32 cout32 M409600" $O/launches_via.csv | awk -F, '{print $NF, substr($2,1,95)}' | sort -k2 | head -40
python -c "
import json; j=json.loads(open('$O/bench_via.json').read().strip().splitlines()[-1]); print(j['ms_per_step'], {k:v['kernel_ms_per_step'] for k,v in j['roofline']['kernels'].items()})"
