#!/bin/bash
# config-5 (fp8) A/B over env switches: bash tools/dev/r04_c5ab.sh "<name:ENV=V,ENV=V> ..."
cd $GRAFT_REPO_ROOT; O=gpurun_out/c5ab; mkdir -p $O
for t in $1; do
  IFS=: read name ev <<< "$t"; ev=${ev//,/ }
  env $ev timeout 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-nms --no-infer --size x --imgsz 1280 --batch 16 --dtype ${C5_DTYPE:-fp8} > $O/$name.json 2> $O/$name.err
  python -c "
import json; j=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('%-10s' % '$name', j['ms_per_step'], {k:round(v['kernel_ms_per_step'],2) for k,v in j['roofline']['kernels'].items()}, j['loss_items'])"
done
