#!/bin/bash
# DMA weight ring A/B: tests, config 2 (alternating), config 5 fp8
cd $GRAFT_REPO_ROOT; O=gpurun_out/ring; mkdir -p $O
timeout 1500 python -m pytest tests/test_conv.py tests/test_blocks.py tests/test_model.py tests/test_fp8.py -q -m gpu -x > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
bash tools/dev/r04_ab.sh ring - "new:-: g2off:g2off: new:-: g2off:g2off:"
for v in ring old ring old; do
  la=""; [ $v = old ] && la="--lib build/libyolosharp_hip_g2off.so"
  timeout 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-nms --no-infer --size x --imgsz 1280 --batch 16 --dtype fp8 $la > $O/c5_$v.json 2> $O/c5_$v.err
  python -c "
import json; j=json.loads(open('$O/c5_$v.json').read().strip().splitlines()[-1]); print('c5 %-6s' % '$v', j['ms_per_step'], {k:round(v['kernel_ms_per_step'],2) for k,v in j['roofline']['kernels'].items()}, j['loss_items'])"
done
