#!/bin/bash
# rocprofv3 kernel stats of one configuration: prof_cfg.sh <name> "<bench args>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$1
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-nms --no-infer $2 > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-200
cd $R && python tools/kernel_stats.py $O/s_kernel_stats.csv 9 | head -28
