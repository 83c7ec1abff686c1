#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/s10; mkdir -p $O
timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-nms --no-infer --dump-launches $O/launches.csv > $O/bench.json 2> $O/bench.err
grep "p2grp" $O/launches.csv | awk -F, '{print $NF, substr($2,1,200)}' | sort -k2 | uniq -c -f1 | head -20
