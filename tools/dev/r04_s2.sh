#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/s2; mkdir -p $O
timeout 900 python -m pytest tests/test_production_routing.py -q -m gpu -x > $O/prod.txt 2>&1; tail -5 $O/prod.txt
timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-nms --no-infer --dump-launches $O/launches.csv > $O/bench.json 2> $O/bench.err
python tools/launch_report.py $O/launches.csv > $O/launch_report.txt 2>&1; head -3 $O/launch_report.txt
