#!/bin/bash
# round-4 triage session: launch dump of the default step + s_memtime timelines of representative conv_p2 / conv_gemm layers
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04diag; mkdir -p $O
timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-nms --no-infer --dump-launches $O/launches.csv > $O/bench.json 2> $O/bench.err
python tools/launch_report.py $O/launches.csv > $O/launch_report.txt 2>&1; head -5 $O/launch_report.txt
timeout 300 python tools/dev/p2_timeline.py $GRAFT_REPO_ROOT/$O/p2_timeline.txt > /dev/null 2> $O/tl.err; wc -l $O/p2_timeline.txt
