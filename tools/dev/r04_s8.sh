#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/s8; mkdir -p $O
for g in 1 0; do
YS_GROUP=$g timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-nms --no-infer --dump-launches $O/launches_g$g.csv > $O/bench_g$g.json 2> $O/bench_g$g.err
grep "k33 s1 div1 cin80 cout80\|k33 s1 div1 cin64 cout64 M\(537600\|409600\|102400 acc0 nt256 mr2 nr2 wres1 npu6 tile16x8\|25600\)" $O/launches_g$g.csv | awk -F, '{print $NF, substr($2,1,95)}' | sort | uniq -c | sort -k3 | head -40
done
