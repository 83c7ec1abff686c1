#!/bin/bash
# round-4 A/B session: gpurun --timeout N -- 'bash tools/dev/r04_ab.sh <tag> "<pytest args or ->" "<name[:lib[:ENV=V,ENV=V]]> ..." [timeline] [profile]'
# every item is one `bench.py --steps 30` run (same box, back to back, in the order given; repeat a name to alternate); lib = a
# triage build under build/ (libyolosharp_hip_<lib>.so) or `-` for the product library
cd $GRAFT_REPO_ROOT; TAG=${1:-s}; O=gpurun_out/$TAG; mkdir -p $O
TESTS=${2:--}; AB=${3:-}; TL=${4:-}; PROF=${5:-}
if [ "$TESTS" != "-" ]; then
  timeout 1500 python -m pytest $TESTS -q -m gpu -x > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|error" $O/tests.txt | tail -3; grep -E "^(FAILED|ERROR)|Error" $O/tests.txt | head -10
fi
i=0
for t in $AB; do
  i=$((i+1)); IFS=: read name lib ev <<< "$t"; ev=${ev//,/ }
  la=""; if [ -n "$lib" ] && [ "$lib" != "-" ]; then la="--lib build/libyolosharp_hip_$lib.so"; fi
  env $ev timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms $la ${BENCH_ARGS:-} > $O/${i}_$name.json 2> $O/${i}_$name.err
  python - "$O/${i}_$name.json" "$name" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]; k = r["kernels"]
    print("%-12s %.3f ms/step  %s  infer %s  loss %s" % (sys.argv[2], j["ms_per_step"], " ".join("%s %.2f/%d" % (n.replace("conv_", "").replace("_kernel", ""), v["kernel_ms_per_step"], v["launches_per_step"]) for n, v in k.items()), (j.get("infer") or {}).get("images_per_s"), j["loss_items"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
if [ -n "$TL" ] && [ "$TL" != "-" ]; then
  timeout 300 python tools/dev/p2_timeline.py $GRAFT_REPO_ROOT/$O/p2_timeline.txt > /dev/null 2> $O/tl.err; wc -l $O/p2_timeline.txt
fi
if [ -n "$PROF" ] && [ "$PROF" != "-" ]; then
  cd /tmp && export TMPDIR=/tmp
  env ${PROF_ENV//,/ } YS_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-nms --no-infer ${BENCH_ARGS:-} > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(ls $O/st/*/s_kernel_stats.csv $O/st/s_kernel_stats.csv 2>/dev/null | head -1)
  python tools/kernel_stats.py $f 9 > $O/kernel_table.md; head -24 $O/kernel_table.md; cp $f $O/kernel_stats.csv; rm -rf $O/st
fi
