#!/bin/bash
# One parameterised GPU session (replaces the per-session scripts of round 3, now under tools/dev/r03_sessions/).
#   gpurun --timeout 900 -- 'bash tools/gpu_session.sh <tag> "<pytest args or ->" "<ab list>" [profile]'
# <ab list>: space-separated name:ENV=VAL[,ENV=VAL] items, each one `bench.py --steps 30` run (same box, back to back, order as given).
# profile: after the A/B runs, a rocprofv3 kernel-trace --stats pass of the default configuration (kernel table into gpurun_out/<tag>/).
cd $GRAFT_REPO_ROOT; TAG=${1:-s}; O=gpurun_out/$TAG; mkdir -p $O
TESTS=${2:--}; AB=${3:-}; PROF=${4:-}
if [ "$TESTS" != "-" ]; then
  timeout 1500 python -m pytest $TESTS -q -m gpu ${PYTEST_X:-} > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|error" $O/tests.txt | tail -3; grep -E "^(FAILED|ERROR)|Error" $O/tests.txt | head -10
fi
for t in $AB; do
  name=${t%%:*}; ev=${t#*:}; ev=${ev//,/ }
  env $ev timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms ${BENCH_ARGS:-} > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]; k = r["kernels"]
    print("%-12s %.3f ms/step  %s  infer %s  loss %s" % (sys.argv[2], j["ms_per_step"], " ".join("%s %.2f/%d" % (n.replace("conv_", "").replace("_kernel", ""), v["kernel_ms_per_step"], v["launches_per_step"]) for n, v in k.items()), (j.get("infer") or {}).get("images_per_s"), j["loss_items"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
if [ -n "$PROF" ]; then
  cd /tmp && export TMPDIR=/tmp
  env ${PROF_ENV//,/ } YS_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-nms --no-infer ${BENCH_ARGS:-} > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(ls $O/st/*/s_kernel_stats.csv $O/st/s_kernel_stats.csv 2>/dev/null | head -1)
  python tools/kernel_stats.py $f 9 > $O/kernel_table.md; head -24 $O/kernel_table.md; cp $f $O/kernel_stats.csv; rm -rf $O/st
fi
