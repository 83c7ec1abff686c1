cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/p1 -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-infer > /dev/null 2>&1
f=$(ls /tmp/p1/*/s_kernel_trace.csv /tmp/p1/s_kernel_trace.csv 2>/dev/null | head -1)
head -1 $f
python $GRAFT_REPO_ROOT/tools/dev/r05/gap_analysis.py $f 0.3
