cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer > /dev/null 2>&1
grep -i "nms" $(ls /tmp/p1/*/s_kernel_stats.csv /tmp/p1/s_kernel_stats.csv 2>/dev/null | head -1) | awk -F, '{print $1, $(NF-6), $(NF-4), $(NF-2), $(NF-1)}' | cut -c1-160
