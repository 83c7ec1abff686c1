cd /tmp && export TMPDIR=/tmp
export YS_OVERLAP=0
for c in "" "--size x --imgsz 1280 --batch 16"; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o s -- python $GRAFT_REPO_ROOT/bench.py $c --steps 5 --warmup 2 --no-cpu-baseline --no-nms --no-infer > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/kernel_stats.py $(ls /tmp/p1/*/s_kernel_stats.csv /tmp/p1/s_kernel_stats.csv 2>/dev/null | head -1) 9 | grep -i "bn_fin_apply\|bn_bwd_apply_kernel\|total" ; rm -rf /tmp/p1
done
