cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_s09; mkdir -p $O
timeout 900 python -m pytest tests/test_kat.py tests/test_model.py tests/test_obb_pose.py tests/test_segment.py -q -m gpu -x --no-header -p no:cacheprovider 2>&1 | tail -3
S="--steps 40 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
for rep in 1 2 3; do
python bench.py $S --lib build/libyolosharp_hip_r05.so 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['ms_per_step'])"
python bench.py $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['ms_per_step'], d['loss_items'])"
done
cd /tmp && export TMPDIR=/tmp
YS_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-nms --no-infer > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/st/*/s_kernel_stats.csv $O/st/s_kernel_stats.csv 2>/dev/null | head -1)
python tools/kernel_stats.py $f 9 > $O/kernel_table.md; grep -i "loss\|tal\|total" $O/kernel_table.md; rm -rf $O/st
