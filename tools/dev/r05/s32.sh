cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model.py tests/test_production_routing.py -m gpu -q 2>&1 | grep -E "passed|failed"
run() { python bench.py $1 --steps 30 --warmup 5 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_items'])"; }
echo -n "c2 "; run ""
echo -n "c3 "; run "--size s --batch 32"
echo -n "c5 "; run "--size x --imgsz 1280 --batch 16"
cd /tmp && export TMPDIR=/tmp
for c in "" "--size x --imgsz 1280 --batch 16"; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o s -- python $GRAFT_REPO_ROOT/bench.py $c --steps 5 --warmup 2 --no-cpu-baseline --no-nms --no-infer > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/kernel_stats.py $(ls /tmp/p1/*/s_kernel_stats.csv /tmp/p1/s_kernel_stats.csv 2>/dev/null | head -1) 9 | grep -i "bn_fin_apply\|bn_bwd_apply_kernel\|total" ; rm -rf /tmp/p1
done
