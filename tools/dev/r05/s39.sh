cd $GRAFT_REPO_ROOT
python -m pytest tests/test_blocks.py tests/test_model.py tests/test_heads.py tests/test_configs.py -m gpu -q -k "c2psa or v11 or 11 or dw or c4 or detect" 2>&1 | grep -E "passed|failed|FAILED" | head -4
run() { python bench.py $1 --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_items'])"; }
echo -n "c4 "; run "--family 11 --size m --task segment --batch 32"
echo -n "c4 "; run "--family 11 --size m --task segment --batch 32"
cd /tmp && export TMPDIR=/tmp; export YS_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o s -- python $GRAFT_REPO_ROOT/bench.py --family 11 --size m --task segment --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-nms --no-infer > /dev/null 2>&1
grep -i "dwconv" $(ls /tmp/p1/*/s_kernel_stats.csv /tmp/p1/s_kernel_stats.csv 2>/dev/null | head -1) | awk -F, '{print $1, $(NF-6), $(NF-4)}' | cut -c1-120
