cd $GRAFT_REPO_ROOT
S="--steps 40 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
for rep in 1 2 3; do
python bench.py $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base ', d['ms_per_step'])"
BENCH_WG_FORCE=0,0,1 python bench.py $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgpc1', d['ms_per_step'], d['loss_items'])"
done
for cfg in "--size s --batch 32" "--family 11 --size m --task segment --batch 32"; do
python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-nms --no-infer $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base ', d['ms_per_step'])"
BENCH_WG_FORCE=0,0,1 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-nms --no-infer $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgpc1', d['ms_per_step'])"
done
