cd $GRAFT_REPO_ROOT
S="--steps 40 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
for rep in 1 2 3; do
python bench.py $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('low ', d['ms_per_step'])"
YS_OVERLAP=2 python bench.py $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('high', d['ms_per_step'], d['loss_items'])"
done
