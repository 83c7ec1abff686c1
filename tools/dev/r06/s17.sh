cd $GRAFT_REPO_ROOT
S="--steps 40 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
for rep in 1 2; do
for v in 1 356 292 228 164 132; do
YS_OVERLAP=$v python bench.py $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('overlap=$v', d['ms_per_step'], d['loss_items'])"
done; done
