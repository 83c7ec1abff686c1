cd $GRAFT_REPO_ROOT
S="--steps 40 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
for rep in 1 2; do
for v in 0 2 1; do
BENCH_P2_BWD_PERCU=$v python bench.py $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bwd_percu=$v', d['ms_per_step'])"
done; done
