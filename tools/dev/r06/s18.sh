cd $GRAFT_REPO_ROOT
S="--steps 40 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
for rep in 1 2; do
for v in "A=1" "YS_WG_MAIN_MAXM=25600" "YS_WG_MAIN_MAXM=102400" "YS_WG_MAIN_MAXM=409600" "YS_WG_MAIN_MINM=1638400" "YS_WG_MAIN_MINM=409600"; do
env $v python bench.py $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
done; done
