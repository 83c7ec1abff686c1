cd $GRAFT_REPO_ROOT
run() { python bench.py $1 --steps 30 --warmup 5 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for g in 4096 8192 0 4096 8192 0; do
  echo -n "BNB_GRID=$g: c2 "; if [ $g = 0 ]; then YS_BNB_ATOMIC=0 run ""; else YS_BNB_GRID=$g run ""; fi
done
