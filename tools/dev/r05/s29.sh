cd $GRAFT_REPO_ROOT
run() { python bench.py $1 --steps 30 --warmup 5 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_items'])"; }
for f in 100 50 25 100 50; do
  echo "== WGTR_FILL=$f"
  echo -n "c2 "; YS_WGTR_FILL=$f run ""
  echo -n "c3 "; YS_WGTR_FILL=$f run "--size s --batch 32"
done
