cd $GRAFT_REPO_ROOT
python -m pytest tests/test_bnred.py tests/test_model.py tests/test_production_routing.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | head -5
run() { python bench.py $1 --steps 30 --warmup 5 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_items'])"; }
for f in 1 0 1 0; do
  echo "== BNB_ATOMIC=$f"
  echo -n "c2 "; YS_BNB_ATOMIC=$f run ""
  echo -n "c3 "; YS_BNB_ATOMIC=$f run "--size s --batch 32"
  echo -n "c4 "; YS_BNB_ATOMIC=$f run "--family 11 --size m --task segment --batch 32"
done
DET_ONLY=production python tools/dev/r05/determinism.py 12 8 4 2>&1 | tail -1
