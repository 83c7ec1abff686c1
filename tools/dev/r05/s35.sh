cd $GRAFT_REPO_ROOT
run() { python bench.py $1 --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_items'])"; }
for f in 75 60 75 60; do
  echo "== HALO_MIN_FILL=$f"
  echo -n "c3 "; YS_HALO_MIN_FILL=$f run "--size s --batch 32"
  echo -n "c4 "; YS_HALO_MIN_FILL=$f run "--family 11 --size m --task segment --batch 32"
  echo -n "c2 "; YS_HALO_MIN_FILL=$f run ""
  echo -n "c5 "; YS_HALO_MIN_FILL=$f run "--size x --imgsz 1280 --batch 16"
done
