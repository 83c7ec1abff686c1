cd $GRAFT_REPO_ROOT
run() { python bench.py --lib $1 --steps 40 --warmup 5 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_items'])"; }
for i in 1 2 3; do
  echo -n "prev "; run $GRAFT_REPO_ROOT/_bisect/prev/yolosharp_amd/libyolosharp_hip.so
  echo -n "new  "; run $GRAFT_REPO_ROOT/yolosharp_amd/libyolosharp_hip.so
done
