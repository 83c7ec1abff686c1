# Round-3 GPU session 25: weight-gradient stream at the lowest priority (own hardware queue class) -- local and one-rank data-parallel step
cd $GRAFT_REPO_ROOT; export MASTER_ADDR=127.0.0.1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
for v in "X=1" "YS_ST2_PRIO=0" "X=1" "YS_ST2_PRIO=0"; do
  l=$(env $v timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  d=$(MASTER_PORT=$((29600 + RANDOM % 200)) env $v timeout 300 python bench.py --force-dist --steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$v local $l force-dist $d"
done
