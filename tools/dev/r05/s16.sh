cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
A="--steps 3 --warmup 1 --batch 8 --no-cpu-baseline --no-nms --no-infer"
for extra in "" "" "--force-dist" "--force-dist --dist-backend c" "" ; do
  echo "== $extra"; python bench.py $A $extra 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['loss_items'], j.get('dist',{}).get('allreduce_exposed_ms'))"
done
for i in 1 2 3; do python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('B64', j['loss_items'], j['ms_per_step'])"; done
python -m pytest tests/test_dist.py tests/test_conv.py tests/test_blocks.py -m gpu -q 2>&1 | tail -3
