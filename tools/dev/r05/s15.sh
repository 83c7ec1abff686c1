cd $GRAFT_REPO_ROOT
S="--steps 30 --warmup 5 --no-cpu-baseline --no-nms --no-infer"
for a in 1 0 1 0; do echo "BN_ATOMIC=$a"; YS_BN_ATOMIC=$a timeout 300 python bench.py $S 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['loss_items'])"; done
python -m pytest tests/test_dist.py -m gpu -q -k bench_force 2>&1 | grep -E "^E |passed|failed" | head -8
timeout 1200 python -m pytest tests/test_model.py tests/test_blocks.py tests/test_conv.py -m gpu -q 2>&1 | tail -3
