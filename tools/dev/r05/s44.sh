cd $GRAFT_REPO_ROOT
python -m pytest tests/test_conv.py tests/test_blocks.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5
S="64,80,80,80,80,3,1;64,64,80,80,64,3,1;16,80,320,320,80,3,1;64,64,80,80,80,3,1"
for f in 1 0; do echo "== HALO_KSPLIT=$f"; YS_HALO_KSPLIT=$f YS_LB_SHAPES="$S" python tools/dev/r05/layer_bench.py 2>&1 | tail -4; done
run() { python bench.py $1 --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_items'])"; }
for f in 1 0 1 0; do
  echo "== HALO_KSPLIT=$f"
  echo -n "c2 "; YS_HALO_KSPLIT=$f run ""
  echo -n "c5 "; YS_HALO_KSPLIT=$f run "--size x --imgsz 1280 --batch 16"
  echo -n "c5f8 "; YS_HALO_KSPLIT=$f run "--size x --imgsz 1280 --batch 16 --dtype fp8"
done
