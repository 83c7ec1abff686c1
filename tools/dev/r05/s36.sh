cd $GRAFT_REPO_ROOT
python -m pytest tests/test_conv.py tests/test_blocks.py -m gpu -q -k "16x8 or random_shapes or tile_stream or halo" 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5
run() { python bench.py $1 --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_items'])"; }
for f in 1 0 1 0; do
  echo "== HALO_MR4=$f"
  echo -n "c3 "; YS_HALO_MR4=$f run "--size s --batch 32"
  echo -n "c4 "; YS_HALO_MR4=$f run "--family 11 --size m --task segment --batch 32"
  echo -n "c2 "; YS_HALO_MR4=$f run ""
  echo -n "c5 "; YS_HALO_MR4=$f run "--size x --imgsz 1280 --batch 16"
done
YS_LB_SHAPES="32,128,40,40,128,3,1;32,256,40,40,256,3,1;16,320,40,40,320,3,1;32,128,20,20,128,3,1" python tools/dev/r05/layer_bench.py 2>&1 | tail -5
YS_HALO_MR4=0 YS_LB_SHAPES="32,128,40,40,128,3,1;32,256,40,40,256,3,1;16,320,40,40,320,3,1;32,128,20,20,128,3,1" python tools/dev/r05/layer_bench.py 2>&1 | tail -5
