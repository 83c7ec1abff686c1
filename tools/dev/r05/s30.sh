cd $GRAFT_REPO_ROOT
python -m pytest tests/test_blocks.py tests/test_model.py tests/test_configs.py -m gpu -q -k "c2psa or four_row or v11 or 11 or c4" 2>&1 | tail -3
run() { python bench.py $1 --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_items'])"; }
for f in 1 0 1 0; do echo -n "ATTN_MFMA=$f c4 "; YS_ATTN_MFMA=$f run "--family 11 --size m --task segment --batch 32"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o s -- python $GRAFT_REPO_ROOT/bench.py --family 11 --size m --task segment --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-nms --no-infer > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/kernel_stats.py $(ls /tmp/p1/*/s_kernel_stats.csv /tmp/p1/s_kernel_stats.csv 2>/dev/null | head -1) 9 | grep -i "attn"
