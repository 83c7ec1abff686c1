# Round-3 GPU session 26: s_waitcnt lgkmcnt(0) inside ys_wave_sync (every wave-private LDS row exchange) -- full suite, step time, other configs
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03x; mkdir -p $O
( time timeout 1700 python -m pytest tests -x -q -m gpu ) > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt | tail -1
for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('c2', j['ms_per_step'])"; done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-nms 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('infer', j['infer']['images_per_s'])"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('nms', j['nms']['ms'])"
timeout 400 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-nms --no-infer --family 11 --size m --task segment --batch 32 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('c4', j['ms_per_step'])"
