cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_s15; mkdir -p $O
timeout 1200 python -m pytest tests/test_conv.py tests/test_configs.py tests/test_detector.py tests/test_model.py -q -m gpu -x --no-header -p no:cacheprovider 2>&1 | tail -3
python tools/evalprof.py > /dev/null 2>&1; python tools/launch_report.py gpurun_out/eval_launches.csv 2>/dev/null | grep -i "halo\|total"
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-nms 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['infer'])"; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-nms --lib build/libyolosharp_hip_r05.so 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['ms_per_step'], d['infer'])"
