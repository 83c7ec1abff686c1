export YS_LIB_PATH=$GRAFT_REPO_ROOT/yolosharp_amd/libyolosharp_hip_abl.so
for d in 0 64 128 31; do echo "DBG=$d"; YS_DBG=$d timeout 100 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nms --no-infer --dump-launches gpurun_out/abl_$d.csv 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['class_ms_per_step'])"; done
