# Phase ablation of conv_p2_kernel (YS_DBG bit mask: 1 no patch loads, 2 no MFMA loop, 4 no epilogue, 8 no LDS patch store,
# 16 no weight loads, 32 interleaved tile order, 64 empty kernel, 128 prologue only).  Needs a triage build: add -DYS_P2_ABLATE to the hipcc flags
# in yolosharp_amd/build.py (the switches are compiled out of the product library).
for d in ${ABL:-0 16 2 18 4 22}; do echo "DBG=$d"; YS_DBG=$d timeout 100 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nms --no-infer --dump-launches gpurun_out/abl_$d.csv 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['class_ms_per_step'])"; done
