# Round-3 GPU session 29: fp8 blocked-GEMM fragments as units (q, 4+q) instead of (2q, 2q+1): bank conflicts, config-5 fp8 step, fp8 parity tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03z; mkdir -p $O
timeout 900 python -m pytest tests/test_fp8.py tests/test_configs.py -x -q -m gpu -k "fp8 or c5" 2>&1 | grep -E "passed|failed" | tail -1
for i in 1 2; do timeout 400 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-nms --no-infer --size x --imgsz 1280 --batch 16 --dtype fp8 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('c5 fp8', j['ms_per_step'], {k: round(v['kernel_ms_per_step'],2) for k,v in list(r['kernels'].items())[:3]})"; done
PROF_TAG=c5f8 PROF_ARGS="--size x --imgsz 1280 --batch 16 --dtype fp8" bash tools/r03_profile.sh > /dev/null 2>&1
python - <<'PY'
import json
j=json.load(open('gpurun_out/r03p_c5f8/sq_counters.json'))
v=j['kernels']['conv_gemm_kernel']; print('gemm (fp8 run)', {a: v[a] for a in ('mfma_util','lds_bank_conflict_ratio','wait_any_frac')})
PY
head -6 gpurun_out/r03p_c5f8/kernel_table.md
