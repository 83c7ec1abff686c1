# Round-3 GPU session 7: the other BASELINE configurations with the round-3 code (C5 bf16 / fp8, C4, C3 per-GPU shape)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g
mkdir -p $O; cd $R
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = j["roofline"]
    ks = "  ".join("%s %.2f" % (k.replace("conv_", "").replace("_kernel", ""), v["kernel_ms_per_step"]) for k, v in list(r["kernels"].items())[:5])
    print("%-14s %8.3f ms/step %8.1f img/s | dom %s %s frac %.3f | %s | step %.0f TF/s" % (sys.argv[1], j["ms_per_step"], j["value"], r["kernel"], r["bound"], r["frac"], ks, r["step_TFLOPs"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
C5="--size x --imgsz 1280 --batch 16 --steps 10 --warmup 4 --no-cpu-baseline --no-nms --no-infer"
timeout 300 python bench.py $C5 --dtype bf16 > $O/c5_bf16.json 2> $O/c5_bf16.err; line c5_bf16 $O/c5_bf16.json
YS_BNRED=0 timeout 300 python bench.py $C5 --dtype bf16 > $O/c5_bf16_nored.json 2> $O/c5_bf16_nored.err; line c5_bf16_nored $O/c5_bf16_nored.json
timeout 300 python bench.py $C5 --dtype fp8 --dump-launches $O/c5_fp8_launches.csv > $O/c5_fp8.json 2> $O/c5_fp8.err; line c5_fp8 $O/c5_fp8.json
timeout 300 python bench.py --family 11 --size m --task segment --batch 32 --steps 10 --warmup 4 --no-cpu-baseline --no-nms --no-infer > $O/c4.json 2> $O/c4.err; line c4_v11m_seg $O/c4.json
timeout 300 python bench.py --size s --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-infer > $O/c3.json 2> $O/c3.err; line c3_v8s_b32 $O/c3.json
echo done
