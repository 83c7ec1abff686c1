# Round-3 GPU session 4: table-driven epilogue rows + LDS-DMA prologue (product) -- correctness subset, bench, timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d
mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_bnred.py tests/test_conv.py tests/test_model.py tests/test_blocks.py tests/test_segment.py tests/test_fp8.py -x -q -m gpu > $O/tests.txt 2>&1
tail -5 $O/tests.txt
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms"
: > $O/ab.txt
for tag in prod prod2 ${EXTRA_TAGS}; do
  timeout 200 $B --dump-launches $O/launches_$tag.csv > $O/ab_$tag.json 2> $O/ab_$tag.err
  python - $tag $O/ab_$tag.json >> $O/ab.txt <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = j["roofline"]
    k = r["kernels"]
    print("%-10s %7.3f ms/step %8.1f img/s | p2 %.3f gemm %.3f wgrad_tr %.3f | igemm %.3f wgrad %.3f | infer %.0f img/s" % (sys.argv[1], j["ms_per_step"], j["value"], k.get("conv_p2_kernel", {}).get("kernel_ms_per_step", 0), k.get("conv_gemm_kernel", {}).get("kernel_ms_per_step", 0), k.get("conv_wgrad_tr_kernel", {}).get("kernel_ms_per_step", 0), r["class_ms_per_step"]["conv_igemm"], r["class_ms_per_step"]["conv_wgrad"], (j.get("infer") or {}).get("images_per_s", 0)))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cat $O/ab.txt
timeout 200 python tools/dev/p2_timeline.py $O/p2_timeline.txt > /dev/null 2> $O/tl.err
echo done
