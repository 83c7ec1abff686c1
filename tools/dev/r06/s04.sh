# r06 session 4: stem prefetch A/B + the other configurations on the current tree
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_s04; mkdir -p $O
timeout 600 python -m pytest tests/test_model.py -q -m gpu -k "stem or first_step" 2>&1 | tail -3
S="--steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = j["roofline"]["kernels"]
    print("%-10s %.3f ms/step  %s  host %.2f" % (sys.argv[2], j["ms_per_step"], " ".join("%s %.2f/%d" % (n.replace("conv_", "").replace("_kernel", ""), v["kernel_ms_per_step"], v["launches_per_step"]) for n, v in k.items()), j.get("host_enqueue_ms_per_step", -1)))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
python bench.py $S --lib build/libyolosharp_hip_r05.so > $O/c2_prev_$rep.json 2>/dev/null; show $O/c2_prev_$rep.json c2_prev
python bench.py $S > $O/c2_new_$rep.json 2>/dev/null; show $O/c2_new_$rep.json c2_new
done
for cfg in "c3:--size s --batch 32" "c4:--family 11 --size m --task segment --batch 32" "c5:--size x --imgsz 1280 --batch 16" "c5f8:--size x --imgsz 1280 --batch 16 --dtype fp8"; do
  n=${cfg%%:*}; a=${cfg#*:}
  python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-nms --no-infer $a --lib build/libyolosharp_hip_r05.so > $O/${n}_prev.json 2>/dev/null; show $O/${n}_prev.json ${n}_prev
  python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-nms --no-infer $a > $O/${n}_new.json 2>/dev/null; show $O/${n}_new.json ${n}_new
done
