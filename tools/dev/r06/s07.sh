cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_s07; mkdir -p $O
S="--no-cpu-baseline --no-nms --no-infer"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = j["roofline"]["kernels"]
    print("%-14s %.3f ms/step  %s" % (sys.argv[2], j["ms_per_step"], " ".join("%s %.2f/%d" % (n.replace("conv_", "").replace("_kernel", ""), v["kernel_ms_per_step"], v["launches_per_step"]) for n, v in list(k.items())[:6])))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
for d in 100 85 75 65 55; do
  YS_P2_CLS3=$d python bench.py --steps 40 --warmup 8 $S > $O/c2_d${d}_$rep.json 2>/dev/null; show $O/c2_d${d}_$rep.json c2_d$d
done
done
for cfg in "c3:--size s --batch 32" "c4:--family 11 --size m --task segment --batch 32" "c5:--size x --imgsz 1280 --batch 16"; do
  n=${cfg%%:*}; a=${cfg#*:}
  for d in 100 75 55; do
    YS_P2_CLS3=$d python bench.py --steps 15 --warmup 4 $S $a > $O/${n}_d$d.json 2>/dev/null; show $O/${n}_d$d.json ${n}_d$d
  done
done
cd /tmp && export TMPDIR=/tmp
YS_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-nms --no-infer > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/st/*/s_kernel_stats.csv $O/st/s_kernel_stats.csv 2>/dev/null | head -1)
python tools/kernel_stats.py $f 9 > $O/kernel_table.md; grep -i "loss\|tal\|total" $O/kernel_table.md; rm -rf $O/st
