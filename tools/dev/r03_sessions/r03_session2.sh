# Round-3 GPU session 2: where conv_p2_kernel's time goes -- phase ablations (triage build -DYS_P2_ABLATE) and per-phase stamps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b
mkdir -p $O; cd $R
: > $O/abl.txt
for d in 0 4 256 512 1024 2 1 8 16 64 128 6; do
  YS_DBG=$d timeout 120 python bench.py --lib build/libyolosharp_hip_p2abl.so --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-infer > $O/abl_$d.json 2> $O/abl_$d.err
  python - $d $O/abl_$d.json >> $O/abl.txt <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = j["roofline"]
    k = r["kernels"].get("conv_p2_kernel", {})
    print("DBG=%-5s step %7.3f ms   conv_p2 %6.3f ms/step (%s launches, avg %.1f us)   igemm class %.3f" % (sys.argv[1], j["ms_per_step"], k.get("kernel_ms_per_step", 0), k.get("launches_per_step"), 1e3 * k.get("avg_launch_ms", 0), r["class_ms_per_step"]["conv_igemm"]))
except Exception as e:
    print("DBG=%s FAILED %s" % (sys.argv[1], e))
PY
done
cat $O/abl.txt
timeout 200 python tools/dev/p2_timeline.py $O/p2_timeline.txt > /dev/null 2> $O/tl.err
echo done
