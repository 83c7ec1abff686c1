#!/bin/bash
# tal_metrics_kernel phase ablation: kernel duration with the kernel cut after pass 1 / after pass 2 / whole
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/talabl; mkdir -p $O
for v in talabl1 talabl2 full; do
  la="--lib $R/build/libyolosharp_hip_$v.so"; [ $v = full ] && la=""
  YS_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o s -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-nms --no-infer $la > /dev/null 2>&1
  f=$(ls $O/$v/*/s_kernel_stats.csv $O/$v/s_kernel_stats.csv 2>/dev/null | head -1)
  python - "$f" "$v" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r.get("Name") or r.get("KernelName") or ""
    if "tal_" in n or "loss_" in n:
        print(sys.argv[2], n.split("(")[0][:40], r.get("Calls"), r.get("AverageNs") or r.get("Average"))
PY
  rm -rf $O/$v
done
