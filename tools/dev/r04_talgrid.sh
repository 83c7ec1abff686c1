#!/bin/bash
# tal_metrics_kernel duration against the flat grid size (YS_TAL_GRID)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/talgrid; mkdir -p $O
for n in ${TAL_GRIDS:-128 256 544 768 1536}; do
  YS_TAL_GRID=$n YS_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g$n -o s -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-nms --no-infer ${TAL_LIB:+--lib $R/build/libyolosharp_hip_$TAL_LIB.so} > /dev/null 2>&1
  f=$(ls $O/g$n/*/s_kernel_stats.csv $O/g$n/s_kernel_stats.csv 2>/dev/null | head -1)
  python - "$f" "$n" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r.get("Name") or ""
    if "tal_metrics" in n: print("grid", sys.argv[2], r.get("Calls"), r.get("AverageNs"))
PY
  rm -rf $O/g$n
done
