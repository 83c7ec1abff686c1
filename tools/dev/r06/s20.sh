cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_s20; mkdir -p $O
for v in full tal1 tal2; do
  lib=""; [ $v != full ] && lib=$R/build/libyolosharp_hip_$v.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o s -- python $R/tools/dev/r06/crit_time.py $lib > $O/$v.log 2>&1
  f=$(ls $O/$v/*/s_kernel_stats.csv $O/$v/s_kernel_stats.csv 2>/dev/null | head -1)
  echo "== $v"; python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("tal_", "loss_")):
        print("  %-28s calls %s avg %.1f us" % (n.split("(")[0][:28], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $O/$v
done
