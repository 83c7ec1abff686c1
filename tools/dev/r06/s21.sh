cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_s21; mkdir -p $O
timeout 1200 python -m pytest tests/test_kat.py tests/test_model.py tests/test_obb_pose.py tests/test_segment.py tests/test_obb.py -q -m gpu -x --no-header -p no:cacheprovider 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/full -o s -- python $GRAFT_REPO_ROOT/tools/dev/r06/crit_time.py > /dev/null 2>&1
f=$(ls $GRAFT_REPO_ROOT/$O/full/*/s_kernel_stats.csv $GRAFT_REPO_ROOT/$O/full/s_kernel_stats.csv 2>/dev/null | head -1)
python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("tal_", "loss_")):
        print("  %-28s calls %s avg %.1f us" % (n.split("(")[0][:28], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $GRAFT_REPO_ROOT/$O/full
cd $GRAFT_REPO_ROOT
DET_ONLY=production python tools/dev/r05/determinism.py 8 64 4
