cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_s14; mkdir -p $O
python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -c 1500 $O/bench_full.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/ev -o s -- python $GRAFT_REPO_ROOT/tools/evalprof.py > $GRAFT_REPO_ROOT/$O/evalprof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/ev/*/s_kernel_stats.csv $O/ev/s_kernel_stats.csv 2>/dev/null | head -1)
python tools/kernel_stats.py $f 5 > $O/eval_kernel_table.md; cat $O/eval_kernel_table.md | head -20; cp gpurun_out/eval_launches.csv $O/ 2>/dev/null; rm -rf $O/ev
python tools/launch_report.py $O/eval_launches.csv 2>/dev/null | head -25
