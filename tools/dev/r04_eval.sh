#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/evalp; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st -o s -- python $GRAFT_REPO_ROOT/tools/evalprof.py > $GRAFT_REPO_ROOT/$O/log.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/st/*/s_kernel_stats.csv $O/st/s_kernel_stats.csv 2>/dev/null | head -1)
python tools/kernel_stats.py $f 5 > $O/kernel_table.md; head -24 $O/kernel_table.md; rm -rf $O/st
python tools/launch_report.py gpurun_out/eval_launches.csv 2>/dev/null | head -30
