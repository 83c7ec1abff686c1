#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/nms; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st -o s -- python $GRAFT_REPO_ROOT/tools/dev/nms_prof.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/st/*/s_kernel_stats.csv $O/st/s_kernel_stats.csv 2>/dev/null | head -1)
python tools/kernel_stats.py $f 20 > $O/kernel_table.md; head -12 $O/kernel_table.md; rm -rf $O/st
