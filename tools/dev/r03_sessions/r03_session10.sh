cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nms -o n -- python $R/tools/dev/nms_prof.py > $O/nms.log 2>&1
python $R/tools/kernel_stats.py $(ls $O/nms/*/n_kernel_stats.csv $O/nms/n_kernel_stats.csv 2>/dev/null | head -1) 20 > $O/nms_table.md; cat $O/nms_table.md
rm -f $O/nms/*/n_kernel_trace.csv $O/nms/n_kernel_trace.csv
