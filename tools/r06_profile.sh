# Round-6 evidence run (same passes as tools/r06_profile.sh) (config 2 unless PROF_ARGS / PROF_TAG are set): rocprofv3 kernel stats, FETCH / WRITE PMC passes (HBM traffic per
# kernel), SQ counter passes, launch dump, bench line.  gpurun --timeout 1800 -- 'bash tools/r06_profile.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${PROF_TAG:-c2}; O=$R/gpurun_out/r06p_$T
mkdir -p $O
A="${PROF_ARGS:-}"
S="--steps 5 --warmup 2 --no-cpu-baseline --no-nms --no-infer"
export YS_OVERLAP=${PROF_OVERLAP:-0}     # per-kernel evidence: every kernel alone on the chip (matches bench.py's roofline profile steps)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py $A $S > $O/stats_bench.log 2>&1
python $R/tools/kernel_stats.py $(ls $O/stats/*/s_kernel_stats.csv $O/stats/s_kernel_stats.csv 2>/dev/null | head -1) 9 > $O/kernel_table.md
cp $(ls $O/stats/*/s_kernel_stats.csv $O/stats/s_kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python $R/bench.py $A --steps 1 --warmup 1 --no-cpu-baseline --no-nms --no-infer > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python $R/bench.py $A --steps 1 --warmup 1 --no-cpu-baseline --no-nms --no-infer > /dev/null 2>&1
PMC_OUT=r06p_$T PMC_BENCH_ARGS="$A" bash $R/tools/pmc.sh > $O/pmc.log 2>&1
python $R/tools/sq_counters.py $O $O/sq_counters.json > $O/sq_summary.txt 2>&1
unset YS_OVERLAP
cd $R
timeout 600 python bench.py $A --dump-launches $O/launches.csv ${PROF_BENCH_EXTRA:-} > $O/bench.json 2> $O/bench.err
# keep the merge small: drop raw traces, keep per-kernel counter CSVs
rm -f $O/stats/*/s_kernel_trace.csv $O/stats/s_kernel_trace.csv $O/fetch/*/*kernel_trace.csv $O/write/*/*kernel_trace.csv $O/fetch/*kernel_trace.csv $O/write/*kernel_trace.csv $O/pmc_*/p_kernel_trace.csv
head -14 $O/kernel_table.md; cat $O/sq_summary.txt | cut -c1-400 | head -4; tail -c 300 $O/bench.json
