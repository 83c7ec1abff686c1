# Final per-round evidence: kernel stats + HBM traffic passes + bench line.  Run through gpurun from the repo root.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-nms --no-infer > $O/stats_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-nms --no-infer > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-nms --no-infer > /dev/null 2>&1
cd $R && timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
ls $O $O/stats $O/fetch $O/write
