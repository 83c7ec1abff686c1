# Round-3 GPU session 1: correctness of the new paths on the device, then A/B of every new switch on BASELINE config 2.
# Run through gpurun from the repo root:  gpurun --timeout 1500 -- 'bash tools/r03_session1.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
echo "== tests (new paths first)" > $O/log.txt
timeout 600 python -m pytest tests/test_bnred.py tests/test_conv.py tests/test_model.py -x -q -m gpu >> $O/log.txt 2>&1
echo "== bench default" >> $O/log.txt
timeout 400 python bench.py --steps 30 --warmup 10 > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.json >> $O/log.txt
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 200 $B > $O/ab_$tag.json 2> $O/ab_$tag.err
  python - "$tag" $O/ab_$tag.json >> $O/ab.txt <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print("%-28s %8.3f ms/step  %8.1f img/s  p2 %.3f ms (%s frac %.3f)  igemm %.3f wgrad %.3f" % (sys.argv[1], j["ms_per_step"], j["value"], r["kernels"].get("conv_p2_kernel", {}).get("kernel_ms_per_step", 0), r["kernel"], r["frac"], r["class_ms_per_step"]["conv_igemm"], r["class_ms_per_step"]["conv_wgrad"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
: > $O/ab.txt
run all_on X=1
run round2_rules YS_P2_PITCH=0 YS_P2_ROWPAD=0 YS_P2_WPITCH=0 YS_BNRED=0 YS_WGRED_DEFER=0
run no_pitch YS_P2_PITCH=0 YS_P2_ROWPAD=0
run no_rowpad YS_P2_ROWPAD=0
run no_wpitch YS_P2_WPITCH=0
run no_bnred YS_BNRED=0
run no_wgdefer YS_WGRED_DEFER=0
run all_on_again X=1
cat $O/ab.txt >> $O/log.txt
# kernel table of the default build
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-nms --no-infer > $O/stats_bench.log 2>&1
python $R/tools/kernel_stats.py $(ls $O/stats/*/s_kernel_stats.csv $O/stats/s_kernel_stats.csv 2>/dev/null | head -1) 9 > $O/kernel_table.md 2>> $O/log.txt
head -30 $O/kernel_table.md >> $O/log.txt
# SQ counters (LDS bank conflicts, MFMA busy, waits) of the default build: tools/pmc.sh passes, aggregated by tools/sq_counters.py
PMC_OUT=r03a bash $R/tools/pmc.sh >> $O/log.txt 2>&1
python $R/tools/sq_counters.py $O $O/sq_counters.json >> $O/log.txt 2>&1
# per-phase cycle stamps of conv_p2_kernel workgroups (timeline build), new pitch rules vs the round-2 ones
cd $R
timeout 200 python tools/dev/p2_timeline.py $O/p2_timeline_new.txt > /dev/null 2>> $O/log.txt
YS_P2_PITCH=0 YS_P2_ROWPAD=0 YS_P2_WPITCH=0 timeout 200 python tools/dev/p2_timeline.py $O/p2_timeline_old.txt > /dev/null 2>> $O/log.txt
echo done >> $O/log.txt
