# Round-6 evidence: tools/r06_profile.sh (kernel stats, FETCH / WRITE PMC passes, SQ counters, launch dump, bench line) for BASELINE configs 2-5 on ONE box, summaries
# copied to gpurun_out/r06_profiles/ (-> profiles/r06_*).   gpurun --timeout 3000 -- 'bash tools/r06_all.sh'
cd $GRAFT_REPO_ROOT; P=gpurun_out/r06_profiles; mkdir -p $P
run() {   # tag, bench args, config key of tools/hbm_traffic.py
  PROF_TAG=$1 PROF_ARGS="$2" PROF_BENCH_EXTRA="--no-cpu-baseline --no-nms --no-infer --steps 20 --warmup 5" bash tools/r06_profile.sh > $P/$1.log 2>&1
  O=gpurun_out/r06p_$1
  cp $O/kernel_table.md $P/r06_$1_kernel_table.md; cp $O/sq_counters.json $P/r06_$1_sq_counters.json; cp $O/launches.csv $P/r06_$1_launches.csv; cp $O/bench.json $P/r06_$1_bench.json; cp $O/kernel_stats.csv $P/r06_$1_kernel_stats.csv
  python tools/hbm_traffic.py $(ls $O/fetch/*/f_counter_collection.csv $O/fetch/f_counter_collection.csv 2>/dev/null | head -1) $(ls $O/write/*/w_counter_collection.csv $O/write/w_counter_collection.csv 2>/dev/null | head -1) $P/r06_hbm_traffic.json "$3" >> $P/$1.log 2>&1
  tail -1 $P/$1.log; head -8 $O/kernel_table.md
}
run c2 "" "YOLOv8n B=64 640x640 bf16"
run c5 "--size x --imgsz 1280 --batch 16" "YOLOv8x B=16 1280x1280 bf16"
run c5f8 "--size x --imgsz 1280 --batch 16 --dtype fp8" "YOLOv8x B=16 1280x1280 fp8"
run c4 "--family 11 --size m --task segment --batch 32" "YOLOv11m-segment B=32 640x640 bf16"
run c3 "--size s --batch 32" "YOLOv8s B=32 640x640 bf16"
