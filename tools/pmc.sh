# usage: bash tools/pmc.sh            (BASELINE config 2)
#        PMC_OUT=c5 PMC_BENCH_ARGS="--size x --imgsz 1280 --batch 16" bash tools/pmc.sh    (another configuration -> gpurun_out/c5/pmc_*)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/${PMC_OUT:-.}/pmc_$tag -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-nms --no-infer $PMC_BENCH_ARGS > /dev/null 2>&1
  ls $R/gpurun_out/${PMC_OUT:-.}/pmc_$tag | head -3
done
