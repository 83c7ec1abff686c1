# SQ counters of the convolution kernels of tools/dev/r05/layer_bench.py (single layers): gpurun -- 'bash tools/dev/r05/layer_pmc.sh [lib]'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_layer_pmc; rm -rf $O; mkdir -p $O
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  REPS=4 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$tag -o p -- python $R/tools/dev/r05/layer_bench.py $1 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, re
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for path in sorted(glob.glob("$O/pmc_*/**/p_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "conv_halo" not in k and "conv_gemm_kernel" not in k: continue
        name = re.sub(r"\(.*", "", k).replace("void ", "") + " grid" + r.get("Grid_Size", "")
        tot[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[name][r["Counter_Name"]] += 1
for k, c in tot.items():
    wc = c["SQ_WAVE_CYCLES"] or 1.0
    print(k)
    print("   launches %d  wait_any %.3f  wait_inst_any %.3f  active_inst_any %.3f  active_valu %.3f  active_lds %.3f  wait_inst_lds %.3f" % (cnt[k]["SQ_WAVES"], c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_ACTIVE_INST_VALU"] / wc, c["SQ_ACTIVE_INST_LDS"] / wc, c["SQ_WAIT_INST_LDS"] / wc))
    print("   lds_bank_conflict / lds_idx_active %.3f   mfma_util %.3f   lds_idx_active / (gui/8*256 CUs) %.3f   insts per wave: valu %.0f salu %.0f lds %.0f vmem %.0f" % (
        c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), c["SQ_LDS_IDX_ACTIVE"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 256.0),
        c["SQ_INSTS_VALU"] / c["SQ_WAVES"], c["SQ_INSTS_SALU"] / c["SQ_WAVES"], c["SQ_INSTS_LDS"] / c["SQ_WAVES"], c["SQ_INSTS_VMEM_RD"] / c["SQ_WAVES"]))
PY
