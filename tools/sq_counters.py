"""Aggregate the rocprofv3 PMC passes of tools/pmc.sh (gpurun_out/pmc_*/p_counter_collection.csv) per kernel family.

  python tools/sq_counters.py gpurun_out profiles/r01_sq_counters.json

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES and
GRBM_GUI_ACTIVE count cycles (GRBM summed over the 8 XCDs).  mfma_util = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs).
Every pass traces the same 4 training steps, so ratios across passes are per-launch ratios of the same launches.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def main(root, out):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    for path in sorted(glob.glob(os.path.join(root, "pmc_*", "p_counter_collection.csv"))):
        for r in csv.DictReader(open(path)):
            name = re.sub(r"<.*", "", r["Kernel_Name"]).replace("void ", "").split("(")[0]
            tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[name][r["Counter_Name"]] += 1
    res = {}
    for k, c in tot.items():
        n = max(launches[k].values())
        if n < 4 or "SQ_WAVE_CYCLES" not in c:
            continue
        wc = c["SQ_WAVE_CYCLES"] or 1.0
        waves = c.get("SQ_WAVES", 0.0) or 1.0
        e = {"launches": n, "waves_per_launch": round(waves / launches[k]["SQ_WAVES"], 1) if launches[k].get("SQ_WAVES") else None}
        for key, cn in (("wait_any_frac", "SQ_WAIT_ANY"), ("wait_inst_any_frac", "SQ_WAIT_INST_ANY"),
                        ("active_inst_any_frac", "SQ_ACTIVE_INST_ANY"), ("active_inst_valu_frac", "SQ_ACTIVE_INST_VALU"),
                        ("active_inst_lds_frac", "SQ_ACTIVE_INST_LDS"), ("wait_inst_lds_frac", "SQ_WAIT_INST_LDS")):
            if cn in c:
                e[key] = round(c[cn] / wc, 4)
        for key, cn in (("valu_insts_per_wave", "SQ_INSTS_VALU"), ("salu_insts_per_wave", "SQ_INSTS_SALU"),
                        ("lds_insts_per_wave", "SQ_INSTS_LDS"), ("vmem_rd_insts_per_wave", "SQ_INSTS_VMEM_RD"),
                        ("smem_insts_per_wave", "SQ_INSTS_SMEM")):
            if cn in c:
                e[key] = round(c[cn] / waves, 1)
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_ratio"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
            e["mfma_util"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
        res[k] = e
    note = ("rocprofv3 --kernel-trace --pmc <5 SQ/GRBM counters per pass> (tools/pmc.sh), bench.py --steps 1 --warmup 1 --no-infer "
            "(4 train steps traced); sums over all launches of a kernel family. Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / "
            "SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE count cycles (GRBM summed over the 8 XCDs). "
            "mfma_util = MFMA_BUSY / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs).")
    json.dump({"note": note, "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
    for k in ("conv_p2_kernel", "conv_wgrad_tr_kernel", "chan_reduce_kernel", "bn_bwd_apply_kernel", "bn_act_apply_kernel"):
        if k in res:
            print(k, res[k])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
