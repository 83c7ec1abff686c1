"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only).

  python tools/hbm_traffic.py <fetch_dir>/x_counter_collection.csv <write_dir>/x_counter_collection.csv out.json "<config>"

gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): counter unit is KB, and FETCH_SIZE reports one half of the
bytes of wide coalesced streams -> hbm_bytes = 2*FETCH_SIZE + WRITE_SIZE.  pack_input_kernel (fp32 NCHW images in, known
size) is the in-run calibration point.
"""
import collections
import csv
import json
import re
import sys

CONV_CLASS = ("conv_p2_kernel", "conv_p2_group_kernel", "conv_gemm_kernel", "conv_halo_kernel", "conv_igemm_kernel", "conv3x3_tile_kernel")


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"<.*", "", r["Kernel_Name"]).replace("void ", "").split("(")[0]
        agg[name][0] += 1
        agg[name][1] += float(r["Counter_Value"])
    return agg


def main(fetch_csv, write_csv, out, config):
    """config = bench.py's cfg_key, e.g. "YOLOv8n B=64 640x640 bf16".  Merges into `out` (one entry per configuration) and stamps
    the entry with the sha of the device sources it measured (yolosharp_amd/roofline.py source_sha): bench.py only quotes `traffic`
    from an entry whose sha equals the running tree's."""
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from yolosharp_amd.roofline import source_sha
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        n = max(f[k][0], w[k][0], 1)
        fk, wk = f[k][1] / n, w[k][1] / n
        kernels[k] = {"launches": n, "fetch_kb_per_launch": round(fk, 1), "write_kb_per_launch": round(wk, 1),
                      "hbm_bytes_per_launch_corrected": int((2 * fk + wk) * 1024)}
    n = sum(kernels[k]["launches"] for k in CONV_CLASS if k in kernels)
    tot = sum(kernels[k]["hbm_bytes_per_launch_corrected"] * kernels[k]["launches"] for k in CONV_CLASS if k in kernels)
    total = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in kernels.values())
    res = {"note": __doc__.strip().split("\n\n")[-1].replace("\n", " "), "configs": {}}
    if os.path.exists(out):
        try:
            res = json.load(open(out))
            res.setdefault("configs", {})
        except Exception:
            pass
    res["configs"][config] = {"source_sha": source_sha(root), "traced_launch_bytes_total": int(total),
                              "conv_igemm_class": {"kernels": [k for k in CONV_CLASS if k in kernels], "launches": n,
                                                   "hbm_bytes_per_launch_corrected": int(tot / max(n, 1))},
                              "kernels": kernels}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({"config": config, "source_sha": res["configs"][config]["source_sha"], "traced_GB": round(total / 1e9, 2)}))


if __name__ == "__main__":
    main(*sys.argv[1:5])
