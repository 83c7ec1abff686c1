"""Per-kernel roofline bookkeeping for bench.py (measurement only, no device code).

The engine records every convolution launch of a profiled step with HIP events on its own stream and a label that carries the
launch geometry (csrc/conv.hip, conv_gemm.hip, conv_wgrad*.hip: `ys_kprof_*`).  From the label alone the ALGORITHMIC work of a
launch follows SURVEY.md 8d's per-unit figure -- a convolution launch moves its input tensor once and its output tensor once:

    bytes = (input pixels * Cin + output pixels * Cout) * element size        flop = 2 * output pixels * Cout * Cin * KH * KW

so a kernel's roofline line is  sum(bytes or flop of its launches) / sum(their event durations)  against the peak that bounds it
(MI355X_MICROARCH.md: HBM 8 TB/s; dense MFMA 2.5 PFLOP/s bf16, 5 PFLOP/s fp8), the bound chosen by the kernel's own arithmetic
intensity against the ridge point.  Weight-gradient launches: input = the layer input + dy, output = dW (negligible)."""
import csv
import hashlib
import os
import re

HBM_PEAK_GBS = 8000.0
MFMA_PEAK_TF = {"bf16": 2500.0, "fp8": 5000.0, "f32": 157.3}

KERNELS = {   # label prefix -> (kernel symbol, operand type of its MFMAs)
    "p2": ("conv_p2_kernel", "bf16"), "p2f8": ("conv_p2_kernel<F8>", "fp8"),
    "p2grp": ("conv_p2_group_kernel", "bf16"),     # grouped launch (label "p2grpN ... M<sum of the problems' pixels>"): same channels and taps for every problem
    "gemm": ("conv_gemm_kernel", "bf16"), "gemmf8": ("conv_gemm_kernel<F8>", "fp8"),
    "halo": ("conv_halo_kernel", "bf16"),          # round 5: halo-patch form of the 3x3 stride-1 wide layers (csrc/conv_halo.h)
    "direct": ("conv_igemm_kernel", "bf16"), "patch": ("conv3x3_tile_kernel", "bf16"),
    "stem": ("stem_fwd_kernel", "bf16"), "wstem": ("stem_wgrad_kernel", "bf16"),   # model.0 straight from the fp32 NCHW image (conv_stem.hip)
    "wgrad_tr": ("conv_wgrad_tr_kernel", "bf16"), "wgemm": ("conv_wgrad_gemm_kernel", "bf16"), "wgrad": ("conv_wgrad_kernel", "bf16"),
}
_LAB = re.compile(r"^(\w+) k(\d+) s(\d+) (?:div(\d+) )?cin(\d+) cout(\d+) M(\d+)")


def launch_work(label, elem_bytes=2):
    """(kernel symbol, operand type, algorithmic bytes, flop) of one labelled launch; None for labels without geometry."""
    m = _LAB.match(label)
    if not m or re.sub(r"(?<=grp)\d+$", "", m.group(1)) not in KERNELS:
        return None
    kind, k, s, div, cin, cout, M = re.sub(r"(?<=grp)\d+$", "", m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4) or 1), int(m.group(5)), int(m.group(6)), int(m.group(7))
    kh, kw = (k // 10, k % 10) if k >= 10 else (k, k)          # "k33" style (conv) or "k3" (wgrad / round-1 kernels)
    sym, op = KERNELS[kind]
    in_px = M * s * s / (div * div)
    cin_alg = 3 if (cin == 8 and kh == 3 and s == 2) else cin   # the stem's 3 input channels are padded to 8 in memory
    flop = 2.0 * M * cout * cin_alg * kh * kw / (div * div)
    if kind in ("stem", "wstem"):                               # the image is fp32 in HBM (the boundary's own tensor), the output / dy bf16
        byt = in_px * cin_alg * 4 + M * cout * elem_bytes
    elif kind.startswith("w"):                                  # weight gradient: reads x (in_px * cin) and dy (M * cout)
        byt = (in_px * cin_alg + M * cout) * elem_bytes
    else:
        byt = (in_px * cin_alg + M * cout) * elem_bytes
    return sym, op, byt, flop


def per_kernel(csv_path, steps, elem_bytes=2):
    """Aggregate a `ys_ctx_kernel_profile_dump` CSV: {symbol: dict(launches, ms, bytes, flop, op)} per STEP."""
    agg = {}
    for row in csv.DictReader(open(csv_path)):
        w = launch_work(row["label"], elem_bytes)
        if w is None:
            continue
        sym, op, byt, flop = w
        a = agg.setdefault(sym, {"launches": 0, "ms": 0.0, "bytes": 0.0, "flop": 0.0, "op": op, "class": row["class"]})
        a["launches"] += 1; a["ms"] += float(row["us"]) * 1e-3; a["bytes"] += byt; a["flop"] += flop
    for a in agg.values():
        for k in ("launches", "ms", "bytes", "flop"):
            a[k] = a[k] / steps
    return agg


def roofline_of(sym, a):
    """Roofline object fields of one kernel aggregate (per step)."""
    ai = a["flop"] / max(a["bytes"], 1.0)
    peak_tf = MFMA_PEAK_TF[a["op"]]
    ridge = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
    n = max(a["launches"], 1e-9)
    avg_ms = a["ms"] / n
    if ai > ridge:
        ach = a["flop"] / n / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        r = {"bound": "mfma", "achieved": round(ach, 1), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(ach / peak_tf, 4)}
    else:
        ach = a["bytes"] / n / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        r = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
    r.update({"kernel": sym, "launches_per_step": round(a["launches"], 1), "avg_launch_ms": round(avg_ms, 5),
              "algorithmic_bytes_per_launch": int(a["bytes"] / n), "algorithmic_flop_per_launch": int(a["flop"] / n),
              "arithmetic_intensity": round(ai, 1), "ridge": round(ridge, 1), "kernel_ms_per_step": round(a["ms"], 3)})
    return r


def source_sha(root):
    """sha256 over the device sources (csrc/*.hip, *.h and the ABI header): ties a committed PMC file to the code it measured."""
    h = hashlib.sha256()
    d = os.path.join(root, "yolosharp_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../include/yolosharp_hip.h"]:
        p = os.path.join(d, f)
        if os.path.isfile(p) and (f.endswith(".hip") or f.endswith(".h")):
            h.update(f.encode()); h.update(open(p, "rb").read())
    return h.hexdigest()[:16]
