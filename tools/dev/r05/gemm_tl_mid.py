"""s_memtime phase stamps of conv_gemm_kernel on the mid-size launches that dominate BASELINE configs 3 and 4 (needs build/libyolosharp_hip_tl.so)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "gemm_tl_mid.txt")
os.makedirs(os.path.dirname(out), exist_ok=True)
if os.path.exists(out):
    os.remove(out)
os.environ["YS_P2_TL"] = out
os.environ.setdefault("YS_GEMM_HALO", "0")
from yolosharp_amd import Engine
eng = Engine(0, lib_path=os.path.join(ROOT, "build", "libyolosharp_hip_tl.so"))
rng = np.random.default_rng(0)
S = os.environ.get("YS_LB_SHAPES")
shapes = [tuple(int(v) for v in t.split(",")) for t in S.split(";")] if S else [(32, 128, 40, 40, 128, 3, 1), (32, 256, 80, 80, 256, 1, 1), (32, 256, 40, 40, 384, 1, 1), (32, 256, 20, 20, 256, 3, 1), (32, 128, 20, 20, 128, 3, 1)]
for (B, Cin, H, W, Cout, k, s) in shapes:
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) / np.sqrt(Cin * k * k)).astype(np.float32)
    bn = {"weight": np.ones(Cout, np.float32), "bias": np.zeros(Cout, np.float32), "running_mean": np.zeros(Cout, np.float32), "running_var": np.ones(Cout, np.float32)}
    for rep in range(2):
        eng.conv_bn_act(x, w, k, s, bn=bn, act=True, training=True, dtype="bf16")
# summary: per launch (second repetition), median over sampled workgroups of: table, first request, then per tile K loop / epilogue
txt = open(out).read().split("# gemm")
for blk in txt[1:]:
    lines = blk.strip().splitlines()
    hdr = lines[0]
    rows = [[int(v) for v in l.split(":")[1].split()] for l in lines[1:] if ":" in l]
    if not rows:
        continue
    print("gemm" + hdr.split("(stamps")[0])
    for r in rows[:6]:
        d = [r[0]] + [r[i] - r[i - 1] for i in range(1, len(r))]
        print("   ", " ".join("%6d" % v for v in d[:14]), "| exit", r[-1])
