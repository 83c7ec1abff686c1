"""Triage: s_memtime phase stamps of conv_p2_kernel workgroups for single layers (needs build/libyolosharp_hip_tl.so =
python -m yolosharp_amd.build timeline).  Prints, per recorded workgroup, cumulative shader cycles at: prologue issued, then per
tile (top, patch in LDS, MFMA loop done, epilogue done), exit."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r2", "p2_timeline.txt")
os.makedirs(os.path.dirname(out), exist_ok=True)
if os.path.exists(out):
    os.remove(out)
os.environ["YS_P2_TL"] = out
from yolosharp_amd import Engine
eng = Engine(0, lib_path=os.path.join(ROOT, "build", "libyolosharp_hip_tl.so"))
rng = np.random.default_rng(0)
SHAPES = os.environ.get("YS_TL_SHAPES")
shapes = [tuple(int(v) for v in t.split(",")) for t in SHAPES.split(";")] if SHAPES else None
for (B, Cin, H, W, Cout, k, s) in shapes or [(64, 64, 40, 40, 64, 3, 1), (64, 128, 20, 20, 128, 3, 1), (64, 32, 80, 80, 32, 3, 1), (64, 384, 20, 20, 256, 1, 1),
                                   (64, 16, 160, 160, 16, 3, 1), (64, 64, 80, 80, 64, 3, 1), (64, 64, 80, 80, 64, 1, 1), (64, 32, 160, 160, 32, 1, 1)]:
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) / np.sqrt(Cin * k * k)).astype(np.float32)
    bn = {"weight": np.ones(Cout, np.float32), "bias": np.zeros(Cout, np.float32), "running_mean": np.zeros(Cout, np.float32), "running_var": np.ones(Cout, np.float32)}
    for rep in range(2):
        eng.conv_bn_act(x, w, k, s, bn=bn, act=True, training=True, dtype="bf16")
print(open(out).read())
