"""Triage of conv_gemm_kernel on single layers (needs build/libyolosharp_hip_abl.so = python -m yolosharp_amd.build ablate):
launch time with the A / B operand traffic, the MFMAs or the epilogue switched off (YS_GEMM_DBG bits, see conv_gemm.hip)."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("YS_GEMM_MIN_M", "1")
from yolosharp_amd import Engine
eng = Engine(0, lib_path=os.path.join(ROOT, "build", "libyolosharp_hip_abl.so"))
rng = np.random.default_rng(0)
LAYERS = [(16, 320, 80, 80, 320, 3, 1), (16, 160, 160, 160, 160, 3, 1), (16, 320, 160, 160, 1280, 1, 1), (32, 256, 80, 80, 256, 3, 1)]
for (B, Cin, H, W, Cout, k, s) in LAYERS:
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) / np.sqrt(Cin * k * k)).astype(np.float32)
    bn = {"weight": np.ones(Cout, np.float32), "bias": np.zeros(Cout, np.float32), "running_mean": np.zeros(Cout, np.float32), "running_var": np.ones(Cout, np.float32)}
    fl = 2.0 * B * (H // s) * (W // s) * Cout * Cin * k * k
    res = []
    for dbg in (0, 1, 2, 3, 4, 8, 12, 7):
        os.environ["YS_GEMM_DBG"] = str(dbg)
        best = 1e9
        for rep in range(3):
            eng.kernel_profile(True)
            eng.conv_bn_act(x, w, k, s, bn=bn, act=True, training=True, dtype="bf16")
            with tempfile.NamedTemporaryFile(suffix=".csv", delete=False) as f:
                path = f.name
            eng.kernel_profile_dump(path)
            eng.kernel_profile(False)
            us = [float(l.split(",")[-1]) for l in open(path).read().splitlines()[1:] if l.startswith("conv_igemm,gemm")]
            os.remove(path)
            if us:
                best = min(best, us[0])
        res.append((dbg, best))
    print("k%d cin%d cout%d M%d:" % (k, Cin, Cout, B * (H // s) * (W // s)), " ".join("dbg%d %.0fus(%.0fTF)" % (d, u, fl / u / 1e6) for d, u in res))
