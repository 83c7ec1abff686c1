"""Triage harness: a few YOLOv8n (B = 64) conv layers through Engine.conv_bn_act, one launch each after a warm-up, for per-dispatch
counter runs (rocprofv3 --pmc ...).  argv[1] = library (default: the product build)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from yolosharp_amd import Engine
eng = Engine(0, lib_path=sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else Engine(0)
rng = np.random.default_rng(0)
for (B, Cin, H, W, Cout, k, s) in [(64, 32, 80, 80, 32, 3, 1), (64, 64, 40, 40, 64, 3, 1), (64, 16, 160, 160, 16, 3, 1), (64, 64, 80, 80, 64, 3, 1), (64, 64, 80, 80, 80, 3, 1)]:
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) / np.sqrt(Cin * k * k)).astype(np.float32)
    bn = {"weight": np.ones(Cout, np.float32), "bias": np.zeros(Cout, np.float32), "running_mean": np.zeros(Cout, np.float32), "running_var": np.ones(Cout, np.float32)}
    for rep in range(2):
        eng.conv_bn_act(x, w, k, s, bn=bn, act=True, training=True, dtype="bf16")
