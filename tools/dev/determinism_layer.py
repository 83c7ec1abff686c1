"""Triage: run-to-run bit identity of single wide Conv + BN(train) + SiLU layers of the config-5 graph (blocked-GEMM kernel).
argv[1] = optional library path."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from yolosharp_amd import Engine
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] not in ("", "-") else None
eng = Engine(0, lib_path=lib) if lib else Engine(0)
rng = np.random.default_rng(0)
SHAPES = os.environ.get("YS_DET_SHAPES")
shapes = [tuple(int(v) for v in t.split(",")) for t in SHAPES.split(";")] if SHAPES else [(16, 800, 160, 160, 320, 1, 1), (16, 400, 320, 320, 160, 1, 1)]
for (B, Cin, H, W, Cout, k, s) in shapes:
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) / np.sqrt(Cin * k * k)).astype(np.float32)
    outs = []
    raw = [eng.conv_bn_act(x, w, k, s, bn=None, act=False, training=False, dtype="bf16") for _ in range(2)]
    print("   raw conv (no BN): differing reruns %d/5" % sum(0 if np.array_equal(raw[0], r) else 1 for r in raw[1:]))
    REPS = int(os.environ.get('REPS', '6'))
    for rep in range(REPS):
        bn = {"weight": np.ones(Cout, np.float32), "bias": np.zeros(Cout, np.float32), "running_mean": np.zeros(Cout, np.float32), "running_var": np.ones(Cout, np.float32)}
        y = eng.conv_bn_act(x, w, k, s, bn=bn, act=True, training=True, dtype="bf16")
        outs.append((y.copy(), bn["running_mean"].copy(), bn["running_var"].copy()))
    nd_y = sum(0 if np.array_equal(outs[0][0], o[0]) else 1 for o in outs[1:])
    nd_m = sum(0 if np.array_equal(outs[0][1], o[1]) and np.array_equal(outs[0][2], o[2]) else 1 for o in outs[1:])
    worst = max(float(np.abs(outs[0][0] - o[0]).max()) for o in outs[1:])
    nbad = max(int((outs[0][0] != o[0]).sum()) for o in outs[1:])
    for o in outs[1:]:
        if not np.array_equal(outs[0][0], o[0]):
            d = (outs[0][0] != o[0])
            ch = np.nonzero(d.any(axis=(0, 2, 3)))[0]
            print("   channels with differences:", ch[:20], "of", Cout, "| running_mean differs at", np.nonzero(outs[0][1] != o[1])[0][:20])
            break
    print("cin%d cout%d k%d s%d %dx%d: differing reruns y %d (max |d| %.3g, elements %d) running stats %d" % (Cin, Cout, k, s, H, W, nd_y, worst, nbad, nd_m))
