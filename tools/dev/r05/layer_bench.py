"""Per-launch time of single convolution layers (forward, training-mode BN) through the engine's launch records.
usage: python tools/dev/r05/layer_bench.py [lib] ; YS_LB_SHAPES="B,Cin,H,W,Cout,k,s;..." ; REPS=5"""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from yolosharp_amd import Engine
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] not in ("", "-") else None
eng = Engine(0, lib_path=lib) if lib else Engine(0)
rng = np.random.default_rng(0)
S = os.environ.get("YS_LB_SHAPES")
shapes = [tuple(int(v) for v in t.split(",")) for t in S.split(";")] if S else [(16, 320, 80, 80, 320, 3, 1), (16, 160, 160, 160, 160, 3, 1), (32, 256, 80, 80, 256, 3, 1)]
reps = int(os.environ.get("REPS", "5"))
eng.kernel_profile(True)
for (B, Cin, H, W, Cout, k, s) in shapes:
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) / np.sqrt(Cin * k * k)).astype(np.float32)
    for rep in range(reps):
        bn = {"weight": np.ones(Cout, np.float32), "bias": np.zeros(Cout, np.float32), "running_mean": np.zeros(Cout, np.float32), "running_var": np.ones(Cout, np.float32)}
        eng.conv_bn_act(x, w, k, s, bn=bn, act=True, training=True, dtype="bf16")
path = "/tmp/layer_bench_%d.csv" % os.getpid()
eng.kernel_profile_dump(path)
agg = collections.OrderedDict()
for l in open(path).read().splitlines()[1:]:
    c, lab, us = l.rsplit(",", 2)[0].split(",", 1)[0], l.split(",", 1)[1].rsplit(",", 1)[0], float(l.rsplit(",", 1)[1])
    if c != "conv_igemm": continue
    agg.setdefault(lab, []).append(us)
for lab, v in agg.items():
    t = lab.split()
    cin = int([x for x in t if x.startswith("cin")][0][3:]); cout = int([x for x in t if x.startswith("cout")][0][4:]); M = int([x for x in t if x.startswith("M")][0][1:])
    kk = int(t[1][1]) * int(t[1][2])
    best = min(v[1:]) if len(v) > 1 else v[0]
    print("%-100s min %8.1f us  med %8.1f  %6.0f TFLOP/s" % (lab[:100], best, sorted(v)[len(v) // 2], 2.0 * M * cin * kk * cout / best / 1e6))
