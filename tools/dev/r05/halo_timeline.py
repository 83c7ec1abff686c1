"""Triage: s_memtime phase stamps of conv_halo_kernel workgroups (needs build/libyolosharp_hip_tl.so = python -m yolosharp_amd.build timeline)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "halo_timeline.txt")
os.makedirs(os.path.dirname(out), exist_ok=True)
if os.path.exists(out):
    os.remove(out)
os.environ["YS_P2_TL"] = out
from yolosharp_amd import Engine
eng = Engine(0, lib_path=os.path.join(ROOT, "build", "libyolosharp_hip_tl.so"))
rng = np.random.default_rng(0)
for (B, Cin, H, W, Cout, k, s) in [(16, 320, 80, 80, 320, 3, 1), (16, 160, 160, 160, 160, 3, 1)]:
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) / np.sqrt(Cin * k * k)).astype(np.float32)
    bn = {"weight": np.ones(Cout, np.float32), "bias": np.zeros(Cout, np.float32), "running_mean": np.zeros(Cout, np.float32), "running_var": np.ones(Cout, np.float32)}
    for rep in range(2):
        eng.conv_bn_act(x, w, k, s, bn=bn, act=True, training=True, dtype="bf16")
txt = open(out).read().splitlines()
for hdr in [i for i, l in enumerate(txt) if l.startswith("#")][1::2]:      # second launch of every layer
    print(txt[hdr])
    for l in txt[hdr + 1: hdr + 4]:
        if l.startswith("#"): break
        v = [int(t) for t in l.split(":")[1].split()]
        print(l.split(":")[0], "deltas:", [v[0]] + [b - a for a, b in zip(v, v[1:])])
