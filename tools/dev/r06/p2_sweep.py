"""Empirical sweep of conv_p2_kernel plans (register-tile height, tile width, output-channel split) for the convolution shapes of a configuration, against the
cost model's own choice.  usage: p2_sweep.py ["B,Cin,H,W,Cout,k,s;..."]   (default: the conv_p2 shapes of YOLOv8n B = 64 that cost most per step)"""
import ctypes, os, sys, collections, itertools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from yolosharp_amd import Engine
eng = Engine(0)
force = eng.lib.ys_debug_p2_force
force.argtypes = [ctypes.c_int] * 3
rng = np.random.default_rng(0)
S = sys.argv[1] if len(sys.argv) > 1 else "64,64,40,40,64,3,1;64,32,80,80,32,3,1;64,16,160,160,16,3,1;64,32,160,160,32,1,1;64,48,160,160,32,1,1;64,64,80,80,64,1,1;64,128,40,40,128,1,1;64,80,80,80,80,3,1;64,64,80,80,64,3,1;64,16,320,320,32,3,2;64,32,160,160,64,3,2"
shapes = [tuple(int(v) for v in t.split(",")) for t in S.split(";")]
REPS = int(os.environ.get("REPS", "4"))

def run(shape):
    B, Cin, H, W, Cout, k, s = shape
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) / np.sqrt(Cin * k * k)).astype(np.float32)
    eng.kernel_profile(True)
    for rep in range(REPS):
        bn = {"weight": np.ones(Cout, np.float32), "bias": np.zeros(Cout, np.float32), "running_mean": np.zeros(Cout, np.float32), "running_var": np.ones(Cout, np.float32)}
        eng.conv_bn_act(x, w, k, s, bn=bn, act=True, training=True, dtype="bf16")
    path = "/tmp/p2_sweep_%d.csv" % os.getpid()
    eng.kernel_profile_dump(path)
    eng.kernel_profile(False)
    t = collections.OrderedDict()
    for l in open(path).read().splitlines()[1:]:
        c, rest = l.split(",", 1)
        lab, us = rest.rsplit(",", 1)
        if c == "conv_igemm":
            t.setdefault(lab, []).append(float(us))
    return [(lab, min(v[1:]) if len(v) > 1 else v[0]) for lab, v in t.items()]

for shape in shapes:
    B, Cin, H, W, Cout, k, s = shape
    Wo = (W + 2 * (k // 2) - k) // s + 1
    force(0, 0, 0)
    base = run(shape)
    print("== %s  base: %s" % (shape, base), flush=True)
    if not base or not base[0][0].startswith("p2 "):
        continue
    t0 = base[0][1]
    nfr = (Cout + 15) // 16
    nrs = sorted(set([0] + [n for n in (1, 2, 3, 4, 5) if n <= nfr and nfr % n == 0]))
    tws = sorted(set(t for t in (2, 4, 5, 8, 10, 16, 20, 32, 40, 64, 80) if t <= Wo))
    res = []
    for mr, tw, nr in itertools.product((1, 2, 4), tws, nrs):
        force(mr, tw, nr)
        try:
            r = run(shape)
        except Exception as e:
            continue
        if len(r) != 1 or not r[0][0].startswith("p2 "):
            continue
        res.append((r[0][1], mr, tw, nr, r[0][0]))
    res.sort()
    for us, mr, tw, nr, lab in res[:6]:
        print("   %7.1f us (%+5.1f %%)  force mr%d tw%d nr%d  %s" % (us, 100.0 * (us - t0) / t0, mr, tw, nr, lab.split(" M")[1] if " M" in lab else lab), flush=True)
force(0, 0, 0)
