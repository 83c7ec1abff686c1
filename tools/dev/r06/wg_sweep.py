"""Empirical sweep of conv_wgrad_tr_kernel plans (tile height / width, workgroups per CU) for weight-gradient shapes, against the cost model's own choice.
usage: wg_sweep.py ["B,Cin,H,W,Cout,k,s;..."]"""
import ctypes, os, sys, collections, itertools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from yolosharp_amd import Engine
from yolosharp_amd.blocks import Conv
eng = Engine(0)
force = eng.lib.ys_debug_wgrad_force
force.argtypes = [ctypes.c_int] * 3
rng = np.random.default_rng(0)
S = sys.argv[1] if len(sys.argv) > 1 else "64,64,40,40,64,3,1;64,32,80,80,32,3,1;64,16,160,160,16,3,1;64,80,80,80,80,3,1;64,64,80,80,64,3,1;64,16,320,320,32,3,2;64,32,160,160,64,3,2;64,64,80,80,64,1,1;64,48,160,160,32,1,1;64,32,160,160,32,1,1"
shapes = [tuple(int(v) for v in t.split(",")) for t in S.split(";")]
REPS = int(os.environ.get("REPS", "3"))

def run(shape):
    B, Cin, H, W, Cout, k, s = shape
    m = Conv(eng, Cin, Cout, k, s, height=H, width=W, max_batch=B, dtype="bf16")
    m.init_weights(1); m.train()
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    dy = rng.standard_normal((B, Cout, Ho, Wo), dtype=np.float32)
    m.forward(x)
    eng.kernel_profile(True)
    for rep in range(REPS):
        m.zero_grad(); m.backward(dy, need_dx=False)
    path = "/tmp/wg_sweep_%d.csv" % os.getpid()
    eng.kernel_profile_dump(path)
    eng.kernel_profile(False)
    m.close()
    t = collections.OrderedDict()
    for l in open(path).read().splitlines()[1:]:
        c, rest = l.split(",", 1)
        lab, us = rest.rsplit(",", 1)
        if c == "conv_wgrad":
            t.setdefault(lab, []).append(float(us))
    return [(lab, min(v[1:]) if len(v) > 1 else v[0]) for lab, v in t.items()]

for shape in shapes:
    force(0, 0, 0)
    base = run(shape)
    print("== %s  base: %s" % (shape, base), flush=True)
    if not base or not base[0][0].startswith("wgrad_tr"):
        continue
    t0 = base[0][1]
    res = []
    for th, tws, pc in itertools.product((1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20), (2, 3, 4, 5), (0, 1)):
        force(th, tws, pc)
        try:
            r = run(shape)
        except Exception as e:
            continue
        if len(r) != 1 or not r[0][0].startswith("wgrad_tr"):
            continue
        res.append((r[0][1], th, tws, pc, r[0][0]))
    res.sort()
    seen = set()
    n = 0
    for us, th, tws, pc, lab in res:
        key = lab
        if key in seen: continue
        seen.add(key); n += 1
        print("   %7.1f us (%+5.1f %%)  force th%d tw%d pc%d  %s" % (us, 100.0 * (us - t0) / t0, th, 1 << tws, pc, lab.split(" M")[1] if " M" in lab else lab), flush=True)
        if n >= 6: break
force(0, 0, 0)
