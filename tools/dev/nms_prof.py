"""NMS-only run for rocprofv3 (the [64, 84, 8400] tensor of bench.py): 20 calls of ys_nms_batched on device-resident predictions."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from yolosharp_amd import Engine
eng = Engine(0)
prng = np.random.default_rng(3)
A, nc = 8400, 80
wh = prng.uniform(0.03, 0.6, (64, 2, A)) * 640; c = prng.uniform(0, 640, (64, 2, A))
sc = 1 / (1 + np.exp(-prng.normal(-3, 1.5, (64, nc, A))))
sc[:, :, prng.random(A) < 0.92] *= 0.05
pred = np.concatenate([c, wh, sc], 1).astype(np.float32)
d_pred = eng.malloc(pred.nbytes)
d_rows = eng.malloc(64 * 300 * 6 * 4); d_keep = eng.malloc(64 * 300 * 8); d_cnt = eng.malloc(64 * 4)
for it in range(20):
    eng.lib.ys_memcpy_h2d(eng.ctx, d_pred, pred.ctypes.data_as(C.c_void_p), pred.nbytes)
    eng.synchronize()
    eng.nms_device(d_pred, 64, 84, A, 0.25, 0.45, 300, 0, d_rows, d_keep, d_cnt)
    eng.synchronize()
