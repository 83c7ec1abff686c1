"""fp8 vs bf16 gradient agreement of YOLOv8s 320x320 B=8 at the second step (first fp8 step), overall and per parameter group.
Run under different env gates (YS_NO_GEMM_F8=1, YS_F8_MIN_CIN=...) to see which kernels / layers the deviation comes from."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import yolo_oracle as O
from yolosharp_amd import Engine
from yolosharp_amd.model import Yolov8, v8DetectionLoss
eng = Engine(0)
B, H, W, nc = 8, 320, 320, 80
x = np.random.default_rng(0).random((B, 3, H, W), dtype=np.float32)
batch = {k: v.numpy() for k, v in O.synthetic_batch(B, H, W, nc, seed=1).items()}
g = {}
for dt in ("fp8", "bf16"):
    m = Yolov8(eng, nc=nc, size="s", height=H, width=W, max_batch=B, dtype=dt)
    m.init_weights(3); m.train()
    crit = v8DetectionLoss(m)
    for it in range(2):
        m.forward(x, fetch=False); _, items = crit(None, batch); m.zero_grad(); m.backward()
        if it == 1:
            g[dt] = m.grads(); print(dt, items)
        m.adamw_step([1e-3] * 3)
    m.close()
g8, gb = g["fp8"], g["bf16"]
num = sum(float((g8[k].ravel() * gb[k].ravel()).sum()) for k in gb)
den = np.sqrt(sum(float((g8[k] ** 2).sum()) for k in gb) * sum(float((gb[k] ** 2).sum()) for k in gb))
print("overall cosine", num / den)
rows = []
for k in gb:
    a, b = g8[k].ravel().astype(np.float64), gb[k].ravel().astype(np.float64)
    na, nb = np.linalg.norm(a), np.linalg.norm(b)
    if nb > 0 and k.endswith("conv.weight"):
        rows.append((float(a @ b / (na * nb + 1e-30)), k, float(nb)))
for c, k, n in rows:
    print("%.3f %-40s |g| %.3e" % (c, k, n))
