import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolosharp_amd import Engine
from yolosharp_amd.model import Yolov8, v8DetectionLoss
from oracle import yolo_oracle as O
eng = Engine(0)
for size, H in (("x", 1280), ("n", 640), ("s", 640), ("x", 640)):
    B, W, nc = 2, H, 80
    x = np.random.default_rng(0).random((B, 3, H, W), dtype=np.float32)
    batch = {k: v.numpy() for k, v in O.synthetic_batch(B, H, W, nc, seed=43).items()}
    out = {}
    for dt in ("f32", "bf16"):
        m = Yolov8(eng, nc=nc, size=size, height=H, width=W, max_batch=B, dtype=dt)
        m.init_weights(3); m.train()
        _, p = m.forward(x)
        _, items = v8DetectionLoss(m)(None, batch)
        out[dt] = (p["boxes"], p["scores"], items)
        m.close()
    for i, nm in enumerate(("boxes", "scores")):
        a, b = out["bf16"][i], out["f32"][i]
        print(size, H, nm, "max|b|", np.abs(b).max(), "maxdiff", np.abs(a - b).max(), "relL2", np.linalg.norm(a - b) / np.linalg.norm(b),
              "std b", b.std(), "corr", np.corrcoef(a.ravel()[::97], b.ravel()[::97])[0, 1])
    print(size, H, "items", out["bf16"][2], out["f32"][2], flush=True)
