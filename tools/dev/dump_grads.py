"""Dump the bf16 first-step head gradients and parameter gradients (npz) -- for A/B comparisons between two builds."""
import sys
sys.path.insert(0, ".")
import numpy as np
from oracle import yolo_oracle as O
from yolosharp_amd import Engine
from yolosharp_amd.model import Yolov8, v8DetectionLoss
eng = Engine()
B, H, W, nc = 16, 320, 320, 80
x = np.random.default_rng(0).random((B, 3, H, W), dtype=np.float32)
batch = {k: v.numpy() for k, v in O.synthetic_batch(B, H, W, nc, seed=1, kmax=8).items()}
m = Yolov8(eng, nc=nc, size="n", height=H, width=W, max_batch=B, dtype="bf16")
m.init_weights(11); m.train()
_, preds = m.forward(x)
_, items = v8DetectionLoss(m)(None, batch)
out = {"items": items, "boxes": preds["boxes"], "scores": preds["scores"], "dboxes": m.get_output("dboxes"), "dscores": m.get_output("dscores")}
m.zero_grad(); m.backward()
for k, v in m.grads().items():
    out["g:" + k] = v
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1], items)
