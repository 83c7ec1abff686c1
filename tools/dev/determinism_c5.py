"""Triage: step-0 determinism of the config-5 graph (YOLOv8x 1280x1280 B=16, training forward + loss): model instances created one
after the other with the same weights / input must give identical loss items.  argv[1] = optional library path, argv[2] = rounds."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import yolo_oracle as O
from yolosharp_amd import Engine
from yolosharp_amd.model import Yolov8, v8DetectionLoss
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] not in ("", "-") else None
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = Engine(0, lib_path=lib) if lib else Engine(0)
B, H, W, nc = 16, 1280, 1280, 80
x = np.random.default_rng(51).random((B, 3, H, W), dtype=np.float32)
nb = {k: v.numpy() for k, v in O.synthetic_batch(B, H, W, nc, seed=52, kmax=8).items()}
out = []
for r in range(rounds):
    m = Yolov8(eng, nc=nc, size="x", height=H, width=W, max_batch=B, dtype="bf16")
    m.init_weights(7); m.train()
    crit = v8DetectionLoss(m)
    for it in range(2):
        m.forward(x, fetch=False); _, items = crit(None, nb)
        out.append((r, it, items.copy()))
        if it == 0:
            m.zero_grad(); m.backward(); m.adamw_step([2e-4] * 3)
    m.close()
for r, it, items in out:
    print("round %d step %d items %s %s" % (r, it, items, "" if np.array_equal(items, out[it][2]) else "<-- differs from round 0"))
