"""bf16-vs-f32 loss drift over 40 AdamW steps (the configuration of test_bf16_training_tracks_f32_over_40_steps)."""
import sys
sys.path.insert(0, ".")
import numpy as np
from oracle import yolo_oracle as O
from yolosharp_amd import Engine
from yolosharp_amd.model import Yolov8, v8DetectionLoss
eng = Engine()
B, H, W, nc, steps = 16, 320, 320, 80, 40
x = np.random.default_rng(0).random((B, 3, H, W), dtype=np.float32)
batch = {k: v.numpy() for k, v in O.synthetic_batch(B, H, W, nc, seed=1, kmax=8).items()}
curves = {}
for dt in ("f32", "bf16"):
    m = Yolov8(eng, nc=nc, size="n", height=H, width=W, max_batch=B, dtype=dt)
    m.init_weights(11); m.train()
    crit = v8DetectionLoss(m)
    rec = []
    for _ in range(steps):
        m.forward(x, fetch=False); _, items = crit(None, batch); m.zero_grad(); m.backward(); m.adamw_step([5e-4] * 3)
        rec.append(float(items.sum()))
    curves[dt] = np.array(rec)
    m.close()
d = np.abs(curves["bf16"] - curves["f32"]) / curves["f32"]
print("f32 ", np.array2string(curves["f32"][[0, 1, 2, 10, 20, 33, 39]], precision=5))
print("bf16", np.array2string(curves["bf16"][[0, 1, 2, 10, 20, 33, 39]], precision=5))
print("f32 head", np.array2string(curves["f32"][:4], precision=4), "bf16 head", np.array2string(curves["bf16"][:4], precision=4))
print("drift max %.4f at %d; mean %.4f" % (d.max(), d.argmax(), d.mean()))
