"""Triage: YOLOv8x 1280 x 1280 B = 2 bf16 train-mode forward + loss on the engine (kernel routing per the environment) against the plain fp32 oracle AND the
rounding-matched oracle (tests/bf16_ref.py): loss items and head-output distances."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import yolo_oracle as O
import bf16_ref as R
from yolosharp_amd import Engine
from yolosharp_amd.model import Yolov8, v8DetectionLoss
from test_model import make_ref
eng = Engine(0)
B, H, W, nc = 2, 1280, 1280, 80
ref = make_ref(nc=nc, size="x", seed=41)
x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(42))
batch = O.synthetic_batch(B, H, W, nc, seed=43)
m = Yolov8(eng, nc=nc, size="x", height=H, width=W, max_batch=B, dtype="bf16")
m.load_state_dict({k: v.detach().numpy() for k, v in ref.state_dict().items()})
m.train(); ref.train()
_, preds = m.forward(x.numpy())
_, items = v8DetectionLoss(m)(None, {k: v.numpy() for k, v in batch.items()})
print("engine items", items)
if os.environ.get("WITH_REF", "1") == "1":
    with torch.no_grad():
        _, rp = R.forward_bf16(ref, x); _, ri = O.v8DetectionLoss(nc)(rp, batch)
        _, fp = ref(x); _, fi = O.v8DetectionLoss(nc)(fp, batch)
    print("rounding-matched oracle items", ri.numpy(), " fp32 oracle items", fi.numpy())
    for k in ("boxes", "scores"):
        a = preds[k].ravel().astype(np.float64); b = rp[k].numpy().ravel().astype(np.float64); c = fp[k].numpy().ravel().astype(np.float64)
        print(k, "rel L2 vs rounding-matched %.4f  vs fp32 %.4f ; rounding-matched vs fp32 %.4f" % (np.linalg.norm(a - b) / np.linalg.norm(b), np.linalg.norm(a - c) / np.linalg.norm(c), np.linalg.norm(b - c) / np.linalg.norm(c)))
np.save(os.environ.get("OUT", "/tmp/c5_scores.npy"), preds["scores"])
