"""Triage: the two-epoch Trainer run of tests/test_trainer.py::test_trainer_other_tasks_gpu for one task, optional library path: prints the epoch loss items."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import yolo_oracle as O
from yolosharp_amd import Engine, model as M, trainer as T
task = sys.argv[1] if len(sys.argv) > 1 else "pose"
lib = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] not in ("", "-") else None
eng = Engine(0, lib_path=lib) if lib else Engine(0)
B, H, W = 8, 128, 128
nc = {"obb": 15, "pose": 1, "segment": 80}[task]
for seed in (1, 2, 3):
    m = {"obb": M.Yolov8Obb, "pose": M.Yolov8Pose, "segment": M.Yolov8Segment}[task](eng, nc=nc, size="n", height=H, width=W, max_batch=B, dtype="bf16")
    m.init_weights(seed)
    tb = O.synthetic_obb_batch(B, H, W, nc, seed=1, kmax=4) if task == "obb" else O.synthetic_batch(B, H, W, nc, seed=1, kmax=4)
    if task == "pose": tb["keypoints"] = O.synthetic_keypoints(tb)
    if task == "segment": tb["masks"] = O.synthetic_masks(tb, B, H // 4, W // 4)
    data = {k: v.numpy() for k, v in tb.items()}
    data["images"] = np.random.default_rng(0).random((B, 3, H, W), dtype=np.float32)
    with tempfile.TemporaryDirectory() as d:
        tr = T.Trainer(m, epochs=2, nb=6, out_dir=d, lr0=2e-3, warmup_bias_lr=2e-3)
        hist = tr.fit(lambda: [data] * 6, lambda: [data])
    print(task, "seed", seed, "epoch0", hist[0]["train_loss"], hist[0]["train_loss"].sum(), "epoch1", hist[1]["train_loss"], hist[1]["train_loss"].sum())
    m.close()
