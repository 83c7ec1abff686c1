"""Generates the committed fixtures under tests/golden/ from the oracle (run once; outputs are data only).

PARITY UNPINNED: the reference (C#/TorchSharp) has no tests or vectors for this path and cannot run here, so the
expected outputs below come from oracle/ (the restatement), not from the reference itself."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))


def nms_golden():
    from yolosharp_amd import build
    lib = C.CDLL(build.build_oracle())
    rng = np.random.default_rng(7)
    B, A, nc = 2, 320, 80
    wh = rng.uniform(0.03, 0.6, (B, 2, A)) * 640
    c = rng.uniform(0, 640, (B, 2, A))
    sc = np.round(1 / (1 + np.exp(-rng.normal(-3, 1.5, (B, nc, A)))) * 64) / 64   # deliberate score ties
    pred = np.concatenate([c, wh, sc], 1).astype(np.float32)
    p = pred.copy()
    rows = np.zeros((B, 300, 6), np.float32); keep = np.zeros((B, 300), np.int64); cnt = np.zeros(B, np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.ys_oracle_nms(vp(p), B, 84, A, C.c_float(0.25), C.c_float(0.45), 300, 0, 30000, 7680, vp(rows), vp(keep), vp(cnt)) == 0
    m = int(cnt.max())
    np.savez_compressed(os.path.join(HERE, "nms_golden.npz"), pred=pred, pred_after=p, rows=rows[:, :m], keep=keep[:, :m],
                        count=cnt, conf=np.float32(0.25), iou=np.float32(0.45))


def model_golden():
    from oracle import yolo_oracle as O
    torch.manual_seed(0)
    B, H, W, nc = 2, 64, 64, 80
    ref = O.Yolov8(nc=nc, size="n")
    for mod in ref.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.normal_(0, 0.1)
            mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
    sd0 = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    x = torch.rand(B, 3, H, W)
    batch = O.synthetic_batch(B, H, W, nc, seed=1, kmax=6)
    ref.eval()
    with torch.no_grad():
        inf, _ = ref(x)
    ref.train()
    _, preds = ref(x)
    loss, items = O.v8DetectionLoss(nc)(preds, batch)
    loss.sum().backward()
    # keep the fixture small: weights are regenerated from the seed by the test, only results are stored
    out = {"x": x.numpy(), "batch_idx": batch["batch_idx"].numpy(), "cls": batch["cls"].numpy(), "bboxes": batch["bboxes"].numpy(),
           "pred_eval": inf["boxes"].numpy().astype(np.float16), "boxes_train": preds["boxes"].detach().numpy().astype(np.float16),
           "scores_train": preds["scores"].detach().numpy().astype(np.float16), "loss_items": items.numpy(),
           "grad_model.0.conv.weight": ref.model[0].conv.weight.grad.numpy(),
           "grad_model.22.cv3.0.2.bias": ref.model[22].cv3[0][2].bias.grad.numpy(),
           "grad_model.4.m.1.cv2.bn.weight": ref.model[4].m[1].cv2.bn.weight.grad.numpy(),
           "w_model.0.conv.weight": sd0["model.0.conv.weight"].numpy()}
    np.savez_compressed(os.path.join(HERE, "model_golden.npz"), **out)


if __name__ == "__main__":
    nms_golden()
    model_golden()
    print("fixtures written to", HERE)
