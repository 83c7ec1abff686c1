"""Validation matching on the device: Metrics.box_iou (Metrics.cs:16-34) and match_predictions
(YoloBaseTaskModel.cs:377-446) per image, batched.  Integer / boolean outputs are compared exactly with the oracle."""
import numpy as np
import pytest
import torch

from conftest import BACKENDS
from oracle import yolo_oracle as O


def _scene(seed, B, max_det, nc, W=640.0, H=640.0, kmax=12, extra=0):
    """Labels + NMS-shaped detections: jittered copies of GT boxes (several per label, some with the wrong class) and
    random clutter, sorted by confidence like NMS output."""
    g = torch.Generator().manual_seed(seed)
    batch = O.synthetic_batch(B, int(H), int(W), nc, seed=seed, kmax=kmax)
    rows = torch.zeros(B, max_det, 6 + extra)
    count = torch.zeros(B, dtype=torch.int32)
    for b in range(B):
        sel = batch["batch_idx"] == b
        gt = O.xywh2xyxy(batch["bboxes"][sel] * torch.tensor([W, H, W, H]))
        cls = batch["cls"][sel]
        dets = []
        for j in range(gt.shape[0]):
            for _ in range(int(torch.randint(0, 4, (1,), generator=g))):
                jit = (torch.rand(4, generator=g) - 0.5) * 0.35 * (gt[j, 2:] - gt[j, :2]).repeat(2)
                c = cls[j] if torch.rand(1, generator=g) < 0.8 else torch.randint(0, nc, (1,), generator=g).float()[0]
                dets.append(torch.cat((gt[j] + jit, torch.rand(1, generator=g), c.view(1))))
        for _ in range(int(torch.randint(0, 6, (1,), generator=g))):
            xy = torch.rand(2, generator=g) * torch.tensor([W, H]) * 0.8
            dets.append(torch.cat((xy, xy + torch.rand(2, generator=g) * 100 + 5, torch.rand(1, generator=g),
                                   torch.randint(0, nc, (1,), generator=g).float())))
        if b == B - 1:
            dets = []                                               # an image without detections
        if dets:
            d = torch.stack(dets)[:max_det]
            d = d[d[:, 4].argsort(descending=True)]
            rows[b, :d.shape[0], :6] = d
            count[b] = d.shape[0]
    return batch, rows, count


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_val_match_batched(backend, engine, seed):
    B, max_det, nc = 5, 48, 3
    batch, rows, count = _scene(seed, B, max_det, nc, extra=2 * (seed % 2))
    if seed == 2:                                                   # an image without labels
        keep = batch["batch_idx"] != 1
        batch = {k: v[keep] for k, v in batch.items()}
    got = engine.val_match(rows.numpy(), count.numpy(), {k: v.numpy() for k, v in batch.items()}, 640, 640)
    total = 0
    for b in range(B):
        ref = O.val_match_image(rows[b, :count[b]], batch, b, 640.0, 640.0).numpy()
        assert got[b].shape == ref.shape and np.array_equal(got[b], ref), (b, got[b].astype(int), ref.astype(int))
        total += int(ref.sum())
    assert total > 0


@pytest.mark.parametrize("backend", BACKENDS)
def test_match_quirks(backend, engine):
    """Two detections on one label, one detection between two labels, a wrong-class detection, and the reference's
    de-duplication order (per detection first, then per label by the LOWEST detection index, not the best IoU)."""
    W = H = 100.0
    batch = {"batch_idx": torch.tensor([0.0, 0.0]), "cls": torch.tensor([1.0, 1.0]),
             "bboxes": torch.tensor([[0.25, 0.25, 0.2, 0.2], [0.65, 0.25, 0.2, 0.2]])}      # gt0 = [15,15,35,35], gt1 = [55,15,75,35]
    rows = torch.tensor([[[16.0, 15, 36, 35, 0.9, 1],      # d0: gt0, IoU ~0.905
                          [15.0, 15, 35, 35, 0.8, 1],      # d1: gt0, IoU 1.0 (better, but the label is credited to d0 where d0 passes)
                          [55.0, 15, 75, 35, 0.7, 2],      # d2: gt1 geometry, wrong class
                          [57.0, 15, 77, 35, 0.6, 1],      # d3: gt1, IoU ~0.818
                          [0.0, 0, 5, 5, 0.5, 1]]])        # d4: nothing
    count = torch.tensor([5], dtype=torch.int32)
    ref = O.val_match_image(rows[0], batch, 0, W, H).numpy()
    got = engine.val_match(rows.numpy(), count.numpy(), {k: v.numpy() for k, v in batch.items()}, W, H)[0]
    assert np.array_equal(got, ref)
    assert got[0, 0] and not got[1, 0]                      # threshold 0.5: d0 takes gt0 (lowest index), d1 is a duplicate
    assert got[1, 9] and not got[0, 9]                      # threshold 0.95: only d1 clears it
    assert not got[2].any() and got[3, :7].all() and not got[3, 7:].any() and not got[4].any()


@pytest.mark.parametrize("backend", BACKENDS)
def test_box_iou(backend, engine):
    g = torch.Generator().manual_seed(3)
    a = torch.rand(37, 2, generator=g) * 500; b = torch.rand(53, 2, generator=g) * 500
    b1 = torch.cat((a, a + torch.rand(37, 2, generator=g) * 200), 1)
    b2 = torch.cat((b, b + torch.rand(53, 2, generator=g) * 200), 1)
    b2[0] = b1[0]                                            # identical boxes; and a degenerate one
    b2[1, 2:] = b2[1, :2]
    ref = O.box_iou(b1, b2).numpy()
    got = engine.box_iou(b1.numpy(), b2.numpy())
    assert got.shape == ref.shape and np.array_equal(got, ref)          # same fp32 operation order -> bit-exact
    assert engine.box_iou(np.zeros((0, 4), np.float32), b2.numpy()).shape == (0, 53)


def test_thresholds_equal_torch_linspace():
    """The kernel's threshold table restates ATen's fp32 linspace; pin the restatement."""
    start, end, n = np.float32(0.5), np.float32(0.95), 10
    step = np.float32((end - start) / np.float32(n - 1))
    mine = np.array([start + step * np.float32(i) if i < n // 2 else end - step * np.float32(n - 1 - i) for i in range(n)], np.float32)
    assert np.array_equal(mine, torch.linspace(0.5, 0.95, 10, dtype=torch.float32).numpy())


@pytest.mark.parametrize("seed", [0, 1])
def test_ap_per_class_host(seed):
    """Host numpy mirror of Metrics.ap_per_class (Metrics.cs:308-486) against the torch restatement."""
    from yolosharp_amd import metrics as M
    g = torch.Generator().manual_seed(10 + seed)
    n, m, nc = 400, 120, 5
    conf = torch.rand(n, generator=g)
    pred_cls = torch.randint(0, nc, (n,), generator=g).float()
    target_cls = torch.randint(0, nc - seed, (m,), generator=g).float()            # seed 1: a predicted class without labels
    q = torch.rand(n, generator=g) * (0.4 + conf)                                   # confident detections are right more often
    tp = (q[:, None] > torch.linspace(0.3, 0.9, 10)[None, :])
    p, r, f1, ap, uc, tpn, fpn = O.ap_per_class(tp, conf, pred_cls, target_cls)
    got = M.ap_per_class(tp.numpy(), conf.numpy(), pred_cls.numpy(), target_cls.numpy())
    assert np.array_equal(got["unique_classes"], uc.numpy())
    for k, ref in (("p", p), ("r", r), ("f1", f1), ("ap", ap)):
        assert np.allclose(got[k], ref.numpy(), rtol=1e-5, atol=1e-6), k
    assert np.array_equal(got["tp"], tpn.numpy()) and np.array_equal(got["fp"], fpn.numpy())
    P, R, m50, m5095 = M.val_summary(got)
    assert 0 < m5095 <= m50 and 0 < P <= 1 and R > 0        # (synthetic tp flags are not capped by the label count)
    # perfect detector: every label found once with a correct box.  The C# interp() returns `left` = 0 at x <= xp[0] and the
    # sentinel 0 at x >= xp[-1], so the 101-point trapezoid loses both end intervals: AP = 0.99, not 1 (reference quirk)
    tc = torch.arange(6).float()
    perfect = M.ap_per_class(np.ones((6, 10), bool), np.linspace(0.9, 0.4, 6, dtype=np.float32), tc.numpy(), tc.numpy())
    assert np.allclose(perfect["ap"], 0.99, atol=1e-6) and np.allclose(perfect["r"], 1.0, atol=1e-6)


def _mask_scene(seed, nl, n, mh, mw):
    """An overlap-encoded id map (YoloDataset.cs:265-267) with nl instances and n predicted 0/1 masks: jittered copies of
    the instances (so IoUs span 0..1), an exact copy (IoU 1), an empty mask and a full mask."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(mh, mw)
    boxes = []
    for k in range(nl):
        x0, y0 = int(torch.randint(0, mw - 8, (1,), generator=g)), int(torch.randint(0, mh - 8, (1,), generator=g))
        w, h = int(torch.randint(4, mw // 2, (1,), generator=g)), int(torch.randint(4, mh // 2, (1,), generator=g))
        ids[y0:y0 + h, x0:x0 + w] = k + 1          # later instances overwrite earlier ones (overlap encoding)
        boxes.append((x0, y0, w, h))
    pm = torch.zeros(n, mh, mw)
    for j in range(n):
        if j == 0 and nl:
            pm[0] = (ids == 1).float()
        elif j == 1:
            pass                                   # empty prediction: union may be 0 -> 0 / eps
        elif j == 2:
            pm[2] = 1.0
        elif nl:
            x0, y0, w, h = boxes[j % nl]
            dx, dy = int(torch.randint(-3, 4, (1,), generator=g)), int(torch.randint(-3, 4, (1,), generator=g))
            pm[j, max(0, y0 + dy):y0 + dy + h, max(0, x0 + dx):x0 + dx + w] = 1.0
        else:
            pm[j] = (torch.rand(mh, mw, generator=g) > 0.5).float()
    return ids, pm


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("nl,n,mh,mw", [(5, 12, 40, 48), (1, 3, 16, 16), (0, 4, 16, 16), (7, 0, 16, 16), (300, 5, 32, 32)])
def test_mask_iou_exact(backend, engine, nl, n, mh, mw):
    """Metrics.mask_iou (Metrics.cs:120-125) on Segmenter.Val's operands (Segmenter.cs:131-143): bit-exact vs the oracle matmul."""
    ids, pm = _mask_scene(nl + n, nl, n, mh, mw)
    got = engine.mask_iou(ids.numpy(), nl, pm.numpy())
    index = torch.arange(1, nl + 1).view(nl, 1, 1)
    gt = (ids[None] == index).float()
    ref = O.mask_iou(gt.flatten(1), pm.flatten(1)).numpy() if nl and n else np.zeros((nl, n), np.float32)
    assert got.shape == (nl, n) and got.dtype == np.float32
    assert np.array_equal(got, ref)
    if nl and n:
        assert np.all(got[:, 1] == 0)
        if gt[0].sum() > 0:                      # (instance 1 may be fully overwritten by later ones)
            assert got[0, 0] == pytest.approx(1.0, abs=1e-6)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_match_predictions_on_mask_iou(backend, engine, seed):
    """Segmenter.cs:142-143: tp_m = match_predictions(pred_classes, true_classes, mask_iou(...)) -- exact vs the oracle,
    including images without labels / without detections."""
    g = torch.Generator().manual_seed(100 + seed)
    nl, n = [(6, 20), (3, 7), (0, 5), (4, 0)][seed]
    ids, pm = _mask_scene(seed, nl, n, 40, 40)
    tcls = torch.randint(0, 3, (nl,), generator=g).float()
    pcls = torch.tensor([float(tcls[j % nl]) if nl and j % 4 else float(torch.randint(0, 3, (1,), generator=g)) for j in range(n)])
    miou = engine.mask_iou(ids.numpy(), nl, pm.numpy())
    got = engine.match_predictions(pcls.numpy(), tcls.numpy(), miou)
    ref = O.match_predictions(pcls, tcls, torch.from_numpy(miou)).numpy()
    assert got.shape == (n, 10) and np.array_equal(got, ref.astype(bool)), (got.astype(int), ref.astype(int))
    if seed == 0:
        assert got.sum() > 0
