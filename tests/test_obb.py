"""Oriented-box operators of the predict / validation path: Metrics.probiou / batch_probiou (Utils/Metrics.cs:137-177, 223-283) and
Ops.non_max_suppression(rotated: true) = Ops.nms_rotated (Utils/Ops.cs:286, 349-353, 373-401), against the ATen restatement in the
oracle and the independent scalar fp64 arithmetic of tests/kat_ref.py."""
import numpy as np
import pytest
import torch

import kat_ref
from conftest import BACKENDS
from oracle import yolo_oracle as O


def rand_obb(rng, n, span=200.0):
    return np.stack([rng.uniform(0, span, n), rng.uniform(0, span, n), rng.uniform(4, 80, n), rng.uniform(4, 80, n),
                     rng.uniform(-np.pi / 2, np.pi / 2, n)], 1).astype(np.float32)


@pytest.mark.parametrize("backend", BACKENDS)
def test_probiou_matches_oracle_and_scalar_reference(backend, engine):
    rng = np.random.default_rng(0)
    o1, o2 = rand_obb(rng, 64, 100.0), rand_obb(rng, 64, 100.0)
    o2[:8] = o1[:8]                                   # identical boxes: similarity 1 - sqrt(eps)-ish
    o2[8:16, :2] = o1[8:16, :2] + 1.5                 # heavy overlap
    got = engine.probiou(o1, o2)
    ref = O.probiou(torch.from_numpy(o1), torch.from_numpy(o2)).numpy()
    assert np.abs(got - ref).max() <= 2e-6
    kat = np.array([kat_ref.probiou(a.astype(np.float64), b.astype(np.float64)) for a, b in zip(o1, o2)])
    assert np.abs(got - kat).max() <= 2e-4            # fp32 vs fp64 through log / exp / sqrt near iou ~ 1
    assert got[:8].min() > 0.99 and got[16:].max() < 0.9
    # CIoU option: the intended per-pair value (the diagonal of the reference's [N] x [N, 1] broadcast, Metrics.cs:166-173)
    gotc = engine.probiou(o1, o2, CIoU=True)
    refc = O.probiou(torch.from_numpy(o1), torch.from_numpy(o2), CIoU=True).numpy()
    assert np.abs(gotc - refc).max() <= 5e-6
    # all pairs
    gb = engine.batch_probiou(o1[:20], o2[:33])
    rb = O.batch_probiou(torch.from_numpy(o1[:20]), torch.from_numpy(o2[:33])).numpy()
    assert gb.shape == (20, 33) and np.abs(gb - rb).max() <= 2e-6
    assert np.abs(gb[3, 5] - kat_ref.probiou(o1[3].astype(np.float64), o2[5].astype(np.float64))) <= 2e-4


def make_pred(rng, B, nc, extra_mid, A, n_hot):
    """[B, 4 + nc + extra_mid + 1, A]: xywh, class probabilities (n_hot anchors above any threshold), optional extra channels, angle."""
    C = 4 + nc + extra_mid + 1
    p = np.zeros((B, C, A), np.float32)
    p[:, 0] = rng.uniform(20, 300, (B, A)); p[:, 1] = rng.uniform(20, 300, (B, A))
    p[:, 2] = rng.uniform(10, 90, (B, A)); p[:, 3] = rng.uniform(10, 90, (B, A))
    p[:, 4:4 + nc] = rng.uniform(0.0, 0.2, (B, nc, A))
    for b in range(B):
        hot = rng.choice(A, n_hot[b], replace=False)
        p[b, 4 + rng.integers(0, nc, len(hot)), hot] = rng.uniform(0.3, 0.99, len(hot)).astype(np.float32)
        # clusters: every third hot anchor is a near copy of the previous one (overlap >= threshold)
        for k in range(2, len(hot), 3):
            p[b, :4, hot[k]] = p[b, :4, hot[k - 1]] + rng.uniform(-1.0, 1.0, 4).astype(np.float32)
            p[b, -1, hot[k]] = p[b, -1, hot[k - 1]]
            cl = int(np.argmax(p[b, 4:4 + nc, hot[k - 1]]))
            p[b, 4:4 + nc, hot[k]] = 0.0; p[b, 4 + cl, hot[k]] = rng.uniform(0.3, 0.99)
    if extra_mid:
        p[:, 4 + nc:4 + nc + extra_mid] = rng.standard_normal((B, extra_mid, A)).astype(np.float32)
    p[:, -1] = np.where(p[:, -1] == 0, rng.uniform(-np.pi / 4, np.pi / 4, (B, A)), p[:, -1]).astype(np.float32)
    return p


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [(2, 3, 0, 512, (40, 0)), (3, 5, 2, 1100, (300, 17, 1)), (1, 2, 0, 777, (500,))])
def test_rotated_nms_matches_oracle(backend, engine, case):
    B, nc, extra_mid, A, n_hot = case
    rng = np.random.default_rng(B * 100 + A)
    pred = make_pred(rng, B, nc, extra_mid, A, n_hot)
    ref_out, ref_keep = O.non_max_suppression_rotated(torch.from_numpy(pred.copy()), 0.25, 0.45, max_det=300, nc=nc)
    before = pred.copy()
    out, keep = engine.non_max_suppression(pred, 0.25, 0.45, nc=nc, rotated=True)
    assert np.array_equal(pred, before)                                   # rotated: no in-place xyxy conversion (Ops.cs:286)
    for b in range(B):
        assert np.array_equal(keep[b], ref_keep[b].numpy()), b            # same kept anchors, same order
        assert np.array_equal(out[b], ref_out[b].numpy().astype(np.float32)), b
    assert len(keep[0]) > 0 and len(keep[0]) < n_hot[0]                   # something was suppressed


@pytest.mark.parametrize("backend", BACKENDS)
def test_rotated_nms_is_not_greedy_known_answer(backend, engine):
    """Hand-built chain A (0.9) - B (0.8) - C (0.7): B overlaps A, C overlaps B but not A.  torchvision's greedy rule keeps A and C
    (B is gone when C is examined); Ops.nms_rotated drops every box that ANY higher-scored box overlaps (Ops.cs:386-388): only A.
    D (0.6) is far away and stays.  probiou(A, B) etc. are checked against the scalar reference first."""
    boxes = np.array([[100, 100, 40, 20, 0.3], [108, 102, 40, 20, 0.3], [116, 104, 40, 20, 0.3], [300, 300, 40, 20, 0.0]], np.float64)
    ab, bc, ac = kat_ref.probiou(boxes[0], boxes[1]), kat_ref.probiou(boxes[1], boxes[2]), kat_ref.probiou(boxes[0], boxes[2])
    assert ab > 0.6 and bc > 0.6 and ac < 0.6 and kat_ref.probiou(boxes[0], boxes[3]) < 0.01      # 0.750, 0.750, 0.524
    A = 8
    pred = np.zeros((1, 4 + 1 + 1, A), np.float32)
    for k, (a, sc) in enumerate(zip((5, 2, 7, 0), (0.9, 0.8, 0.7, 0.6))):
        pred[0, :4, a] = boxes[k, :4]; pred[0, 4, a] = sc; pred[0, 5, a] = boxes[k, 4]
    out, keep = engine.non_max_suppression(pred, 0.25, 0.6, nc=1, rotated=True)
    assert keep[0].tolist() == [5, 0]
    assert np.allclose(out[0][:, 4], [0.9, 0.6]) and np.allclose(out[0][0, :4], boxes[0, :4]) and np.allclose(out[0][:, 6], [0.3, 0.0])   # rows: xywh, conf, cls, angle
    ref_out, ref_keep = O.non_max_suppression_rotated(torch.from_numpy(pred.copy()), 0.25, 0.6, nc=1)
    assert ref_keep[0].tolist() == [5, 0]
