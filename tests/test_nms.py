"""NMS parity: HIP kernels (Ops.cs:239-371 + torchvision.ops.nms) vs the C oracle -- bit-exact rows and indices."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import BACKENDS

GOLD = os.path.join(os.path.dirname(__file__), "golden", "nms_golden.npz")


def oracle_nms(lib, pred, conf, iou, max_det=300, nc=0, max_nms=30000, max_wh=7680):
    B, Cc, A = pred.shape
    extra = Cc - 4 - (nc or Cc - 4)
    p = pred.copy()
    rows = np.zeros((B, max_det, 6 + extra), np.float32)
    keep = np.zeros((B, max_det), np.int64)
    cnt = np.zeros(B, np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    st = lib.ys_oracle_nms(vp(p), B, Cc, A, C.c_float(conf), C.c_float(iou), max_det, nc, max_nms, max_wh, vp(rows), vp(keep), vp(cnt))
    assert st == 0
    return p, [rows[b, :cnt[b]] for b in range(B)], [keep[b, :cnt[b]] for b in range(B)]


def synth(rng, B, A, nc, extra=0, ties=False, img=640.0):
    wh = rng.uniform(0.03, 0.6, (B, 2, A)) * img
    c = rng.uniform(0, img, (B, 2, A))
    sc = 1 / (1 + np.exp(-rng.normal(-3, 1.5, (B, nc, A))))
    if ties:
        sc = np.round(sc * 16) / 16
    ex = rng.normal(size=(B, extra, A))
    return np.concatenate([c, wh, sc, ex], 1).astype(np.float32)


def check(engine, oracle_lib, pred, conf, iou, **kw):
    ref_p, ref_rows, ref_keep = oracle_nms(oracle_lib, pred, conf, iou, **kw)
    mine = pred.copy()
    out, keepi = engine.non_max_suppression(mine, conf, iou, **kw)
    assert np.array_equal(mine, ref_p), "in-place xywh->xyxy mismatch"
    for b in range(pred.shape[0]):
        assert np.array_equal(keepi[b], ref_keep[b]), f"kept indices differ in image {b}"
        assert np.array_equal(out[b], ref_rows[b]), f"rows differ in image {b}"
    return out, keepi


CASES = [
    dict(B=2, A=400, nc=80, extra=0, ties=False, conf=0.25, iou=0.45, kw={}),
    dict(B=3, A=700, nc=5, extra=3, ties=True, conf=0.1, iou=0.5, kw=dict(nc=5)),       # ties + mask coefficients
    dict(B=2, A=300, nc=80, extra=0, ties=False, conf=0.0, iou=0.7, kw=dict(max_det=20)),  # n > max_det
    dict(B=1, A=200, nc=3, extra=0, ties=False, conf=0.99, iou=0.5, kw={}),              # n == 0
    dict(B=1, A=600, nc=1, extra=0, ties=True, conf=0.05, iou=0.3, kw=dict(max_nms=100)),  # n > max_nms, one class
    dict(B=2, A=257, nc=80, extra=0, ties=False, conf=0.3, iou=0.7, kw={}),              # predictor thresholds
    dict(B=2, A=4300, nc=2, extra=1, ties=True, conf=0.0, iou=0.6, kw=dict(nc=2, max_det=12)),   # n > 4096: serial sweep path next to the bitmask path
    dict(B=1, A=1030, nc=3, extra=0, ties=True, conf=0.02, iou=0.45, kw={}),             # A % 4 != 0: scalar filter; several 64-row mask blocks
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", range(len(CASES)))
def test_nms_matches_oracle(backend, engine, oracle_lib, case):
    c = CASES[case]
    rng = np.random.default_rng(100 + case)
    pred = synth(rng, c["B"], c["A"], c["nc"], c["extra"], c["ties"])
    out, keepi = check(engine, oracle_lib, pred, c["conf"], c["iou"], **c["kw"])
    if case == 3:
        assert all(len(k) == 0 for k in keepi)
    if case == 2:
        assert all(len(k) == 20 for k in keepi)


@pytest.mark.parametrize("backend", BACKENDS)
def test_nms_golden_fixture(backend, engine):
    """Committed vectors (tests/golden/make_golden.py); expected outputs come from the C oracle."""
    g = np.load(GOLD)
    pred = g["pred"].copy()
    out, keepi = engine.non_max_suppression(pred, float(g["conf"]), float(g["iou"]))
    for b in range(pred.shape[0]):
        n = int(g["count"][b])
        assert np.array_equal(keepi[b], g["keep"][b, :n])
        assert np.array_equal(out[b], g["rows"][b, :n])
    assert np.array_equal(pred, g["pred_after"])


def test_oracle_matches_golden(oracle_lib):
    g = np.load(GOLD)
    p, rows, keep = oracle_nms(oracle_lib, g["pred"], float(g["conf"]), float(g["iou"]))
    for b in range(p.shape[0]):
        n = int(g["count"][b])
        assert np.array_equal(keep[b], g["keep"][b, :n]) and np.array_equal(rows[b], g["rows"][b, :n])


def test_oracle_handcrafted_known_answer(oracle_lib):
    """Hand-computed case: two overlapping boxes of one class (IoU 0.8 > 0.5 -> lower score dropped), the same pair in
    different classes (kept: class offset), and an equal-score tie (lower anchor index first)."""
    A = 6
    pred = np.zeros((1, 6, A), np.float32)   # nc = 2
    boxes = [(50, 50, 20, 20), (52, 50, 20, 20), (50, 50, 20, 20), (200, 200, 10, 10), (200, 200, 10, 10), (400, 400, 8, 8)]
    scores = [(0.9, 0), (0.8, 0), (0, 0.7), (0.6, 0), (0.6, 0), (0.2, 0.1)]
    for a, (b, s) in enumerate(zip(boxes, scores)):
        pred[0, :4, a] = b
        pred[0, 4:, a] = s
    _, rows, keep = oracle_nms(oracle_lib, pred, 0.25, 0.5)
    assert keep[0].tolist() == [0, 2, 3]
    assert rows[0][:, 5].tolist() == [0.0, 1.0, 0.0]
    assert np.allclose(rows[0][0, :4], [40, 40, 60, 60])


@pytest.mark.parametrize("backend", BACKENDS)
def test_nms_invalid_thresholds_raise(backend, engine):
    """Ops.cs:248-255 throws ArgumentException."""
    from yolosharp_amd import YsError
    pred = np.zeros((1, 6, 8), np.float32)
    for conf, iou in [(-0.1, 0.5), (1.5, 0.5), (0.3, -0.2), (0.3, 1.01)]:
        with pytest.raises(YsError) as e:
            engine.non_max_suppression(pred.copy(), conf, iou)
        assert e.value.status == 1


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gpu"])
def test_nms_full_size_matches_oracle(backend, engine, oracle_lib):
    """BASELINE shape [64, 84, 8400], plus the all-pass worst case on a smaller batch."""
    rng = np.random.default_rng(3)
    pred = synth(rng, 64, 8400, 80)
    check(engine, oracle_lib, pred, 0.25, 0.45)
    pred = synth(rng, 2, 8400, 80)
    pred[:, 4:] = np.maximum(pred[:, 4:], 0.3)
    out, keepi = check(engine, oracle_lib, pred, 0.25, 0.7)
    assert all(len(k) == 300 for k in keepi)
    # idempotence property: NMS of the kept boxes (already xyxy -> feed back as xywh) keeps all of them
    for b in range(2):
        r = out[b]
        p2 = np.zeros((1, 84, len(r)), np.float32)
        p2[0, 0] = (r[:, 0] + r[:, 2]) / 2; p2[0, 1] = (r[:, 1] + r[:, 3]) / 2
        p2[0, 2] = r[:, 2] - r[:, 0]; p2[0, 3] = r[:, 3] - r[:, 1]
        p2[0, 4 + r[:, 5].astype(int), np.arange(len(r))] = r[:, 4]
        o2, k2 = engine.non_max_suppression(p2, 0.25, 0.7)
        assert len(k2[0]) >= len(r) - 2   # re-derived xywh can move an IoU across the threshold by 1 ulp
