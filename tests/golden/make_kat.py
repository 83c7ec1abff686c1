"""Generates tests/golden/kat_loss.json: a hand-built v8DetectionLoss case evaluated by tests/kat_ref.py (scalar fp64 Python,
no torch / numpy arithmetic, no oracle import, finite-difference gradients).  Run from the repo root:
    python tests/golden/make_kat.py
The case (32x32 image -> 16 + 4 + 1 = 21 anchors, nc = 3, B = 2) is constructed to contain, and the script asserts it contains:
  * an anchor claimed by two ground truths (resolved by arg-max CIoU overlap, Tal.cs:231-241),
  * in-GT anchors whose clamped CIoU is exactly 0 (alignment ties at zero; 'lower index first'),
  * a ground truth narrower than 8 px (inflated to 16 px for the in-GT test only, Tal.cs:206-211),
  * an image with fewer labels than the batch maximum (a zero-padded GT row, mask_gt = 0),
  * DFL targets below the lower clamp (0); the upper clamp (14.99) is pinned by the scalar DFL cases in the same file.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import kat_ref as K  # noqa: E402

H = W = 32
NC, REG, B = 3, 16, 2
rng = random.Random(20260927)
anchors = K.make_anchors(H, W)
A = len(anchors)
batch_idx = [0.0, 0.0, 1.0]
cls = [1.0, 2.0, 0.0]
bboxes = [[0.5, 0.5, 0.875, 0.875],      # image 0: (2,2)-(30,30)
          [0.5625, 0.5625, 0.5, 0.5],    # image 0: (10,10)-(26,26), nested in the first
          [0.40625, 0.5, 0.125, 0.75]]   # image 1: 4 px wide (inflated to 16 for the in-GT test), 24 px tall
per = K._labels_by_image(batch_idx, cls, bboxes, B, H, W)


def peaked(d, sharp):
    """16 logits whose softmax expectation is close to d."""
    return [-sharp * (j - d) ** 2 + rng.uniform(-0.3, 0.3) for j in range(REG)]


boxes = [[[0.0] * A for _ in range(4 * REG)] for _ in range(B)]
scores = [[[rng.gauss(-1.0, 1.0) for _ in range(A)] for _ in range(NC)] for _ in range(B)]
for b in range(B):
    for a, (ax, ay, s) in enumerate(anchors):
        g = per[b][(a + b) % len(per[b])]
        tgt = ((ax * s - g[1]) / s, (ay * s - g[2]) / s, (g[3] - ax * s) / s, (g[4] - ay * s) / s)
        far = (a % 5 == 3)                      # every fifth anchor predicts a far-away sliver -> CIoU <= 0 -> overlap clamps to 0
        for k in range(4):
            d = min(max(tgt[k] + rng.uniform(-0.4, 0.4), 0.2), 14.5)
            if far:
                d = 14.0 if k == 0 else 0.05
            row = peaked(d, 1.5 if not far else 4.0)
            for j in range(REG):
                boxes[b][k * REG + j][a] = row[j]

items, total, tg = K.detection_loss(boxes, scores, batch_idx, cls, bboxes, H, W, NC, REG)

# ---- the properties the case was built for
pbox0 = K.decode([[boxes[0][c][a] for c in range(4 * REG)] for a in range(A)], anchors, REG)
px0 = [tuple(v * anchors[a][2] for v in pbox0[a]) for a in range(A)]
sig0 = [[K.sigmoid(scores[0][c][a]) for c in range(NC)] for a in range(A)]
claims, zero_ties = 0, 0
for a in range(A):
    in_both = all(min(anchors[a][0] * anchors[a][2] - g[1], anchors[a][1] * anchors[a][2] - g[2], g[3] - anchors[a][0] * anchors[a][2],
                      g[4] - anchors[a][1] * anchors[a][2]) > 1e-9 for g in per[0])
    ovs = [max(K.ciou(g[1:5], px0[a]), 0.0) for g in per[0]]
    if in_both and min(ovs) > 0:
        claims += 1
    if ovs[0] == 0.0:
        zero_ties += 1
fg_total = sum(sum(t[0]) for t in tg)
assert claims >= 1 and zero_ties >= 2 and fg_total >= 6, (claims, zero_ties, fg_total)
dfl_clamped_lo = dfl_clamped_hi = 0
for b in range(B):
    for a in range(A):
        if tg[b][0][a]:
            ax, ay, s = anchors[a]
            tb = [v / s for v in tg[b][3][a]]
            for v in (ax - tb[0], ay - tb[1], tb[2] - ax, tb[3] - ay):
                dfl_clamped_lo += v < 0
                dfl_clamped_hi += v > 14.99
print("fg anchors:", fg_total, "double-claim candidates:", claims, "zero-overlap in-GT anchors:", zero_ties,
      "dfl clamps lo/hi:", dfl_clamped_lo, dfl_clamped_hi, "items:", items)

dboxes, dscores = K.detection_loss_grads(boxes, scores, batch_idx, cls, bboxes, H, W, NC, REG)
out = {"H": H, "W": W, "nc": NC, "reg_max": REG, "B": B, "batch_idx": batch_idx, "cls": cls, "bboxes": bboxes,
       "boxes": boxes, "scores": scores, "items": items, "total": total,
       "fg": [[bool(v) for v in t[0]] for t in tg], "gt_idx": [t[1] for t in tg], "target_scores": [t[2] for t in tg],
       "dboxes": dboxes, "dscores": dscores}

# ---- scalar known answers (CIoU on five box pairs, DFL at four targets, BN statistics of a tiny Conv unit)
pairs = [((0, 0, 4, 4), (1, 1, 5, 5)), ((0, 0, 4, 2), (0, 0, 2, 4)), ((0, 0, 2, 2), (5, 5, 7, 9)), ((1, 1, 3, 3), (1, 1, 3, 3)),
         ((0, 0, 3, 0), (0, 0, 3, 1))]            # the last pair has a zero-height box: h clamps to eps
out["ciou_pairs"] = [[list(p[0]), list(p[1]), K.ciou(p[0], p[1])] for p in pairs]
lg = [rng.gauss(0, 1.5) for _ in range(16)]
out["dfl_logits"] = lg
out["dfl_cases"] = [[t, K.dfl(lg, t)] for t in (0.0, 7.3, 14.99, 20.0)]
Cc, N = 4, 2 * 4 * 4
xin = [[rng.gauss(0.3, 1.2) for _ in range(Cc)] for _ in range(N)]                  # [N = B*H*W][Cin]
wk = [[rng.uniform(-0.7, 0.7) for _ in range(Cc)] for _ in range(Cc)]               # [Cout][Cin], 1x1
gam = [rng.uniform(0.5, 1.5) for _ in range(Cc)]
bet = [rng.gauss(0, 0.2) for _ in range(Cc)]
rm = [rng.gauss(0, 0.1) for _ in range(Cc)]
rv = [rng.uniform(0.5, 1.5) for _ in range(Cc)]
y = [[sum(wk[co][ci] * xin[n][ci] for ci in range(Cc)) for co in range(Cc)] for n in range(N)]
z, nm, nv = K.bn_train_stats(y, gam, bet, rm, rv)
act = [[v * K.sigmoid(v) for v in row] for row in z]
out["bn"] = {"x": xin, "w": wk, "gamma": gam, "beta": bet, "running_mean": rm, "running_var": rv,
             "out": act, "new_running_mean": nm, "new_running_var": nv}
json.dump(out, open(os.path.join(HERE, "kat_loss.json"), "w"))
print("wrote", os.path.join(HERE, "kat_loss.json"))
