"""Generates tests/golden/kat_tasks.json: hand-built v8PoseLoss and v8OBBLoss cases evaluated by tests/kat_ref.py (scalar fp64
Python, finite-difference gradients, no torch / oracle import).  Run from the repo root:
    python tests/golden/make_kat_tasks.py
Pose: the detection case of kat_loss.json (32x32, 21 anchors, nc = 3, B = 2) + 17 x 3 COCO keypoints (OKS sigmas), with unlabelled
points (visibility 0) and one object whose keypoints are all unlabelled (kpt_loss_factor = 17 / 1e-6, mask all zero).
OBB: 32x32, nc = 2, B = 2: nested oriented boxes, a label thinner than 2 px (filtered, Loss.cs:563), a label 5 px wide (kept; width
becomes 16 inside the assigner and in every later use, Tal.cs:283-287), an image with fewer labels than the batch maximum.
"""
import json
import math
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import kat_ref as K  # noqa: E402

rng = random.Random(20260928)
H = W = 32
REG, B = 16, 2
anchors = K.make_anchors(H, W)
A = len(anchors)
out = {"H": H, "W": W, "B": B}

# ------------------------------------------------------------------ pose
det = json.load(open(os.path.join(HERE, "kat_loss.json")))
KP, D, nc = 17, 3, det["nc"]
keypoints = []
for i, bb in enumerate(det["bboxes"]):
    rows = []
    for k in range(KP):
        x = min(max(bb[0] + rng.uniform(-0.6, 0.6) * bb[2], 0.0), 1.0)
        y = min(max(bb[1] + rng.uniform(-0.6, 0.6) * bb[3], 0.0), 1.0)
        v = 0.0 if (i == 1 or k % 4 == 3) else float(rng.choice((1, 2)))      # label 1: every keypoint unlabelled
        rows.append([x, y, v])
    keypoints.append(rows)
kpts = [[[rng.gauss(0.0, 0.6) for _ in range(A)] for _ in range(KP * D)] for _ in range(B)]
items, total, tg = K.pose_loss(det["boxes"], det["scores"], kpts, det["batch_idx"], det["cls"], det["bboxes"], keypoints, H, W, nc, KP, D)
assert [[bool(v) for v in t[0]] for t in tg] == det["fg"]
used = {(b, tg[b][1][a]) for b in range(B) for a in range(A) if tg[b][0][a]}
assert (0, 1) in used and (0, 0) in used, used                                  # the all-unlabelled object owns foreground anchors too
dk = K.fd_grad(kpts, lambda: sum(K.pose_terms(kpts, tg, det["batch_idx"], det["bboxes"], keypoints, H, W, KP, D)) * B)
print("pose items", items, "max |dkpts|", max(abs(v) for b in dk for c in b for v in c))
out["pose"] = {"nc": nc, "K": KP, "D": D, "keypoints": keypoints, "kpts": kpts, "items": items, "total": total, "dkpts": dk}

# ------------------------------------------------------------------ obb
nc = 2
batch_idx = [0.0, 0.0, 0.0, 0.0, 1.0]
cls = [1.0, 0.0, 1.0, 0.0, 0.0]
bboxes = [[0.5, 0.5, 0.8, 0.55, 0.35],          # image 0: 25.6 x 17.6 px, rotated 20 deg
          [0.55, 0.5, 0.35, 0.3, -0.4],         # image 0: nested, other orientation
          [0.3, 0.3, 0.5, 0.05, 0.2],           # image 0: 1.6 px high -> filtered before padding
          [0.6, 0.45, 0.16, 0.6, 1.1],          # image 0: 5.1 px wide -> kept, width 16 from the assigner on
          [0.45, 0.55, 0.6, 0.6, 2.0]]          # image 1
kept = [i for i, bb in enumerate(bboxes) if bb[2] * W >= 2 and bb[3] * H >= 2]
assert kept == [0, 1, 3, 4]
per = [[(cls[i], bboxes[i][0] * W, bboxes[i][1] * H, bboxes[i][2] * W, bboxes[i][3] * H, bboxes[i][4]) for i in kept if int(batch_idx[i]) == b]
       for b in range(B)]


def peaked(d, sharp):
    return [-sharp * (j - d) ** 2 + rng.uniform(-0.3, 0.3) for j in range(REG)]


boxes = [[[0.0] * A for _ in range(4 * REG)] for _ in range(B)]
scores = [[[rng.gauss(-1.0, 1.0) for _ in range(A)] for _ in range(nc)] for _ in range(B)]
angle = [[[0.0] * A] for _ in range(B)]
for b in range(B):
    for a, (ax, ay, s) in enumerate(anchors):
        g = per[b][(a + b) % len(per[b])]
        tx, ty, tw, th, tr = g[1] / s, g[2] / s, g[3] / s, g[4] / s, g[5]
        ox, oy = tx - ax, ty - ay
        xf, yf = ox * math.cos(tr) + oy * math.sin(tr), -ox * math.sin(tr) + oy * math.cos(tr)
        tgt = (tw / 2 - xf, th / 2 - yf, tw / 2 + xf, th / 2 + yf)
        pr = min(max((tr + rng.uniform(-0.25, 0.25)) / math.pi + 0.25, 0.03), 0.97)     # angle = (sigmoid(z) - 0.25) * pi
        angle[b][0][a] = math.log(pr / (1 - pr))
        for k in range(4):
            d = min(max(tgt[k] + rng.uniform(-0.4, 0.4), 0.2), 14.5)
            row = peaked(d, 1.5)
            for j in range(REG):
                boxes[b][k * REG + j][a] = row[j]
items, total, tg = K.obb_loss(boxes, scores, angle, batch_idx, cls, bboxes, H, W, nc)
fg_total = sum(sum(t[0]) for t in tg)
widened = [tg[0][3][a] for a in range(A) if tg[0][0][a] and tg[0][1][a] == 2]
assert fg_total >= 8 and widened and all(abs(t[2] - 16.0) < 1e-12 for t in widened), (fg_total, widened)
assert all(v > 0 for v in items), items


def frozen():
    return K.obb_loss(boxes, scores, angle, batch_idx, cls, bboxes, H, W, nc, targets=tg)[1]


db, ds, da = K.fd_grad(boxes, frozen), K.fd_grad(scores, frozen), K.fd_grad(angle, frozen)
print("obb items", items, "fg anchors", fg_total, "anchors of the widened label", len(widened))
out["obb"] = {"nc": nc, "batch_idx": batch_idx, "cls": cls, "bboxes": bboxes, "boxes": boxes, "scores": scores, "angle_logit": angle,
              "items": items, "total": total, "fg": [[bool(v) for v in t[0]] for t in tg], "gt_idx": [t[1] for t in tg],
              "dboxes": db, "dscores": ds, "dangle": da}
with open(os.path.join(HERE, "kat_tasks.json"), "w") as f:
    json.dump(out, f)
print("wrote kat_tasks.json", os.path.getsize(os.path.join(HERE, "kat_tasks.json")), "bytes")

# ------------------------------------------------------------------ segment (mask term on the detection case of kat_loss.json)
nm, mh, mw = 32, H // 4, W // 4      # Head.Segment: nm = 32 prototypes at H/4 x W/4 (Head.cs:238-252)
nc = det["nc"]
_, _, dtg = K.detection_loss(det["boxes"], det["scores"], det["batch_idx"], det["cls"], det["bboxes"], H, W, nc)
coeff = [[[rng.gauss(0.0, 1.0) for _ in range(A)] for _ in range(nm)] for _ in range(B)]
proto = [[[[rng.gauss(0.0, 0.8) for _ in range(mw)] for _ in range(mh)] for _ in range(nm)] for _ in range(B)]
masks = [[[0.0] * mw for _ in range(mh)] for _ in range(B)]
slots = [[i for i, bi in enumerate(det["batch_idx"]) if int(bi) == b] for b in range(B)]
for b in range(B):                                   # overlap-encoded ids: later labels painted on top (YoloDataset.cs:265-267)
    for slot, i in enumerate(slots[b]):
        bb = det["bboxes"][i]
        for r in range(mh):
            for c in range(mw):
                u, v = (c + 0.5) / mw, (r + 0.5) / mh
                if ((u - bb[0]) / (bb[2] / 2)) ** 2 + ((v - bb[1]) / (bb[3] / 2)) ** 2 <= 1.0:
                    masks[b][r][c] = float(slot + 1)
seg = K.seg_term(coeff, proto, dtg, masks, H, W)
seg_trunc = K.seg_term(coeff, proto, dtg, masks, H, W, trunc_crop=True)
assert seg > 0 and abs(seg - seg_trunc) > 1e-6, (seg, seg_trunc)      # the two crop branches differ on this case (fractional box edges)
dc = K.fd_grad(coeff, lambda: K.seg_term(coeff, proto, dtg, masks, H, W) * B)
dp = [K.fd_grad(proto[b], lambda: K.seg_term(coeff, proto, dtg, masks, H, W) * B) for b in range(B)]
print("seg item", seg, "cpu-crop branch", seg_trunc)
out["segment"] = {"nm": nm, "coeff": coeff, "proto": proto, "masks": masks, "item": seg, "item_trunc": seg_trunc, "dcoeff": dc, "dproto": dp}
with open(os.path.join(HERE, "kat_tasks.json"), "w") as f:
    json.dump(out, f)
print("wrote kat_tasks.json", os.path.getsize(os.path.join(HERE, "kat_tasks.json")), "bytes")
