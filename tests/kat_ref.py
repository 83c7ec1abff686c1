"""tests/kat_ref.py -- INDEPENDENT known-answer arithmetic for the loss path (test infrastructure only).

A second restatement of the cited C# that shares no code, no tensor library and no autograd with oracle/yolo_oracle.py:
scalar fp64 Python (`math` only), explicit loops, and central finite differences for the gradients.  It pins both the ATen
oracle and the HIP kernels (tests/test_kat.py) on small hand-built cases:

  ciou            Utils/Metrics.cs:36-111   (xywh = false, CIoU = true; h clamped to eps, alpha not detached)
  dfl             Utils/Loss.cs:94-120      (clamp(0, reg_max - 1 - 0.01), floor / floor+1 cross-entropies)
  tal_assign      Utils/Tal.cs:50-258       (in-GT test incl. the 8 px -> 16 px inflation, CIoU overlaps, score^0.5 * ov^6,
                                             top-10 with 'selected exactly once', multi-GT resolution, normalised targets)
  detection_loss  Utils/Loss.cs:411-477     (decode, assignment, BCE / CIoU / DFL terms, gains 7.5 / 0.5 / 1.5, tss = max(sum, 1))
  bn_train_stats  Modules/Convs.cs:41-48    (biased batch variance in the normalisation, unbiased in running_var, momentum 0.03)

Ties: the reference's torch.topk order among EQUAL metrics is unspecified; this file and the engine use 'lower anchor index first'
(SURVEY.md Appendix C).  The fixtures contain exact ties only at metric == 0.
"""
import math

EPS_IOU = 1e-7


def sigmoid(x):
    return 1.0 / (1.0 + math.exp(-x))


def ciou(b1, b2, eps=EPS_IOU):
    """Metrics.cs:76-103."""
    w1, h1 = b1[2] - b1[0], max(b1[3] - b1[1], eps)
    w2, h2 = b2[2] - b2[0], max(b2[3] - b2[1], eps)
    inter = max(min(b1[2], b2[2]) - max(b1[0], b2[0]), 0.0) * max(min(b1[3], b2[3]) - max(b1[1], b2[1]), 0.0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = max(b1[2], b2[2]) - min(b1[0], b2[0])
    ch = max(b1[3], b2[3]) - min(b1[1], b2[1])
    c2 = cw * cw + ch * ch + eps
    rho2 = ((b2[0] + b2[2] - b1[0] - b1[2]) ** 2 + (b2[1] + b2[3] - b1[1] - b1[3]) ** 2) / 4.0
    v = 4.0 / (math.pi * math.pi) * (math.atan(w2 / h2) - math.atan(w1 / h1)) ** 2
    alpha = v / (v - iou + (1.0 + eps))
    return iou - (rho2 / c2 + v * alpha)


def log_softmax(row):
    m = max(row)
    lse = m + math.log(sum(math.exp(x - m) for x in row))
    return [x - lse for x in row]


def dfl(logits, target, reg_max=16):
    """Loss.cs:104-118 for ONE side: logits = reg_max bins, target = distance in grid units."""
    t = min(max(target, 0.0), reg_max - 1 - 0.01)
    tl = int(t)
    wl = (tl + 1) - t
    ls = log_softmax(logits)
    return -ls[tl] * wl - ls[tl + 1] * (1.0 - wl)


def bce_logits(x, t):
    """BCEWithLogitsLoss(reduction none): max(x,0) - x t + log(1 + exp(-|x|))."""
    return max(x, 0.0) - x * t + math.log1p(math.exp(-abs(x)))


def make_anchors(H, W, strides=(8, 16, 32)):
    """Tal.cs:313-338: levels concatenated, row-major, cell centre offset 0.5; returns [(ax, ay, stride)]."""
    out = []
    for s in strides:
        for y in range(H // s):
            for x in range(W // s):
                out.append((x + 0.5, y + 0.5, float(s)))
    return out


def decode(box_logits, anchors, reg_max=16):
    """Loss.cs:398-409 + Tal.cs:340-356 (xywh = false): box_logits [A][4*reg_max] -> xyxy in grid units."""
    out = []
    for a, (ax, ay, _) in enumerate(anchors):
        d = []
        for s in range(4):
            row = box_logits[a][s * reg_max:(s + 1) * reg_max]
            m = max(row)
            e = [math.exp(x - m) for x in row]
            d.append(sum(j * ej for j, ej in enumerate(e)) / sum(e))
        out.append((ax - d[0], ay - d[1], ax + d[2], ay + d[3]))
    return out


def tal_assign(ps_sig, pb_px, anchors, gts, nmax, nc, topk=10, alpha=0.5, beta=6.0, eps=1e-9, stride0=8, stride_val=16):
    """One image.  ps_sig [A][nc] probabilities, pb_px [A] xyxy pixels, anchors [(ax, ay, s)], gts = [(cls, x1, y1, x2, y2)] (real
    labels; rows up to nmax are zero padding).  Returns (fg [A] bool, gt_idx [A], tscore [A][nc], tbox [A] xyxy pixels)."""
    A = len(anchors)
    rows = list(gts) + [(0.0, 0.0, 0.0, 0.0, 0.0)] * (nmax - len(gts))
    mask_gt = [1.0 if (g[1] + g[2] + g[3] + g[4]) > 0.0 else 0.0 for g in rows]          # Loss.cs:431
    in_gts = [[0.0] * A for _ in rows]
    ov = [[0.0] * A for _ in rows]
    align = [[0.0] * A for _ in rows]
    for gi, g in enumerate(rows):
        cx, cy, w, h = (g[1] + g[3]) / 2, (g[2] + g[4]) / 2, g[3] - g[1], g[4] - g[2]       # Tal.cs:206-211
        if w < stride0 and mask_gt[gi]:
            w = float(stride_val)
        if h < stride0 and mask_gt[gi]:
            h = float(stride_val)
        x1, y1, x2, y2 = cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2
        for a, (ax, ay, s) in enumerate(anchors):
            px, py = ax * s, ay * s
            in_gts[gi][a] = 1.0 if min(px - x1, py - y1, x2 - px, y2 - py) > eps else 0.0
            if in_gts[gi][a] * mask_gt[gi]:
                ov[gi][a] = max(ciou(g[1:5], pb_px[a]), 0.0)                               # Tal.cs:136,141
                align[gi][a] = ps_sig[a][int(g[0])] ** alpha * ov[gi][a] ** beta
    mask_pos = [[0.0] * A for _ in rows]
    for gi in range(len(rows)):
        order = sorted(range(A), key=lambda a: (-align[gi][a], a))[:topk]                  # stable: lower index first on ties
        if not mask_gt[gi]:
            order = [0] * topk                                                             # Tal.cs:153
        cnt = [0] * A
        for a in order:
            cnt[a] += 1
        for a in range(A):
            sel = 1.0 if cnt[a] == 1 else 0.0                                              # count > 1 -> 0 (Tal.cs:164)
            mask_pos[gi][a] = sel * in_gts[gi][a] * mask_gt[gi]
    for a in range(A):                                                                     # Tal.cs:225-241
        if sum(mask_pos[gi][a] for gi in range(len(rows))) > 1:
            best = max(range(len(rows)), key=lambda gi: (ov[gi][a], -gi))                  # argmax over ALL rows, first max
            for gi in range(len(rows)):
                mask_pos[gi][a] = 1.0 if gi == best else 0.0
    fg = [sum(mask_pos[gi][a] for gi in range(len(rows))) > 0 for a in range(A)]
    gt_idx = [max(range(len(rows)), key=lambda gi: (mask_pos[gi][a], -gi)) for a in range(A)]
    pos_align = [max(align[gi][a] * mask_pos[gi][a] for a in range(A)) for gi in range(len(rows))]
    pos_ov = [max(ov[gi][a] * mask_pos[gi][a] for a in range(A)) for gi in range(len(rows))]
    tscore = [[0.0] * nc for _ in range(A)]
    tbox = [rows[gt_idx[a]][1:5] for a in range(A)]
    for a in range(A):
        norm = max(align[gi][a] * mask_pos[gi][a] * pos_ov[gi] / (pos_align[gi] + eps) for gi in range(len(rows)))
        if fg[a]:
            tscore[a][max(int(rows[gt_idx[a]][0]), 0)] = 1.0 * norm
    return fg, gt_idx, tscore, tbox


def _labels_by_image(batch_idx, cls, bboxes, B, H, W):
    per = [[] for _ in range(B)]
    for bi, c, bb in zip(batch_idx, cls, bboxes):                                          # Loss.cs:363-390 + Ops.cs:68-81
        cx, cy, w, h = bb[0] * W, bb[1] * H, bb[2] * W, bb[3] * H
        per[int(bi)].append((float(c), cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2))
    return per


def detection_loss(boxes, scores, batch_idx, cls, bboxes, H, W, nc, reg_max=16, targets=None):
    """boxes [B][4*reg_max][A], scores [B][nc][A] (reference layout), labels = collate arrays.  Returns
    (items [box, cls, dfl] (gains applied), total = sum(items) * B, targets) -- pass `targets` back in to freeze the assignment."""
    B = len(boxes)
    anchors = make_anchors(H, W)
    A = len(anchors)
    per = _labels_by_image(batch_idx, cls, bboxes, B, H, W)
    nmax = max((len(p) for p in per), default=0)
    pd = [[[boxes[b][c][a] for c in range(4 * reg_max)] for a in range(A)] for b in range(B)]
    ps = [[[scores[b][c][a] for c in range(nc)] for a in range(A)] for b in range(B)]
    pbox = [decode(pd[b], anchors, reg_max) for b in range(B)]
    if targets is None:
        targets = []
        for b in range(B):
            if nmax == 0:
                targets.append(([False] * A, [0] * A, [[0.0] * nc for _ in range(A)], [(0.0,) * 4] * A))
                continue
            sig = [[sigmoid(x) for x in ps[b][a]] for a in range(A)]
            px = [tuple(v * anchors[a][2] for v in pbox[b][a]) for a in range(A)]
            targets.append(tal_assign(sig, px, anchors, per[b], nmax, nc))
    tss = max(sum(sum(t) for b in range(B) for t in targets[b][2]), 1.0)
    l_cls = sum(bce_logits(ps[b][a][c], targets[b][2][a][c]) for b in range(B) for a in range(A) for c in range(nc)) / tss
    l_box = l_dfl = 0.0
    for b in range(B):
        fg, _, tscore, tbox = targets[b]
        for a in range(A):
            if not fg[a]:
                continue
            ax, ay, s = anchors[a]
            w = sum(tscore[a])
            tb = tuple(v / s for v in tbox[a])
            l_box += (1.0 - ciou(pbox[b][a], tb)) * w
            ltrb = (ax - tb[0], ay - tb[1], tb[2] - ax, tb[3] - ay)
            ltrb = [min(max(v, 0.0), reg_max - 1 - 0.01) for v in ltrb]                    # Tal.cs:375 (then again Loss.cs:108)
            l_dfl += sum(dfl(pd[b][a][k * reg_max:(k + 1) * reg_max], ltrb[k], reg_max) for k in range(4)) / 4.0 * w
    items = [7.5 * l_box / tss, 0.5 * l_cls, 1.5 * l_dfl / tss]
    return items, sum(items) * B, targets


def detection_loss_grads(boxes, scores, batch_idx, cls, bboxes, H, W, nc, reg_max=16, h=1e-6):
    """d(sum(items) * B) / d(boxes), d(scores) by central differences with the (no-grad) assignment frozen."""
    _, _, tg = detection_loss(boxes, scores, batch_idx, cls, bboxes, H, W, nc, reg_max)

    def total():
        return detection_loss(boxes, scores, batch_idx, cls, bboxes, H, W, nc, reg_max, targets=tg)[1]

    def fd(arr):
        g = [[[0.0] * len(arr[b][c]) for c in range(len(arr[b]))] for b in range(len(arr))]
        for b in range(len(arr)):
            for c in range(len(arr[b])):
                for a in range(len(arr[b][c])):
                    x = arr[b][c][a]
                    arr[b][c][a] = x + h
                    fp = total()
                    arr[b][c][a] = x - h
                    fm = total()
                    arr[b][c][a] = x
                    g[b][c][a] = (fp - fm) / (2 * h)
        return g

    return fd(boxes), fd(scores)


def bn_train_stats(y, gamma, beta, run_mean, run_var, eps=1e-3, momentum=0.03):
    """y [N][C] conv outputs (N = B*H*W).  Returns (z [N][C] normalised+affine, new running_mean, new running_var)."""
    N, C = len(y), len(y[0])
    z = [[0.0] * C for _ in range(N)]
    nm, nv = [], []
    for c in range(C):
        col = [y[n][c] for n in range(N)]
        mu = sum(col) / N
        var = sum((v - mu) ** 2 for v in col) / N
        for n in range(N):
            z[n][c] = (col[n] - mu) / math.sqrt(var + eps) * gamma[c] + beta[c]
        nm.append((1 - momentum) * run_mean[c] + momentum * mu)
        nv.append((1 - momentum) * run_var[c] + momentum * var * N / (N - 1))
    return z, nm, nv


def probiou(o1, o2, eps=1e-7):
    """Metrics.cs:223-258 (one pair of xywhr boxes), scalar fp64: Bhattacharyya distance of the two Gaussians N(xy, cov(w, h, r)),
    cov = R diag(w^2 / 12, h^2 / 12) R^T  ->  a = sxx, b = syy, c = sxy."""
    def cov(o):
        a, b = o[2] * o[2] / 12.0, o[3] * o[3] / 12.0
        cs, sn = math.cos(o[4]), math.sin(o[4])
        return a * cs * cs + b * sn * sn, a * sn * sn + b * cs * cs, (a - b) * cs * sn
    a1, b1, c1 = cov(o1)
    a2, b2, c2 = cov(o2)
    det = (a1 + a2) * (b1 + b2) - (c1 + c2) ** 2
    t1 = ((a1 + a2) * (o1[1] - o2[1]) ** 2 + (b1 + b2) * (o1[0] - o2[0]) ** 2) / (det + eps) * 0.25
    t2 = ((c1 + c2) * (o2[0] - o1[0]) * (o1[1] - o2[1])) / (det + eps) * 0.5
    t3 = math.log(det / (4.0 * math.sqrt(max(a1 * b1 - c1 * c1, 0.0) * max(a2 * b2 - c2 * c2, 0.0)) + eps) + eps) * 0.5
    bd = min(max(t1 + t2 + t3, eps), 100.0)
    return 1.0 - math.sqrt(1.0 - math.exp(-bd) + eps)
