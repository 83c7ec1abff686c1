"""tests/kat_ref.py -- INDEPENDENT known-answer arithmetic for the loss path (test infrastructure only).

A second restatement of the cited C# that shares no code, no tensor library and no autograd with oracle/yolo_oracle.py:
scalar fp64 Python (`math` only), explicit loops, and central finite differences for the gradients.  It pins both the ATen
oracle and the HIP kernels (tests/test_kat.py) on small hand-built cases:

  ciou            Utils/Metrics.cs:36-111   (xywh = false, CIoU = true; h clamped to eps, alpha not detached)
  dfl             Utils/Loss.cs:94-120      (clamp(0, reg_max - 1 - 0.01), floor / floor+1 cross-entropies)
  tal_assign      Utils/Tal.cs:50-258       (in-GT test incl. the 8 px -> 16 px inflation, CIoU overlaps, score^0.5 * ov^6,
                                             top-10 with 'selected exactly once', multi-GT resolution, normalised targets)
  detection_loss  Utils/Loss.cs:411-477     (decode, assignment, BCE / CIoU / DFL terms, gains 7.5 / 0.5 / 1.5, tss = max(sum, 1))
  pose_loss       Utils/Loss.cs:870-1071    (KeypointLoss :169-188: OKS sigmas, kpt_loss_factor, visibility BCE; gains 12 / 1)
  obb_loss        Utils/Loss.cs:486-684     (2-px filter, RotatedTaskAlignedAssigner Tal.cs:260-310 incl. the in-place thin-box
                                             widening, probiou box term, rbox2dist DFL targets, angle term)
  seg_term        Utils/Loss.cs:786-863     (mask term: einsum coeff . proto, BCE vs the overlap-encoded ids, both Ops.crop_mask
                                             branches (Ops.cs:421-447), mean / normalised area, / foreground count, gain 7.5)
  probiou         Utils/Metrics.cs:137-160,264-283
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


# ----------------------------------------------------------------------------- v8PoseLoss (Loss.cs:870-1071), scalar
OKS_SIGMA = [v / 10.0 for v in (0.26, 0.25, 0.25, 0.35, 0.35, 0.79, 0.79, 0.72, 0.72, 0.62, 0.62, 1.07, 1.07, 0.87, 0.87, 0.89, 0.89)]


def pose_terms(kpts, targets, batch_idx, bboxes, keypoints, H, W, K, D):
    """kpts [B][K*D][A] raw head outputs; targets = the detection assignment (detection_loss(...)[2]).  Returns (pose, kobj) with
    the gains 12 / 1 applied (Loss.cs:962-963): means over (foreground anchors x K) of the OKS-style location term and the
    visibility BCE (KeypointLoss, Loss.cs:177-187; calculate_keypoints_loss :1033-1069)."""
    B = len(kpts)
    anchors = make_anchors(H, W)
    sig = OKS_SIGMA if (K == 17 and D == 3) else [1.0 / K] * K                             # Loss.cs:903-905
    rows_of = [[i for i, bi in enumerate(batch_idx) if int(bi) == b] for b in range(B)]    # slot g of image b -> label row
    n = 0
    lk = lo = 0.0
    for b in range(B):
        fg, gt_idx, _, tbox = targets[b]
        for a, (ax, ay, s) in enumerate(anchors):
            if not fg[a]:
                continue
            n += 1
            lab = keypoints[rows_of[b][gt_idx[a]]]
            area = (tbox[a][2] / s - tbox[a][0] / s) * (tbox[a][3] / s - tbox[a][1] / s)   # Loss.cs:1047-1049
            mask = [1.0 if (D == 2 or lab[k][2] != 0.0) else 0.0 for k in range(K)]
            fac = K / (sum(mask) + 1e-6)
            for k in range(K):
                px = kpts[b][k * D][a] * 2.0 + (ax - 0.5)                                  # kpts_decode (Loss.cs:967-974)
                py = kpts[b][k * D + 1][a] * 2.0 + (ay - 0.5)
                gx, gy = lab[k][0] * W / s, lab[k][1] * H / s
                e = ((px - gx) ** 2 + (py - gy) ** 2) / ((2.0 * sig[k]) ** 2 * (area + 1e-9) * 2.0)
                lk += fac * (1.0 - math.exp(-e)) * mask[k]
                if D == 3:
                    lo += bce_logits(kpts[b][k * D + 2][a], mask[k])
    if n == 0:
        return 0.0, 0.0
    return 12.0 * lk / (n * K), 1.0 * lo / (n * K)


def pose_loss(boxes, scores, kpts, batch_idx, cls, bboxes, keypoints, H, W, nc, K, D, reg_max=16, targets=None):
    """Items (box, pose, kobj, cls, dfl), total = sum(items) * B, targets."""
    det, _, targets = detection_loss(boxes, scores, batch_idx, cls, bboxes, H, W, nc, reg_max, targets)
    pose, kobj = pose_terms(kpts, targets, batch_idx, bboxes, keypoints, H, W, K, D)
    items = [det[0], pose, kobj, det[1], det[2]]
    return items, sum(items) * len(boxes), targets


def fd_grad(arr, total, h=1e-6):
    """Central differences of total() w.r.t. every element of the nested list arr [B][C][A]."""
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


# ----------------------------------------------------------------------------- v8OBBLoss (Loss.cs:486-684), scalar
def obb_decode(box_logits, angle, anchors, reg_max=16):
    """bbox_decode (Loss.cs:634-645) = dist2rbox (Tal.cs:389-408) + angle: [A] of (x, y, w, h, angle), grid units."""
    out = []
    for a, (ax, ay, _) in enumerate(anchors):
        d = []
        for s in range(4):
            row = box_logits[a][s * reg_max:(s + 1) * reg_max]
            m = max(row)
            e = [math.exp(x - m) for x in row]
            d.append(sum(j * ej for j, ej in enumerate(e)) / sum(e))
        cs, sn = math.cos(angle[a]), math.sin(angle[a])
        xf, yf = (d[2] - d[0]) / 2.0, (d[3] - d[1]) / 2.0
        out.append((xf * cs - yf * sn + ax, xf * sn + yf * cs + ay, d[0] + d[2], d[1] + d[3], angle[a]))
    return out


def rot_tal_assign(ps_sig, pb_px, anchors, gts, nmax, nc, topk=10, alpha=0.5, beta=6.0, eps=1e-9, stride0=8, stride_val=16):
    """RotatedTaskAlignedAssigner (Tal.cs:260-310) on one image.  gts = [(cls, x, y, w, h, r)] pixels; pb_px [A] xywhr pixels.
    Returns (fg, gt_idx, tscore, tbox [A] xywhr pixels) -- tbox carries the IN-PLACE widening of thin boxes (Tal.cs:283-287)."""
    A = len(anchors)
    rows = [list(g) for g in gts] + [[0.0] * 6 for _ in range(nmax - len(gts))]
    mask_gt = [1.0 if sum(g[1:6]) > 0.0 else 0.0 for g in rows]                            # Loss.cs:571
    for gi, g in enumerate(rows):                                                           # the reference edits gt_bboxes itself
        if mask_gt[gi] and g[3] < stride0:
            g[3] = float(stride_val)
        if mask_gt[gi] and g[4] < stride0:
            g[4] = float(stride_val)
    in_gts = [[0.0] * A for _ in rows]
    ov = [[0.0] * A for _ in rows]
    align = [[0.0] * A for _ in rows]
    for gi, g in enumerate(rows):
        cs, sn = math.cos(g[5]), math.sin(g[5])
        v1 = (g[3] / 2 * cs, g[3] / 2 * sn)
        v2 = (-g[4] / 2 * sn, g[4] / 2 * cs)
        pa = (g[1] + v1[0] + v2[0], g[2] + v1[1] + v2[1])                                  # Ops.cs:30-33
        pb = (g[1] + v1[0] - v2[0], g[2] + v1[1] - v2[1])
        pd_ = (g[1] - v1[0] + v2[0], g[2] - v1[1] + v2[1])
        ab, ad = (pb[0] - pa[0], pb[1] - pa[1]), (pd_[0] - pa[0], pd_[1] - pa[1])
        nab, nad = ab[0] ** 2 + ab[1] ** 2, ad[0] ** 2 + ad[1] ** 2
        for a, (ax, ay, s) in enumerate(anchors):
            ap = (ax * s - pa[0], ay * s - pa[1])
            dab, dad = ap[0] * ab[0] + ap[1] * ab[1], ap[0] * ad[0] + ap[1] * ad[1]
            in_gts[gi][a] = 1.0 if (0 <= dab <= nab and 0 <= dad <= nad) else 0.0          # Tal.cs:306
            if in_gts[gi][a] * mask_gt[gi]:
                ov[gi][a] = max(probiou(g[1:6], pb_px[a]), 0.0)                            # Tal.cs:267-270
                align[gi][a] = ps_sig[a][int(g[0])] ** alpha * ov[gi][a] ** beta
    mask_pos = [[0.0] * A for _ in rows]
    for gi in range(len(rows)):
        order = sorted(range(A), key=lambda a: (-align[gi][a], a))[:topk]
        if not mask_gt[gi]:
            order = [0] * topk
        cnt = [0] * A
        for a in order:
            cnt[a] += 1
        for a in range(A):
            mask_pos[gi][a] = (1.0 if cnt[a] == 1 else 0.0) * in_gts[gi][a] * mask_gt[gi]
    for a in range(A):
        if sum(mask_pos[gi][a] for gi in range(len(rows))) > 1:
            best = max(range(len(rows)), key=lambda gi: (ov[gi][a], -gi))
            for gi in range(len(rows)):
                mask_pos[gi][a] = 1.0 if gi == best else 0.0
    fg = [sum(mask_pos[gi][a] for gi in range(len(rows))) > 0 for a in range(A)]
    gt_idx = [max(range(len(rows)), key=lambda gi: (mask_pos[gi][a], -gi)) for a in range(A)]
    pos_align = [max(align[gi][a] * mask_pos[gi][a] for a in range(A)) for gi in range(len(rows))]
    pos_ov = [max(ov[gi][a] * mask_pos[gi][a] for a in range(A)) for gi in range(len(rows))]
    tscore = [[0.0] * nc for _ in range(A)]
    tbox = [tuple(rows[gt_idx[a]][1:6]) for a in range(A)]
    for a in range(A):
        norm = max(align[gi][a] * mask_pos[gi][a] * pos_ov[gi] / (pos_align[gi] + eps) for gi in range(len(rows)))
        if fg[a]:
            tscore[a][max(int(rows[gt_idx[a]][0]), 0)] = 1.0 * norm
    return fg, gt_idx, tscore, tbox


def obb_loss(boxes, scores, angle_logit, batch_idx, cls, bboxes5, H, W, nc, reg_max=16, targets=None):
    """boxes [B][4*reg_max][A], scores [B][nc][A], angle_logit [B][1][A] (angle = (sigmoid - 0.25) * pi, Head.cs:429); labels
    [N][5] = normalised cx, cy, w, h + angle.  Returns (items [box, cls, dfl, angle], total = sum * B, targets)."""
    B = len(boxes)
    anchors = make_anchors(H, W)
    A = len(anchors)
    per = [[] for _ in range(B)]
    for bi, c, bb in zip(batch_idx, cls, bboxes5):
        if bb[2] * W >= 2 and bb[3] * H >= 2:                                              # Loss.cs:561-563
            per[int(bi)].append((float(c), bb[0] * W, bb[1] * H, bb[2] * W, bb[3] * H, bb[4]))
    nmax = max((len(p) for p in per), default=0)
    pd = [[[boxes[b][c][a] for c in range(4 * reg_max)] for a in range(A)] for b in range(B)]
    ps = [[[scores[b][c][a] for c in range(nc)] for a in range(A)] for b in range(B)]
    ang = [[(sigmoid(angle_logit[b][0][a]) - 0.25) * math.pi for a in range(A)] for b in range(B)]
    pbox = [obb_decode(pd[b], ang[b], anchors, reg_max) for b in range(B)]
    if targets is None:
        targets = []
        for b in range(B):
            if nmax == 0:
                targets.append(([False] * A, [0] * A, [[0.0] * nc for _ in range(A)], [(0.0,) * 5] * A))
                continue
            sig = [[sigmoid(x) for x in ps[b][a]] for a in range(A)]
            px = [(p[0] * anchors[a][2], p[1] * anchors[a][2], p[2] * anchors[a][2], p[3] * anchors[a][2], p[4]) for a, p in enumerate(pbox[b])]
            targets.append(rot_tal_assign(sig, px, anchors, per[b], nmax, nc))
    tss = max(sum(sum(t) for b in range(B) for t in targets[b][2]), 1.0)
    l_cls = sum(bce_logits(ps[b][a][c], targets[b][2][a][c]) for b in range(B) for a in range(A) for c in range(nc)) / tss
    l_box = l_dfl = l_ang = 0.0
    for b in range(B):
        fg, _, tscore, tbox = targets[b]
        for a in range(A):
            if not fg[a]:
                continue
            ax, ay, s = anchors[a]
            w = sum(tscore[a])
            tb = (tbox[a][0] / s, tbox[a][1] / s, tbox[a][2] / s, tbox[a][3] / s, tbox[a][4])   # Loss.cs:596
            l_box += (1.0 - probiou(pbox[b][a], tb)) * w                                   # Loss.cs:203-204
            ox, oy = tb[0] - ax, tb[1] - ay                                                 # rbox2dist (Tal.cs:418-453)
            ct, st = math.cos(tb[4]), math.sin(tb[4])
            xf, yf = ox * ct + oy * st, -ox * st + oy * ct
            ltrb = (tb[2] / 2 - xf, tb[3] / 2 - yf, tb[2] / 2 + xf, tb[3] / 2 + yf)
            ltrb = [min(max(v, 0.0), reg_max - 1 - 0.01) for v in ltrb]
            l_dfl += sum(dfl(pd[b][a][k * reg_max:(k + 1) * reg_max], ltrb[k], reg_max) for k in range(4)) / 4.0 * w
            lar = math.log((tb[2] + 1e-9) / (tb[3] + 1e-9))                                # calculate_angle_loss (Loss.cs:657-676)
            dlt = pbox[b][a][4] - tb[4]
            wrapped = dlt - round(dlt / math.pi) * math.pi                                  # python round = half-to-even, like torch
            l_ang += math.exp(-(lar ** 2) / 9.0) * math.sin(2.0 * wrapped) ** 2 * w
    items = [7.5 * l_box / tss, 0.5 * l_cls, 1.5 * l_dfl / tss, 1.0 * l_ang / tss]
    return items, sum(items) * B, targets


# ----------------------------------------------------------------------------- v8SegmentationLoss mask term (Loss.cs:786-863), scalar
def seg_term(coeff, proto, targets, masks, H, W, trunc_crop=False):
    """coeff [B][nm][A] mask coefficients, proto [B][nm][mh][mw], masks [B][mh][mw] overlap-encoded instance ids (id = slot + 1),
    targets = the detection assignment.  Returns the mask item with the gain 7.5 applied (Loss.cs:777): for every foreground anchor the
    BCE of (coeff . proto) against (masks == slot + 1), cropped to the target box in mask units (Ops.crop_mask: x1 <= col < x2 in the
    float form :437-447, C# int truncation + slices in the CPU form :421-435 when trunc_crop), mean over ALL mask pixels, divided by the
    normalised box area; summed and divided by the number of foreground anchors."""
    B = len(coeff)
    nm, mh, mw = len(proto[0]), len(proto[0][0]), len(proto[0][0][0])
    n_fg = sum(sum(1 for f in t[0] if f) for t in targets)
    if n_fg == 0:
        return 0.0
    total = 0.0
    for b in range(B):
        fg, gt_idx, _, tbox = targets[b]
        entries = [a for a in range(len(fg)) if fg[a]]
        n_rows = len(entries)
        for a in entries:
            nb = (tbox[a][0] / W, tbox[a][1] / H, tbox[a][2] / W, tbox[a][3] / H)           # Loss.cs:808-809
            area = (nb[2] - nb[0]) * (nb[3] - nb[1])                                        # xyxy2xywh(...)[2:].prod (Loss.cs:812)
            x1, y1, x2, y2 = nb[0] * mw, nb[1] * mh, nb[2] * mw, nb[3] * mh                  # Loss.cs:815
            if trunc_crop and n_rows < 50:
                cols = range(max(int(x1), 0), min(int(x2), mw)) if int(x2) >= 0 else range(0)
                rows = range(max(int(y1), 0), min(int(y2), mh)) if int(y2) >= 0 else range(0)
                inside = lambda r, c: r in rows and c in cols
            else:
                inside = lambda r, c: (c >= x1) and (c < x2) and (r >= y1) and (r < y2)
            s = 0.0
            for r in range(mh):
                for c in range(mw):
                    if not inside(r, c):
                        continue
                    z = sum(coeff[b][k][a] * proto[b][k][r][c] for k in range(nm))          # einsum "in,nhw->ihw" (Loss.cs:790)
                    t = 1.0 if masks[b][r][c] == gt_idx[a] + 1 else 0.0                     # Loss.cs:826
                    s += bce_logits(z, t)
            total += s / (mh * mw) / area                                                   # .mean(1, 2) / area (Loss.cs:792)
    return 7.5 * total / n_fg                                                               # Loss.cs:861, gain :777
