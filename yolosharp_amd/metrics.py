"""Epoch-level detection metrics on the host (numpy): Metrics.ap_per_class / compute_ap / interp / smooth
(Utils/Metrics.cs:308-486) and the Detector.Val summary (Models/Detector.cs:128-150).  The per-image part of validation
(box_iou + match_predictions) runs on the device: Engine.val_match.  fp32 like the reference."""
import numpy as np

F = np.float32


def interp(x, xp, fp, left=0.0):
    """Metrics.cs:425-472: linear interpolation on the sorted (xp, fp); x >= max -> last value; x <= min -> `left`
    (the left rule is applied last, as upstream)."""
    x, xp, fp = np.asarray(x, F), np.asarray(xp, F), np.asarray(fp, F)
    order = np.argsort(xp, kind="stable")
    xs, fs = xp[order], fp[order]
    out = np.empty_like(x)
    out[x >= xs[-1]] = fs[-1]
    out[x <= xs[0]] = F(left)
    inner = (x > xs[0]) & (x < xs[-1])
    if inner.any():
        xi = x[inner]
        idx = np.clip(np.searchsorted(xs, xi, side="left") - 1, 0, xs.shape[0] - 2)
        x0, x1, y0, y1 = xs[idx], xs[idx + 1], fs[idx], fs[idx + 1]
        t = (xi - x0) / (x1 - x0)
        out[inner] = y0 + t * (y1 - y0)
    return out


def compute_ap(recall, precision):
    """Metrics.cs:396-422 (method "interp": 101-point COCO interpolation, trapezoid rule)."""
    mrec = np.concatenate(([F(0)], np.asarray(recall, F), [F(1)])).astype(F)
    mpre = np.concatenate(([F(1)], np.asarray(precision, F), [F(0)])).astype(F)
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    x = np.linspace(0, 1, 101, dtype=F)
    y = interp(x, mrec, mpre)
    ap = F(np.sum((x[1:] - x[:-1]) * (y[1:] + y[:-1]) / F(2), dtype=F))
    return float(ap), mpre, mrec


def smooth(y, f=0.05):
    """Metrics.cs:475-485: box filter; BOTH paddings use y[0] (upstream pads the tail with y[-1])."""
    y = np.asarray(y, F)
    nf = int(y.shape[0] * f * 2) // 2 * 2 + 1
    p = np.ones(nf // 2, F) * y[0]
    yp = np.concatenate((p, y, p))
    return np.convolve(yp, np.ones(nf, F) / F(nf), mode="valid").astype(F)


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """Metrics.cs:308-384.  tp [n, 10] bool, conf [n], pred_cls [n], target_cls [m].
    Returns dict(tp, fp, p, r, f1, ap [nc, 10], unique_classes, p_curve, r_curve, f1_curve, x, prec_values)."""
    tp = np.asarray(tp, bool).reshape(len(conf), -1)
    conf, pred_cls, target_cls = np.asarray(conf, F), np.asarray(pred_cls, F), np.asarray(target_cls, F)
    ii = np.argsort(-conf, kind="stable")
    tp, conf, pred_cls = tp[ii], conf[ii], pred_cls[ii]
    unique_classes, nt = np.unique(target_cls, return_counts=True)
    nc = unique_classes.shape[0]
    x = np.linspace(0, 1, 1000, dtype=F)
    ap = np.zeros((nc, tp.shape[1]), F)
    p_curve, r_curve = np.zeros((nc, 1000), F), np.zeros((nc, 1000), F)
    prec_values = []
    for ci, c in enumerate(unique_classes):
        i = pred_cls == c
        n_l, n_p = int(nt[ci]), int(i.sum())
        if n_p == 0 or n_l == 0:
            continue
        fpc = np.cumsum(~tp[i], 0).astype(F)
        tpc = np.cumsum(tp[i], 0).astype(F)
        recall = tpc / F(n_l + eps)
        r_curve[ci] = interp(-x, -conf[i], recall[:, 0], left=0)
        precision = tpc / (tpc + fpc)
        p_curve[ci] = interp(-x, -conf[i], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j], mpre, mrec = compute_ap(recall[:, j], precision[:, j])
            if j == 0:
                prec_values.append(interp(x, mrec, mpre))
    if not prec_values:
        prec_values = [np.zeros(1000, F)]
    f1_curve = F(2) * p_curve * r_curve / (p_curve + r_curve + F(eps))
    iii = int(np.argmax(smooth(f1_curve.mean(0), 0.1))) if nc else 0
    p, r, f1 = p_curve[:, iii], r_curve[:, iii], f1_curve[:, iii]
    tpn = np.round(r * nt.astype(F))
    fpn = np.round(tpn / (p + F(eps)) - tpn)
    return {"tp": tpn, "fp": fpn, "p": p, "r": r, "f1": f1, "ap": ap, "unique_classes": unique_classes.astype(np.int32),
            "p_curve": p_curve, "r_curve": r_curve, "f1_curve": f1_curve, "x": x, "prec_values": np.stack(prec_values)}


def val_summary(stats):
    """Detector.cs:138-141: (P, R, mAP50, mAP50-95) from ap_per_class' result."""
    ap = stats["ap"]
    return (float(stats["p"].mean()), float(stats["r"].mean()), float(ap[:, 0].mean()), float(ap[:, 1:].mean()))
