"""Host mirror of Models/Detector.cs on the engine: ImagePredict (:26-72) and Val (:76-154).

Everything per image / per anchor runs on the device (eval forward + decode, NMS, box_iou + match_predictions); only the
epoch-level ap_per_class (small) and the reference's integer truncation of the results stay on the host.  The model must have
been created with (height, width) = the padded image size: the reference pads bottom/right with 114 to a multiple of 32
before dividing by 255 (Detector.cs:35-41), this does the same."""
import numpy as np

from . import _lib
from . import metrics as M
from .model import AMPWrapper, v8DetectionLoss


class YoloResult:
    """Types/YoloResult: integer box (truncated like Detector.cs:52-68)."""
    __slots__ = ("ClassID", "Score", "CenterX", "CenterY", "Width", "Height", "Radian", "KeyPoints")

    def __init__(self, row):
        x, y = int(row[0]), int(row[1])
        rw, rh = int(row[2]) - x, int(row[3]) - y
        self.ClassID, self.Score = int(row[5]), float(row[4])
        self.CenterX, self.CenterY, self.Width, self.Height = x + rw // 2, y + rh // 2, rw, rh
        self.Radian, self.KeyPoints = 0.0, None

    def __repr__(self):
        return f"YoloResult(cls={self.ClassID}, score={self.Score:.3f}, cx={self.CenterX}, cy={self.CenterY}, w={self.Width}, h={self.Height})"


def pad_to_32(image_chw_u8):
    """Detector.cs:33-41: zero-pad mode with value 114 on the bottom / right to a multiple of 32, then / 255."""
    c, h, w = image_chw_u8.shape
    ph, pw = (32 - h % 32) % 32, (32 - w % 32) % 32
    out = np.full((c, h + ph, w + pw), 114.0, np.float32)
    out[:, :h, :w] = image_chw_u8
    return out / np.float32(255.0)


def clip_boxes(boxes, shape):
    """Ops.clip_boxes: xyxy clipped to (h, w)."""
    b = boxes.copy()
    b[:, [0, 2]] = np.clip(b[:, [0, 2]], 0, shape[1])
    b[:, [1, 3]] = np.clip(b[:, [1, 3]], 0, shape[0])
    return b


class Detector:
    def __init__(self, model):
        self.model, self.engine = model, model.engine
        self.amp = AMPWrapper(model)

    def ImagePredict(self, image_chw_u8, predict_threshold=0.25, iou_threshold=0.5):
        """image: uint8 / float [3,H,W] in 0..255 (RGB).  Returns a list of YoloResult."""
        x = pad_to_32(np.asarray(image_chw_u8, np.float32))[None]
        assert x.shape[2:] == (self.model.height, self.model.width), "create the model with the padded image size"
        inference, _ = self.amp.Evaluate(x)
        output, _ = self.engine.non_max_suppression(inference["boxes"], predict_threshold, iou_threshold)
        return [YoloResult(r) for r in output[0]]

    def Val(self, batches, conf_thres=0.1, iou_thres=0.7, max_det=300):
        """batches: iterable of dicts (images [B,3,H,W] in [0,1], batch_idx, cls, bboxes).  Returns
        (SUM over batches of the loss items, (P, R, mAP50, mAP50-95)) like Detector.Val (Detector.cs:76-154; :126 adds the items up).
        The predictions never leave the device between the eval forward, NMS and the matching (ys_model_pred_device ->
        ys_nms_batched -> ys_val_match_batched, all on device pointers); only the kept rows, their count and the `correct`
        matrix come back per batch, for the epoch-level ap_per_class on the host."""
        crit = v8DetectionLoss(self.model)
        eng, m = self.engine, self.model
        tps, confs, pcls, tcls = [], [], [], []
        loss_sum, count = None, 0
        nc = m.nc
        Cc = 4 + nc + m.NM
        stride = Cc - nc + 2
        d_rows = d_keep = d_cnt = d_cor = None
        cap_b = 0
        try:
            for data in batches:
                if np.asarray(data["batch_idx"]).size < 1:          # Detector.cs:91-94
                    continue
                images = np.ascontiguousarray(data["images"], np.float32)
                B = images.shape[0]
                # eval forward, then the criterion on the eval-mode preds (Detector.cs:95-97): one forward serves loss and NMS
                m.eval()
                m.forward(images, fetch=False)
                _, items = crit.forward(None, data)
                loss_sum = items if loss_sum is None else loss_sum + items
                if B > cap_b:
                    for p_ in (d_rows, d_keep, d_cnt, d_cor):
                        if p_ is not None:
                            eng.free(p_)
                    d_rows, d_keep = eng.malloc(B * max_det * stride * 4), eng.malloc(B * max_det * 8)
                    d_cnt, d_cor = eng.malloc(B * 4), eng.malloc(B * max_det * 10)
                    cap_b = B
                eng.nms_device(m.pred_device(), B, Cc, m.A, conf_thres, iou_thres, max_det, nc, d_rows, d_keep, d_cnt)
                bi = np.ascontiguousarray(np.asarray(data["batch_idx"], np.float32).reshape(-1))
                cl = np.ascontiguousarray(np.asarray(data["cls"], np.float32).reshape(-1))
                bb = np.ascontiguousarray(np.asarray(data["bboxes"], np.float32).reshape(-1, 4))
                d_lab = []
                try:                       # the per-batch label buffers are released on the error paths too
                    for a in (bi, cl, bb):
                        d_lab.append(eng.to_device(a))
                    _lib.check(eng.lib, eng.lib.ys_val_match_batched(eng.ctx, d_rows, d_cnt, 1, B, max_det, stride, d_lab[0], d_lab[1], d_lab[2],
                                                                     bi.shape[0], float(images.shape[3]), float(images.shape[2]), d_cor))
                    rows = eng.from_device(d_rows, (B, max_det, stride), np.float32)
                    cnt = eng.from_device(d_cnt, (B,), np.int32)
                    cor = eng.from_device(d_cor, (B, max_det, 10), np.uint8)
                finally:
                    for p_ in d_lab:
                        eng.free(p_)
                for b in range(B):
                    tps.append(cor[b, :cnt[b]].astype(bool)); confs.append(rows[b, :cnt[b], 4]); pcls.append(rows[b, :cnt[b], 5])
                    tcls.append(cl[bi == b])
                count += B
        finally:
            for p_ in (d_rows, d_keep, d_cnt, d_cor):
                if p_ is not None:
                    eng.free(p_)
        if not tps:
            return np.zeros(3, np.float32), (0.0, 0.0, 0.0, 0.0)
        stats = M.ap_per_class(np.concatenate(tps), np.concatenate(confs), np.concatenate(pcls), np.concatenate(tcls))
        return loss_sum, M.val_summary(stats)


class Segmenter(Detector):
    """Models/Segmenter.cs:28-84: ImagePredict with instance masks -- eval forward (pred carries the 32 mask coefficients
    per anchor), NMS (the coefficients ride along as extra columns), Ops.process_mask on the device with upsample=True, boxes
    clipped to the original image and masks cropped to it (the padded region is bottom / right, so the resize of
    Segmenter.cs:56-57 is an identity crop here)."""

    def ImagePredict(self, image_chw_u8, predict_threshold=0.25, iou_threshold=0.5):
        img = np.asarray(image_chw_u8)
        h, w = img.shape[1:]
        self.model.eval()
        inference, preds = self.model.forward_u8(np.ascontiguousarray(img, np.uint8)[None])
        proto = self.model.get_output("proto")[0]
        output, _ = self.engine.non_max_suppression(inference["boxes"], predict_threshold, iou_threshold, nc=self.model.nc)
        rows = output[0]
        results = []
        if len(rows):
            masks = self.engine.process_mask(proto, rows[:, 6:], rows[:, :4], (self.model.height, self.model.width), upsample=True)
            rows[:, :4] = clip_boxes(rows[:, :4], (h, w))
            masks = masks[:, :h, :w]
            for r, mk in zip(rows, masks):
                res = YoloResult(r)
                results.append((res, mk))
        return results

    def Val(self, batches, conf_thres=0.01, iou_thres=0.7, max_det=300):
        """Models/Segmenter.cs:85-185: per batch eval forward + loss, NMS (conf 0.01, IoU 0.7), and per image
        process_mask at the prototype resolution (:126 passes the proto size as `shape`, so the boxes are NOT rescaled -- kept),
        box IoU matching and mask IoU matching (Metrics.mask_iou on the overlap-encoded `masks`, :131-143) -- all on the device;
        the two epoch-level ap_per_class reductions stay on the host.  batches: dicts with images [B,3,H,W] in [0,1], batch_idx,
        cls, bboxes, masks [B,H/4,W/4].  Returns (summed loss items [5], box (P,R,mAP50,mAP50-95), mask (P,R,mAP50,mAP50-95))."""
        from .model import v8SegmentationLoss
        crit = v8SegmentationLoss(self.model)
        tps, tpms, confs, pcls, tcls = [], [], [], [], []
        loss_sum = None
        nc = self.model.nc
        for data in batches:
            if np.asarray(data["batch_idx"]).size < 1:          # Segmenter.cs:105-108
                continue
            images = np.ascontiguousarray(data["images"], np.float32)
            B, _, H, W = images.shape
            inference, _ = self.amp.Evaluate(images)                # Segmenter.cs:110-112: eval preds feed loss and NMS
            _, items = crit.forward(None, data)
            loss_sum = items if loss_sum is None else loss_sum + items
            proto = self.model.get_output("proto")
            mh, mw = proto.shape[2:]
            output, _ = self.engine.non_max_suppression(inference["boxes"], conf_thres, iou_thres, max_det=max_det, nc=nc)
            bi = np.asarray(data["batch_idx"], np.float32).reshape(-1)
            cl = np.asarray(data["cls"], np.float32).reshape(-1)
            bb = np.asarray(data["bboxes"], np.float32).reshape(-1, 4)
            gm = np.asarray(data["masks"], np.float32).reshape(B, mh, mw)
            for b, rows in enumerate(output):
                sel = bi == b
                true_cls = cl[sel]
                nl = int(sel.sum())
                masks = self.engine.process_mask(proto[b], rows[:, 6:], rows[:, :4], (mw, mh)) if len(rows) else np.zeros((0, mh, mw), bool)
                gt = bb[sel] * np.array([W, H, W, H], np.float32)
                gt_xyxy = np.concatenate((gt[:, :2] - gt[:, 2:] / 2, gt[:, :2] + gt[:, 2:] / 2), 1).astype(np.float32)
                iou = self.engine.box_iou(gt_xyxy, rows[:, :4])
                tps.append(self.engine.match_predictions(rows[:, 5], true_cls, iou))
                miou = self.engine.mask_iou(gm[b], nl, masks)
                tpms.append(self.engine.match_predictions(rows[:, 5], true_cls, miou))
                confs.append(rows[:, 4]); pcls.append(rows[:, 5]); tcls.append(true_cls)
        conf, pc, tc = np.concatenate(confs), np.concatenate(pcls), np.concatenate(tcls)
        box = M.val_summary(M.ap_per_class(np.concatenate(tps), conf, pc, tc))
        mask = M.val_summary(M.ap_per_class(np.concatenate(tpms), conf, pc, tc))
        return loss_sum, box, mask


class Obber(Detector):
    """Models/Obber.cs: ImagePredict (:28-68) = eval forward (pred = dist2rbox boxes, probabilities, angle) + rotated NMS, result =
    truncated centre / size + Radian; Val (:70-163) = eval forward + v8OBBLoss on the eval preds, rotated NMS (conf 0.01, IoU 0.7),
    Metrics.batch_probiou(labels xywh * scale + angle, predictions xywh + angle) -> match_predictions -> ap_per_class."""

    def ImagePredict(self, image_chw_u8, predict_threshold=0.25, iou_threshold=0.5):
        x = pad_to_32(np.asarray(image_chw_u8, np.float32))[None]
        assert x.shape[2:] == (self.model.height, self.model.width), "create the model with the padded image size"
        inference, _ = self.amp.Evaluate(x)
        output, _ = self.engine.non_max_suppression(inference["boxes"], predict_threshold, iou_threshold, nc=self.model.nc, rotated=True)
        results = []
        for r in output[0]:                                        # Obber.cs:55-63: the rotated rows stay xywh
            res = YoloResult.__new__(YoloResult)
            res.CenterX, res.CenterY, res.Width, res.Height = int(r[0]), int(r[1]), int(r[2]), int(r[3])
            res.Score, res.ClassID, res.Radian, res.KeyPoints = float(r[4]), int(r[5]), float(r[6]), None
            results.append(res)
        return results

    def Val(self, batches, conf_thres=0.01, iou_thres=0.7, max_det=300):
        from .model import v8OBBLoss
        crit = v8OBBLoss(self.model)
        tps, confs, pcls, tcls = [], [], [], []
        loss_sum = None
        nc = self.model.nc
        for data in batches:
            if np.asarray(data["batch_idx"]).size < 1:
                continue
            images = np.ascontiguousarray(data["images"], np.float32)
            B, _, H, W = images.shape
            inference, _ = self.amp.Evaluate(images)
            _, items = crit.forward(None, data)
            loss_sum = items if loss_sum is None else loss_sum + items
            output, _ = self.engine.non_max_suppression(inference["boxes"], conf_thres, iou_thres, max_det=max_det, nc=nc, rotated=True)
            bi = np.asarray(data["batch_idx"], np.float32).reshape(-1)
            cl = np.asarray(data["cls"], np.float32).reshape(-1)
            bb = np.asarray(data["bboxes"], np.float32).reshape(-1, 5)
            for b, rows in enumerate(output):
                sel = bi == b
                gt = np.concatenate((bb[sel, :4] * np.array([W, H, W, H], np.float32), bb[sel, 4:5]), 1).astype(np.float32)   # Obber.cs:108
                pred = np.concatenate((rows[:, :4], rows[:, 6:7]), 1).astype(np.float32)                                      # Obber.cs:102
                iou = self.engine.batch_probiou(gt, pred) if len(gt) and len(pred) else np.zeros((len(gt), len(pred)), np.float32)
                tps.append(self.engine.match_predictions(rows[:, 5], cl[sel], iou))
                confs.append(rows[:, 4]); pcls.append(rows[:, 5]); tcls.append(cl[sel])
        if not tps:
            return np.zeros(4, np.float32), (0.0, 0.0, 0.0, 0.0)
        stats = M.ap_per_class(np.concatenate(tps), np.concatenate(confs), np.concatenate(pcls), np.concatenate(tcls))
        return loss_sum, M.val_summary(stats)


class KeyPoint:
    __slots__ = ("X", "Y", "VisibilityScore")

    def __init__(self, x, y, v):
        self.X, self.Y, self.VisibilityScore = float(x), float(y), float(v)


class PoseDetector(Detector):
    """Models/PoseDetector.cs: ImagePredict (:39-98) = eval forward (pred carries the decoded keypoints) + NMS, result = the
    truncated box + KeyPoints (visibility 2.0 for 2-D keypoints); Val (:100-200) = eval forward + v8PoseLoss, NMS (conf 0.01, IoU
    0.7), box_iou matching and Metrics.kpt_iou (area = w * h * 0.53) matching -> ap_per_class twice."""

    def ImagePredict(self, image_chw_u8, predict_threshold=0.25, iou_threshold=0.5):
        x = pad_to_32(np.asarray(image_chw_u8, np.float32))[None]
        assert x.shape[2:] == (self.model.height, self.model.width), "create the model with the padded image size"
        inference, _ = self.amp.Evaluate(x)
        output, _ = self.engine.non_max_suppression(inference["boxes"], predict_threshold, iou_threshold, nc=self.model.nc)
        D = self.model.kpt_dim
        results = []
        for r in output[0]:
            res = YoloResult(r)
            kp = r[6:].reshape(-1, D)
            res.KeyPoints = [KeyPoint(k[0], k[1], k[2] if D == 3 else 2.0) for k in kp]
            results.append(res)
        return results

    def Val(self, batches, conf_thres=0.01, iou_thres=0.7, max_det=300):
        from .model import v8PoseLoss
        crit = v8PoseLoss(self.model)
        tps, tpps, confs, pcls, tcls = [], [], [], [], []
        loss_sum = None
        nc, K, D = self.model.nc, self.model.kpt_num, self.model.kpt_dim
        for data in batches:
            if np.asarray(data["batch_idx"]).size < 1:
                continue
            images = np.ascontiguousarray(data["images"], np.float32)
            B, _, H, W = images.shape
            inference, _ = self.amp.Evaluate(images)
            _, items = crit.forward(None, data)
            loss_sum = items if loss_sum is None else loss_sum + items
            output, _ = self.engine.non_max_suppression(inference["boxes"], conf_thres, iou_thres, max_det=max_det, nc=nc)
            bi = np.asarray(data["batch_idx"], np.float32).reshape(-1)
            cl = np.asarray(data["cls"], np.float32).reshape(-1)
            bb = np.asarray(data["bboxes"], np.float32).reshape(-1, 4)
            kp = np.asarray(data["keypoints"], np.float32).reshape(len(bi), K, -1)
            if kp.shape[2] == 2:                                   # PoseDetector.cs:144-148: "seen" column of ones
                kp = np.concatenate((kp, np.ones(kp.shape[:2] + (1,), np.float32)), 2)
            for b, rows in enumerate(output):
                sel = bi == b
                gt = bb[sel] * np.array([W, H, W, H], np.float32)
                gt_xyxy = np.concatenate((gt[:, :2] - gt[:, 2:] / 2, gt[:, :2] + gt[:, 2:] / 2), 1).astype(np.float32)
                tps.append(self.engine.match_predictions(rows[:, 5], cl[sel], self.engine.box_iou(gt_xyxy, rows[:, :4])))
                gk = (kp[sel] * np.array([W, H, 1.0], np.float32)).astype(np.float32)                       # PoseDetector.cs:153-155
                area = ((gt_xyxy[:, 2] - gt_xyxy[:, 0]) * (gt_xyxy[:, 3] - gt_xyxy[:, 1]) * np.float32(0.53)).astype(np.float32)
                oks = self.engine.kpt_iou(gk, rows[:, 6:].reshape(-1, K, D), area)
                tpps.append(self.engine.match_predictions(rows[:, 5], cl[sel], oks))
                confs.append(rows[:, 4]); pcls.append(rows[:, 5]); tcls.append(cl[sel])
        if not tps:
            return np.zeros(5, np.float32), (0.0, 0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 0.0)
        conf, pc, tc = np.concatenate(confs), np.concatenate(pcls), np.concatenate(tcls)
        box = M.val_summary(M.ap_per_class(np.concatenate(tps), conf, pc, tc))
        pose = M.val_summary(M.ap_per_class(np.concatenate(tpps), conf, pc, tc))
        return loss_sum, box, pose
