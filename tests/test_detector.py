"""Detector.ImagePredict / Val host mirror (Models/Detector.cs:26-154) end to end on the engine vs the oracle pieces."""
import numpy as np
import pytest
import torch

from conftest import BACKENDS
from oracle import yolo_oracle as O
from test_model import make_ref


@pytest.mark.parametrize("backend", BACKENDS)
def test_predict_and_val(backend, engine):
    from yolosharp_amd.detector import Detector, pad_to_32
    from yolosharp_amd.model import Yolov8
    from yolosharp_amd import metrics as M
    nc, H, W, B = 80, 32, 32, 2
    ref = make_ref(nc=nc)
    m = Yolov8(engine, nc=nc, size="n", height=H, width=W, max_batch=B, dtype="f32")
    m.load_state_dict({k: v.detach().numpy() for k, v in ref.state_dict().items()})
    det = Detector(m)
    # ---- ImagePredict: 20x29 image -> padded with 114 to 32x32 (Detector.cs:33-41)
    img = (torch.rand(3, 20, 29, generator=torch.Generator().manual_seed(0)) * 255).floor()
    x = pad_to_32(img.numpy())
    assert x.shape == (3, 32, 32) and x[0, 31, 31] == np.float32(114.0 / 255.0) and x[1, 10, 10] == np.float32(img[1, 10, 10] / 255.0)
    res = det.ImagePredict(img.numpy(), predict_threshold=0.001, iou_threshold=0.7)
    # the device-side uint8 input path (pad 114, / 255, pack) feeds the network the same tensor
    m.eval()
    inf_u8, _ = m.forward_u8(img.numpy().astype(np.uint8)[None])
    inf_f, _ = m.forward(x[None])
    assert np.array_equal(inf_u8["boxes"], inf_f["boxes"])
    ref.eval()
    with torch.no_grad():
        rinf, _ = ref(torch.from_numpy(x)[None])
    p = rinf["boxes"].numpy().copy()
    rows, _ = engine.non_max_suppression(p, 0.001, 0.7)          # NMS itself is pinned bit-exactly in test_nms.py
    assert len(res) > 0 and abs(len(res) - len(rows[0])) <= max(2, len(rows[0]) // 10)
    assert all(isinstance(r.CenterX, int) and 0 <= r.ClassID < nc for r in res)
    # ---- Val: loss items + (P, R, mAP50, mAP50-95) through the device matching; running statistics are left alone
    g = torch.Generator().manual_seed(4)
    tb = O.synthetic_batch(B, H, W, nc, seed=20, kmax=4)
    d = {k: v.numpy() for k, v in tb.items()}
    d["images"] = torch.rand(B, 3, H, W, generator=g).numpy()
    bn_before = m.state_dict()["model.0.bn.running_mean"].copy()
    empty = {"images": d["images"], "batch_idx": np.zeros(0, np.float32), "cls": np.zeros(0, np.float32), "bboxes": np.zeros((0, 4), np.float32)}
    loss_items, (P, R, m50, m5095) = det.Val([empty, d], conf_thres=0.001)      # an empty batch is skipped (Detector.cs:91-94)
    # the validation loss is the criterion on the EVAL-mode preds (running statistics), Detector.cs:95-97
    with torch.no_grad():
        _, rpreds = ref(torch.from_numpy(d["images"]))
        _, ritems = O.v8DetectionLoss(nc)(rpreds, tb)
    assert loss_items.shape == (3,) and np.allclose(loss_items, ritems.numpy(), rtol=1e-3, atol=1e-5), (loss_items, ritems)
    assert 0.0 <= m5095 <= m50 <= 1.0 and 0.0 <= P <= 1.0
    assert np.array_equal(bn_before, m.state_dict()["model.0.bn.running_mean"])
    m.close()


@pytest.mark.gpu
def test_segmenter_predict_gpu():
    """Segmenter.ImagePredict (Segmenter.cs:28-84) on the device path vs the oracle pieces on the same NMS rows."""
    from yolosharp_amd import Engine
    from yolosharp_amd.detector import Segmenter, pad_to_32
    from yolosharp_amd.model import Yolov8Segment
    from test_segment import make_ref
    eng = Engine(0)
    nc, H, W = 80, 128, 160
    ref = make_ref(8, nc, "n")
    m = Yolov8Segment(eng, nc=nc, size="n", height=H, width=W, max_batch=1, dtype="f32")
    m.load_state_dict({k: v.detach().numpy() for k, v in ref.state_dict().items()})
    img = (torch.rand(3, 117, 150, generator=torch.Generator().manual_seed(1)) * 255).floor()
    res = Segmenter(m).ImagePredict(img.numpy(), predict_threshold=0.01, iou_threshold=0.7)
    assert len(res) > 0
    ref.eval()
    with torch.no_grad():
        rinf, _ = ref(torch.from_numpy(pad_to_32(img.numpy()))[None])
    rows, _ = eng.non_max_suppression(rinf["boxes"].numpy().copy(), 0.01, 0.7, nc=nc)
    rows = torch.from_numpy(rows[0])
    rmasks = O.process_mask(rinf["proto"][0], rows[:, 6:], rows[:, :4].clone(), (H, W), upsample=True).numpy().astype(bool)[:, :117, :150]
    n = min(len(res), rmasks.shape[0])
    assert abs(len(res) - rmasks.shape[0]) <= max(2, rmasks.shape[0] // 10)
    agree = np.mean([np.mean(res[i][1] == rmasks[i]) for i in range(n)])
    assert agree > 0.98 and res[0][1].shape == (117, 150)
    m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_segmenter_val(backend, engine):
    """Segmenter.Val (Segmenter.cs:85-185) through the device pieces vs the same loop on the oracle's pieces (process_mask at
    proto resolution with un-rescaled boxes, box_iou + mask_iou -> match_predictions -> ap_per_class twice)."""
    from yolosharp_amd.detector import Segmenter
    from yolosharp_amd.model import Yolov8Segment
    from yolosharp_amd import metrics as M
    from test_segment import make_ref
    nc, H, W, B = 80, 64, 64, 2
    ref = make_ref(8, nc, "n")
    m = Yolov8Segment(engine, nc=nc, size="n", height=H, width=W, max_batch=B, dtype="f32")
    m.load_state_dict({k: v.detach().numpy() for k, v in ref.state_dict().items()})
    tb = O.synthetic_batch(B, H, W, nc, seed=21, kmax=4)
    tb["masks"] = O.synthetic_masks(tb, B, H // 4, W // 4)
    d = {k: v.numpy() for k, v in tb.items()}
    d["images"] = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(4)).numpy()
    before = m.state_dict()
    loss_items, box, mask = Segmenter(m).Val([d], conf_thres=0.001)
    after = m.state_dict()
    assert all(np.array_equal(before[k], after[k]) for k in before if "running" in k)
    assert loss_items.shape == (5,) and np.all(np.isfinite(loss_items))
    # the oracle's pieces on the engine's own NMS rows (NMS, forward and process_mask are pinned in their own tests)
    m.eval()
    inference, _ = m.forward(d["images"])
    proto = torch.from_numpy(m.get_output("proto"))
    output, _ = engine.non_max_suppression(inference["boxes"], 0.001, 0.7, nc=nc)
    tps, tpms, confs, pcls, tcls = [], [], [], [], []
    for b, rows in enumerate(output):
        rows = torch.from_numpy(rows)
        sel = tb["batch_idx"].view(-1) == b
        tcl = tb["cls"].view(-1)[sel]
        nl = int(sel.sum())
        masks = O.process_mask(proto[b], rows[:, 6:], rows[:, :4].clone(), (W // 4, H // 4))
        gt = O.xywh2xyxy(tb["bboxes"][sel] * torch.tensor([W, H, W, H], dtype=torch.float32))
        tps.append(O.match_predictions(rows[:, 5], tcl, O.box_iou(gt, rows[:, :4])).numpy())
        bm = (tb["masks"][b].view(1, H // 4, W // 4) == torch.arange(1, nl + 1).view(nl, 1, 1)).float()
        tpms.append(O.match_predictions(rows[:, 5], tcl, O.mask_iou(bm.flatten(1), masks.flatten(1).float())).numpy())
        confs.append(rows[:, 4].numpy()); pcls.append(rows[:, 5].numpy()); tcls.append(tcl.numpy())
    conf, pc, tc = np.concatenate(confs), np.concatenate(pcls), np.concatenate(tcls)
    assert len(conf) > 0
    rbox = M.val_summary(M.ap_per_class(np.concatenate(tps), conf, pc, tc))
    rmask = M.val_summary(M.ap_per_class(np.concatenate(tpms), conf, pc, tc))
    assert np.allclose(box, rbox, atol=1e-6) and np.allclose(mask, rmask, atol=1e-6), (box, rbox, mask, rmask)
    m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_letterbox_and_rectangle(backend, engine):
    """Augment.LetterBox / Augment.Rectangle on the device (Data/Augment.cs:698-857) vs the oracle restatement (ATen nearest resize,
    constant pad): every pixel and both offsets, uint8 images (colour 114) and fp32 masks (colour 0), up- and down-scaling."""
    rng = np.random.default_rng(0)
    cases = [((3, 480, 640), 640, 640), ((3, 1080, 810), 640, 640), ((3, 37, 53), 64, 96), ((1, 120, 160), 160, 160), ((3, 640, 640), 640, 640)]
    for shape, rw, rh in cases:
        img = rng.integers(0, 256, shape).astype(np.uint8)
        out, pl, pu = engine.letterbox(img, rw, rh, 114)
        rpl, rpu, ref = O.letterbox_image(img, rw, rh, 114)
        assert (pl, pu) == (rpl, rpu) and out.shape == (shape[0], rh, rw)
        assert np.array_equal(out, ref.numpy()), shape
        m = rng.integers(0, 5, (1,) + shape[1:]).astype(np.float32)
        outm, plm, pum = engine.letterbox(m, rw // 4, rh // 4, 0)
        rplm, rpum, refm = O.letterbox_image(m, rw // 4, rh // 4, 0)
        assert (plm, pum) == (rplm, rpum) and np.array_equal(outm, refm.numpy())
    # Rectangle: fit box = the label's resized shape, canvas = its rectangle shape
    img = rng.integers(0, 256, (3, 300, 500)).astype(np.uint8)
    out, pl, pu = engine.letterbox(img, 640, 640, 114, rectangle_shape=(640, 416))
    rpl, rpu, ref = O.letterbox_image(img, 640, 640, 114, canvas=(640, 416))
    assert (pl, pu) == (rpl, rpu) and np.array_equal(out, ref.numpy())
    from yolosharp_amd import YsError
    with pytest.raises(YsError):
        engine.letterbox(img, 640, 640, 114, rectangle_shape=(320, 100))      # the resized image does not fit the canvas


@pytest.mark.parametrize("backend", BACKENDS)
def test_obber_predict_and_val(backend, engine):
    """Obber.ImagePredict / Val (Obber.cs:28-163) through the device pieces vs the same loop on the oracle's pieces (rotated NMS rows
    -> batch_probiou -> match_predictions -> ap_per_class); the validation loss is v8OBBLoss on the eval-mode preds."""
    from yolosharp_amd.detector import Obber
    from yolosharp_amd.model import Yolov8Obb
    from yolosharp_amd import metrics as M
    from test_obb_pose import make_ref
    nc, H, W, B = 4, 64, 64, 2
    ref = make_ref(O.Yolov8Obb, nc, "n", seed=3)
    with torch.no_grad():
        for seq in ref.model[-1].cv3:
            seq[2].bias.add_(1.0); seq[2].weight.mul_(8.0)
    m = Yolov8Obb(engine, nc=nc, size="n", height=H, width=W, max_batch=B, dtype="f32")
    m.load_state_dict({k: v.detach().numpy() for k, v in ref.state_dict().items()})
    tb = O.synthetic_obb_batch(B, H, W, nc, seed=5, kmax=4)
    d = {k: v.numpy() for k, v in tb.items()}
    d["images"] = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(4)).numpy()
    ob = Obber(m)
    loss_items, box = ob.Val([d], conf_thres=0.3)
    ref.eval()
    with torch.no_grad():
        rinf, rpreds = ref(torch.from_numpy(d["images"]))
        _, ritems = O.v8OBBLoss(nc)(rpreds, tb)
    assert loss_items.shape == (4,) and np.allclose(loss_items, ritems.numpy(), rtol=2e-3, atol=1e-4)
    output, _ = O.non_max_suppression_rotated(rinf["boxes"], 0.3, 0.7, nc=nc)
    tps, confs, pcls, tcls = [], [], [], []
    scale = torch.tensor([W, H, W, H], dtype=torch.float32)
    for b, rows in enumerate(output):
        sel = tb["batch_idx"].view(-1) == b
        gt = torch.cat((tb["bboxes"][sel][:, :4] * scale, tb["bboxes"][sel][:, 4:5]), 1)
        pred = torch.cat((rows[:, :4], rows[:, 6:7]), 1)
        tps.append(O.match_predictions(rows[:, 5], tb["cls"].view(-1)[sel], O.batch_probiou(gt, pred)).numpy())
        confs.append(rows[:, 4].numpy()); pcls.append(rows[:, 5].numpy()); tcls.append(tb["cls"].view(-1)[sel].numpy())
    assert sum(len(c) for c in confs) > 4
    rbox = M.val_summary(M.ap_per_class(np.concatenate(tps), np.concatenate(confs), np.concatenate(pcls), np.concatenate(tcls)))
    assert np.allclose(box, rbox, atol=1e-6), (box, rbox)
    img = (d["images"][0] * 255).astype(np.uint8)
    res = ob.ImagePredict(img, 0.5, 0.3)
    want, _ = O.non_max_suppression_rotated(rinf["boxes"][:1], 0.5, 0.3, nc=nc)
    assert len(res) == len(want[0]) > 0
    for r, w in zip(res, want[0].numpy()):
        assert (r.ClassID, r.Width, r.Height) == (int(w[5]), int(w[2]), int(w[3])) and abs(r.Radian - w[6]) < 1e-3 and abs(r.Score - w[4]) < 1e-3
    m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_pose_detector_predict_and_val(backend, engine):
    """PoseDetector.ImagePredict / Val (PoseDetector.cs:39-200): NMS rows carry the decoded keypoints; box_iou matching and OKS
    matching (kpt_iou, area * 0.53) vs the same loop on the oracle's pieces."""
    from yolosharp_amd.detector import PoseDetector
    from yolosharp_amd.model import Yolov8Pose
    from yolosharp_amd import metrics as M
    from test_obb_pose import make_ref
    nc, H, W, B, K, D = 1, 64, 64, 2, 17, 3
    ref = make_ref(O.Yolov8Pose, nc, "n", seed=3)
    with torch.no_grad():
        for seq in ref.model[-1].cv3:
            seq[2].bias.add_(1.5); seq[2].weight.mul_(8.0)
    m = Yolov8Pose(engine, nc=nc, size="n", height=H, width=W, max_batch=B, dtype="f32")
    m.load_state_dict({k: v.detach().numpy() for k, v in ref.state_dict().items()})
    tb = O.synthetic_batch(B, H, W, nc, seed=21, kmax=4)
    tb["keypoints"] = O.synthetic_keypoints(tb, K, D)
    d = {k: v.numpy() for k, v in tb.items()}
    d["images"] = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(4)).numpy()
    pdt = PoseDetector(m)
    loss_items, box, pose = pdt.Val([d], conf_thres=0.3)
    ref.eval()
    with torch.no_grad():
        rinf, rpreds = ref(torch.from_numpy(d["images"]))
        _, ritems = O.v8PoseLoss(nc, K, D)(rpreds, tb)
    assert loss_items.shape == (5,) and np.allclose(loss_items, ritems.numpy(), rtol=2e-3, atol=1e-4)
    m.eval()
    inference, _ = m.forward(d["images"])
    output, _ = engine.non_max_suppression(inference["boxes"], 0.3, 0.7, nc=nc)
    tps, tpps, confs, pcls, tcls = [], [], [], [], []
    for b, rows in enumerate(output):
        rows = torch.from_numpy(rows)
        sel = tb["batch_idx"].view(-1) == b
        tcl = tb["cls"].view(-1)[sel]
        gt = O.xywh2xyxy(tb["bboxes"][sel] * torch.tensor([W, H, W, H], dtype=torch.float32))
        tps.append(O.match_predictions(rows[:, 5], tcl, O.box_iou(gt, rows[:, :4])).numpy())
        area = O.xyxy2xywh(gt)[:, 2:].prod(1) * 0.53
        oks = O.kpt_iou(tb["keypoints"][sel] * torch.tensor([W, H, 1.0]), rows[:, 6:].view(-1, K, D), area)
        tpps.append(O.match_predictions(rows[:, 5], tcl, oks).numpy())
        confs.append(rows[:, 4].numpy()); pcls.append(rows[:, 5].numpy()); tcls.append(tcl.numpy())
    conf, pc, tc = np.concatenate(confs), np.concatenate(pcls), np.concatenate(tcls)
    assert len(conf) > 2
    assert np.allclose(box, M.val_summary(M.ap_per_class(np.concatenate(tps), conf, pc, tc)), atol=1e-6)
    assert np.allclose(pose, M.val_summary(M.ap_per_class(np.concatenate(tpps), conf, pc, tc)), atol=1e-6)
    res = pdt.ImagePredict((d["images"][0] * 255).astype(np.uint8), 0.5, 0.5)
    want, _ = O.non_max_suppression(rinf["boxes"][:1].clone(), 0.5, 0.5, nc=nc) if hasattr(O, "non_max_suppression") else (None, None)
    assert len(res) > 0 and all(len(r.KeyPoints) == K for r in res)
    if want is not None:
        assert len(res) == len(want[0])
        for r, w in zip(res, want[0].numpy()):
            assert abs(r.KeyPoints[3].X - w[6 + 9]) < 1e-2 and abs(r.KeyPoints[3].VisibilityScore - w[6 + 11]) < 1e-3
    m.close()
