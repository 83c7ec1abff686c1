"""Independent known-answer tests (the oracle was parity-unpinned).

tests/kat_ref.py is a SECOND restatement of the cited C# -- scalar fp64 Python, explicit loops, finite-difference gradients --
that shares nothing with oracle/yolo_oracle.py (no torch, no autograd).  tests/golden/kat_loss.json holds its answers for a
hand-built case (tests/golden/make_kat.py).  Here both the ATen oracle AND the HIP kernels are checked against those answers:
CIoU, DFL incl. both clamps, TAL (double-claimed anchor, zero-metric ties, tiny-box inflation, padded GT row), the three loss
items, d(loss)/d(boxes, scores), and BatchNorm's batch / running statistics (momentum 0.03, unbiased running_var).
tests/golden/kat_tasks.json (make_kat_tasks.py) does the same for v8PoseLoss, v8OBBLoss and the mask term of v8SegmentationLoss: items and finite-difference gradients
w.r.t. the keypoint outputs, the box / class logits and the angle logit."""
import json
import os

import numpy as np
import pytest
import torch

import kat_ref as K
from conftest import BACKENDS
from oracle import yolo_oracle as O

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_loss.json")))


def _case():
    k = KAT
    return (np.array(k["boxes"], np.float64), np.array(k["scores"], np.float64),
            {"batch_idx": np.array(k["batch_idx"], np.float32), "cls": np.array(k["cls"], np.float32),
             "bboxes": np.array(k["bboxes"], np.float32)})


def test_kat_file_reproduces():
    """The committed answers are what kat_ref computes today (the fixture did not drift from its generator)."""
    k = KAT
    items, total, tg = K.detection_loss(k["boxes"], k["scores"], k["batch_idx"], k["cls"], k["bboxes"], k["H"], k["W"], k["nc"])
    assert np.allclose(items, k["items"], rtol=1e-12) and abs(total - k["total"]) < 1e-9
    assert [[bool(v) for v in t[0]] for t in tg] == k["fg"] and [t[1] for t in tg] == k["gt_idx"]
    for (b1, b2, want) in k["ciou_pairs"]:
        assert abs(K.ciou(b1, b2) - want) < 1e-14
    # closed-form spot checks of the independent restatement itself
    assert abs(K.ciou((1, 1, 3, 3), (1, 1, 3, 3)) - (4.0 / (4.0 + 1e-7))) < 1e-12            # identical boxes: iou = 4/(4+eps), no penalty
    iou = 9.0 / (16 + 16 - 9 + 1e-7)                                                          # (0,0,4,4) vs (1,1,5,5): v = 0
    assert abs(K.ciou((0, 0, 4, 4), (1, 1, 5, 5)) - (iou - 2.0 / (50 + 1e-7))) < 1e-12
    lg = k["dfl_logits"]
    ls = K.log_softmax(lg)
    assert abs(K.dfl(lg, 7.3) - (-(ls[7] * 0.7 + ls[8] * 0.3))) < 1e-9
    assert abs(K.dfl(lg, 20.0) - K.dfl(lg, 14.99)) < 1e-15 and abs(K.dfl(lg, 14.99) - (-(ls[14] * 0.01 + ls[15] * 0.99))) < 1e-9
    assert abs(K.dfl(lg, 0.0) + ls[0]) < 1e-12


def test_oracle_matches_kat():
    """ATen-CPU oracle vs the independent fp64 answers: scalar CIoU / DFL, assignment, loss items, autograd vs finite differences."""
    k = KAT
    for (b1, b2, want) in k["ciou_pairs"]:
        got = float(O.bbox_iou_ciou(torch.tensor([b1], dtype=torch.float64), torch.tensor([b2], dtype=torch.float64)))
        assert abs(got - want) < 1e-9, (b1, b2, got, want)
    lg = torch.tensor(k["dfl_logits"], dtype=torch.float64)
    for t, want in k["dfl_cases"]:
        tt = torch.tensor([t], dtype=torch.float64).clamp(0, 16 - 1 - 0.01)                  # Loss.cs:108
        tl = tt.long(); wl = (tl + 1) - tt
        got = float(torch.nn.functional.cross_entropy(lg[None], tl) * wl + torch.nn.functional.cross_entropy(lg[None], tl + 1) * (1 - wl))
        assert abs(got - want) < 1e-9, (t, got, want)
    bx, sc, batch = _case()
    B, nc = k["B"], k["nc"]
    for dt, tol in ((torch.float64, 1e-7), (torch.float32, 2e-4)):
        boxes = torch.tensor(bx, dtype=dt, requires_grad=True)
        scores = torch.tensor(sc, dtype=dt, requires_grad=True)
        feats = [torch.zeros(B, 1, k["H"] // s, k["W"] // s, dtype=dt) for s in (8, 16, 32)]
        tb = {kk: torch.from_numpy(v) for kk, v in batch.items()}
        loss, items, tg = O.v8DetectionLoss(nc)({"boxes": boxes, "scores": scores, "feats": feats}, tb, return_targets=True)
        assert np.allclose(items.detach().numpy(), k["items"], rtol=tol, atol=tol), (dt, items, k["items"])
        assert tg["fg_mask"].tolist() == k["fg"]
        fg = np.array(k["fg"])
        assert np.array_equal(tg["target_gt_idx"].numpy()[fg], np.array(k["gt_idx"])[fg])
        assert np.allclose(tg["target_scores"].detach().numpy(), np.array(k["target_scores"]), rtol=tol * 10, atol=tol)
        loss.sum().backward()
        for got, want in ((boxes.grad, k["dboxes"]), (scores.grad, k["dscores"])):
            want = np.array(want)
            assert np.abs(got.numpy() - want).max() <= max(tol * 50, 2e-5) * np.abs(want).max(), dt


@pytest.mark.parametrize("backend", BACKENDS)
def test_engine_loss_matches_kat(backend, engine):
    """HIP v8DetectionLoss on caller-supplied preds (ys_model_set_preds) vs the independent answers: items within 1e-3
    (north star), d(sum(loss)*B)/d(boxes), d(scores) within 1e-3 of the gradient's max."""
    from yolosharp_amd.model import Yolov8, v8DetectionLoss
    k = KAT
    bx, sc, batch = _case()
    m = Yolov8(engine, nc=k["nc"], size="n", height=k["H"], width=k["W"], max_batch=k["B"], dtype="f32")
    assert m.A == bx.shape[2]
    m.set_preds({"boxes": bx, "scores": sc})
    loss, items = v8DetectionLoss(m)(None, batch)
    assert np.allclose(items, k["items"], rtol=1e-3, atol=1e-5), (items, k["items"])
    assert np.allclose(loss.sum(), k["total"], rtol=1e-3)
    for key, want in (("dboxes", k["dboxes"]), ("dscores", k["dscores"])):
        want = np.array(want)
        got = m.get_output(key)
        assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max(), (key, np.abs(got - want).max(), np.abs(want).max())
    from yolosharp_amd import YsError
    with pytest.raises(YsError):
        m.backward()                                       # no graph state behind caller-supplied preds
    m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_conv_bn_silu_matches_kat(backend, engine):
    """Conv unit (1x1 conv + BatchNorm(eps 1e-3, momentum 0.03) + SiLU, Convs.cs:36-62) in training mode: outputs, running_mean
    and the UNBIASED running_var against the fp64 loops; same check for the oracle module."""
    from yolosharp_amd import blocks
    bn = KAT["bn"]
    Cc, B, Hh, Ww = 4, 2, 4, 4
    x = np.array(bn["x"], np.float32).reshape(B, Hh, Ww, Cc).transpose(0, 3, 1, 2).copy()     # NHWC rows -> NCHW
    want = np.array(bn["out"]).reshape(B, Hh, Ww, Cc).transpose(0, 3, 1, 2)
    sd = {"conv.weight": np.array(bn["w"], np.float32).reshape(Cc, Cc, 1, 1), "bn.weight": np.array(bn["gamma"], np.float32),
          "bn.bias": np.array(bn["beta"], np.float32), "bn.running_mean": np.array(bn["running_mean"], np.float32),
          "bn.running_var": np.array(bn["running_var"], np.float32), "bn.num_batches_tracked": np.zeros(1, np.float32)}
    ref = O.Conv(Cc, Cc, 1, 1).train()
    ref.load_state_dict({kk: torch.from_numpy(v).reshape(()).long() if "num_batches" in kk else torch.from_numpy(v) for kk, v in sd.items()})
    ry = ref(torch.from_numpy(x)).detach().numpy()
    assert np.abs(ry - want).max() < 1e-5
    assert np.allclose(ref.bn.running_mean.numpy(), bn["new_running_mean"], rtol=1e-5, atol=1e-6)
    assert np.allclose(ref.bn.running_var.numpy(), bn["new_running_var"], rtol=1e-5)
    blk = blocks.Conv(engine, Cc, Cc, 1, 1, True, height=Hh, width=Ww, max_batch=B, dtype="f32")
    blk.load_state_dict(sd)
    blk.train()
    y = blk.forward(x)
    assert np.abs(y - want).max() < 1e-4 * max(1.0, np.abs(want).max())
    after = blk.state_dict()
    assert np.allclose(after["bn.running_mean"], bn["new_running_mean"], rtol=1e-4, atol=1e-6)
    assert np.allclose(after["bn.running_var"], bn["new_running_var"], rtol=1e-4)
    assert after["bn.num_batches_tracked"][0] == 1
    blk.close()


# ----------------------------------------------------------------------------- v8PoseLoss / v8OBBLoss known answers
TASKS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_tasks.json")))


def test_task_kat_file_reproduces():
    """kat_tasks.json is what kat_ref computes today (tests/golden/make_kat_tasks.py), plus closed-form spot checks of probiou."""
    t, k = TASKS, KAT
    p = t["pose"]
    items, total, _ = K.pose_loss(k["boxes"], k["scores"], p["kpts"], k["batch_idx"], k["cls"], k["bboxes"], p["keypoints"], t["H"], t["W"],
                                  p["nc"], p["K"], p["D"])
    assert np.allclose(items, p["items"], rtol=1e-12) and abs(total - p["total"]) < 1e-9
    o = t["obb"]
    items, total, tg = K.obb_loss(o["boxes"], o["scores"], o["angle_logit"], o["batch_idx"], o["cls"], o["bboxes"], t["H"], t["W"], o["nc"])
    assert np.allclose(items, o["items"], rtol=1e-12) and abs(total - o["total"]) < 1e-9
    assert [[bool(v) for v in x[0]] for x in tg] == o["fg"] and [x[1] for x in tg] == o["gt_idx"]
    g = t["segment"]
    _, _, dtg = K.detection_loss(k["boxes"], k["scores"], k["batch_idx"], k["cls"], k["bboxes"], t["H"], t["W"], k["nc"])
    assert abs(K.seg_term(g["coeff"], g["proto"], dtg, g["masks"], t["H"], t["W"]) - g["item"]) < 1e-9
    assert abs(K.seg_term(g["coeff"], g["proto"], dtg, g["masks"], t["H"], t["W"], trunc_crop=True) - g["item_trunc"]) < 1e-9
    same = K.probiou((3.0, 4.0, 6.0, 2.0, 0.3), (3.0, 4.0, 6.0, 2.0, 0.3))        # identical boxes: bd = 0.5 log(1 + eps') -> clamp(eps)
    assert abs(same - (1.0 - np.sqrt(1.0 - np.exp(-1e-7) + 1e-7))) < 1e-6
    assert abs(K.probiou((0, 0, 4, 2, 0.0), (1, 0, 4, 2, 0.0)) - K.probiou((0, 0, 2, 4, np.pi / 2), (1, 0, 2, 4, np.pi / 2))) < 1e-12


def _pose_case():
    k, p = KAT, TASKS["pose"]
    bx, sc, batch = _case()
    batch["keypoints"] = np.array(p["keypoints"], np.float32)
    return bx, sc, np.array(p["kpts"], np.float64), batch


def _obb_case():
    o = TASKS["obb"]
    batch = {"batch_idx": np.array(o["batch_idx"], np.float32), "cls": np.array(o["cls"], np.float32), "bboxes": np.array(o["bboxes"], np.float32)}
    return np.array(o["boxes"], np.float64), np.array(o["scores"], np.float64), np.array(o["angle_logit"], np.float64), batch


def test_oracle_matches_task_kat():
    """ATen-CPU v8PoseLoss / v8OBBLoss (+ RotatedTaskAlignedAssigner) vs the independent fp64 answers: items, autograd vs finite differences."""
    t = TASKS
    B, H, W = t["B"], t["H"], t["W"]
    for dt, tol in ((torch.float64, 1e-7), (torch.float32, 2e-4)):
        feats = [torch.zeros(B, 1, H // s, W // s, dtype=dt) for s in (8, 16, 32)]
        # pose
        p = t["pose"]
        bx, sc, kp, batch = _pose_case()
        kpts = torch.tensor(kp, dtype=dt, requires_grad=True)
        preds = {"boxes": torch.tensor(bx, dtype=dt), "scores": torch.tensor(sc, dtype=dt), "kpts": kpts, "feats": feats}
        loss, items = O.v8PoseLoss(p["nc"], p["K"], p["D"])(preds, {kk: torch.from_numpy(v) for kk, v in batch.items()})
        assert np.allclose(items.detach().numpy(), p["items"], rtol=tol, atol=tol), (dt, items, p["items"])
        loss.sum().backward()
        want = np.array(p["dkpts"])
        assert np.abs(kpts.grad.numpy() - want).max() <= max(tol * 50, 2e-5) * np.abs(want).max(), dt
        # obb
        o = t["obb"]
        bx, sc, al, batch = _obb_case()
        boxes = torch.tensor(bx, dtype=dt, requires_grad=True)
        scores = torch.tensor(sc, dtype=dt, requires_grad=True)
        logit = torch.tensor(al, dtype=dt, requires_grad=True)
        preds = {"boxes": boxes, "scores": scores, "angle": (logit.sigmoid() - 0.25) * np.pi, "feats": feats}
        loss, items = O.v8OBBLoss(o["nc"])(preds, {kk: torch.from_numpy(v) for kk, v in batch.items()})
        assert np.allclose(items.detach().numpy(), o["items"], rtol=tol, atol=tol), (dt, items, o["items"])
        loss.sum().backward()
        for got, key in ((boxes.grad, "dboxes"), (scores.grad, "dscores"), (logit.grad, "dangle")):
            want = np.array(o[key])
            assert np.abs(got.numpy() - want).max() <= max(tol * 50, 2e-5) * np.abs(want).max(), (dt, key)


@pytest.mark.parametrize("backend", BACKENDS)
def test_engine_task_losses_match_kat(backend, engine):
    """HIP v8PoseLoss / v8OBBLoss on caller-supplied preds vs the independent answers: items within 1e-3, gradients within 1e-3 of
    their maximum (the OBB case holds a filtered label, a widened label, nested boxes and a padded GT row)."""
    from yolosharp_amd.model import Yolov8Obb, Yolov8Pose, v8OBBLoss, v8PoseLoss
    t = TASKS
    p = t["pose"]
    bx, sc, kp, batch = _pose_case()
    m = Yolov8Pose(engine, nc=p["nc"], size="n", height=t["H"], width=t["W"], max_batch=t["B"], dtype="f32", kpt_num=p["K"], kpt_dim=p["D"])
    m.set_preds({"boxes": bx, "scores": sc, "kpts": kp})
    loss, items = v8PoseLoss(m)(None, batch)
    assert np.allclose(items, p["items"], rtol=1e-3, atol=1e-5), (items, p["items"])
    assert np.allclose(loss.sum(), p["total"], rtol=1e-3)
    want = np.array(p["dkpts"])
    assert np.abs(m.get_output("dkpts") - want).max() <= 1e-3 * np.abs(want).max()
    m.close()
    o = t["obb"]
    bx, sc, al, batch = _obb_case()
    m = Yolov8Obb(engine, nc=o["nc"], size="n", height=t["H"], width=t["W"], max_batch=t["B"], dtype="f32")
    m.set_preds({"boxes": bx, "scores": sc, "angle": (1.0 / (1.0 + np.exp(-al)) - 0.25) * np.pi})
    loss, items = v8OBBLoss(m)(None, batch)
    assert np.allclose(items, o["items"], rtol=1e-3, atol=1e-5), (items, o["items"])
    assert np.allclose(loss.sum(), o["total"], rtol=1e-3)
    for key in ("dboxes", "dscores", "dangle"):
        want = np.array(o[key])
        got = m.get_output(key)
        assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max(), (key, np.abs(got - want).max(), np.abs(want).max())
    m.close()


def _seg_case():
    g = TASKS["segment"]
    bx, sc, batch = _case()
    batch["masks"] = np.array(g["masks"], np.float32)
    return bx, sc, np.array(g["coeff"], np.float64), np.array(g["proto"], np.float64), batch


def test_oracle_matches_segment_kat():
    """ATen-CPU v8SegmentationLoss mask term vs the independent fp64 answers: both Ops.crop_mask branches, autograd vs finite differences."""
    t, g, k = TASKS, TASKS["segment"], KAT
    B, H, W = t["B"], t["H"], t["W"]
    bx, sc, cf, pr, batch = _seg_case()
    for dt, tol in ((torch.float64, 1e-7), (torch.float32, 2e-4)):
        feats = [torch.zeros(B, 1, H // s, W // s, dtype=dt) for s in (8, 16, 32)]
        for branch, want in ((False, g["item"]), (True, g["item_trunc"])):
            coeff = torch.tensor(cf, dtype=dt, requires_grad=True)
            proto = torch.tensor(pr, dtype=dt, requires_grad=True)
            preds = {"boxes": torch.tensor(bx, dtype=dt), "scores": torch.tensor(sc, dtype=dt), "mask_coefficient": coeff, "proto": proto, "feats": feats}
            loss, items = O.v8SegmentationLoss(k["nc"], cpu_crop_branch=branch)(preds, {kk: torch.from_numpy(v) for kk, v in batch.items()})
            assert abs(float(items[1]) - want) <= max(tol, tol * abs(want)) * 10, (dt, branch, float(items[1]), want)
            assert np.allclose(items.detach().numpy()[[0, 2, 3]], k["items"], rtol=tol, atol=tol)      # box, cls, dfl
        loss.sum().backward()   # (the CPU-branch run is the last one: its crop only changes which pixels count)
        coeff2 = torch.tensor(cf, dtype=dt, requires_grad=True); proto2 = torch.tensor(pr, dtype=dt, requires_grad=True)
        preds = {"boxes": torch.tensor(bx, dtype=dt), "scores": torch.tensor(sc, dtype=dt), "mask_coefficient": coeff2, "proto": proto2, "feats": feats}
        O.v8SegmentationLoss(k["nc"])(preds, {kk: torch.from_numpy(v) for kk, v in batch.items()})[0].sum().backward()
        for got, key in ((coeff2.grad, "dcoeff"), (proto2.grad, "dproto")):
            want = np.array(g[key])
            assert np.abs(got.numpy() - want).max() <= max(tol * 50, 2e-5) * np.abs(want).max(), (dt, key)


@pytest.mark.parametrize("backend", BACKENDS)
def test_engine_segment_loss_matches_kat(backend, engine):
    """HIP v8SegmentationLoss on caller-supplied preds vs the independent answers: the mask item in both crop modes, d(coeff), d(proto)."""
    from yolosharp_amd.model import Yolov8Segment, v8SegmentationLoss
    t, g, k = TASKS, TASKS["segment"], KAT
    bx, sc, cf, pr, batch = _seg_case()
    m = Yolov8Segment(engine, nc=k["nc"], size="n", height=t["H"], width=t["W"], max_batch=t["B"], dtype="f32")
    for branch, want in ((True, g["item_trunc"]), (False, g["item"])):
        m.set_preds({"boxes": bx, "scores": sc, "mask_coefficient": cf, "proto": pr})
        loss, items = v8SegmentationLoss(m, cpu_crop_branch=branch)(None, batch)
        assert abs(items[1] - want) <= 1e-3 * abs(want), (branch, items, want)
        assert np.allclose(items[[0, 2, 3]], k["items"], rtol=1e-3, atol=1e-5) and items[4] == 0
    for key, wk in (("dmask_coefficient", "dcoeff"), ("dproto", "dproto")):
        want = np.array(g[wk])
        got = m.get_output(key)
        assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max(), (key, np.abs(got - want).max(), np.abs(want).max())
    m.close()


def test_oracle_adamw_matches_torch_optimizer():
    """The oracle's AdamW restatement (used to check ys_optim_adamw_step) against PyTorch's own torch.optim.AdamW -- the optimizer
    semantics TorchSharp mirrors (decoupled decay, bias corrections, eps outside the corrected sqrt): three steps, three groups."""
    g = torch.Generator().manual_seed(0)
    names = ["model.0.conv.weight", "model.0.bn.weight", "model.0.bn.bias", "model.22.cv2.0.2.bias"]
    params = {n: torch.randn(7, 5, generator=g, dtype=torch.float64) for n in names}
    ref = {n: torch.nn.Parameter(p.clone()) for n, p in params.items()}
    lrs = [3e-3, 1e-3, 2e-3]                                      # groups: bias, conv weight, bn weight
    opt = torch.optim.AdamW([{"params": [ref[n]], "lr": lrs[O.param_group_of(n)]} for n in names], betas=(0.9, 0.999), eps=1e-8,
                            weight_decay=5e-4)
    state = {}
    for step in range(1, 4):
        grads = {n: torch.randn(7, 5, generator=g, dtype=torch.float64) * (0.1 if "bn" in n else 1.0) for n in names}
        for n in names:
            ref[n].grad = grads[n].clone()
        opt.step()
        O.adamw_step(params, grads, state, lrs, step=step)
        for n in names:
            assert torch.allclose(params[n], ref[n].detach(), rtol=1e-12, atol=1e-14), (step, n)


@pytest.mark.parametrize("backend", BACKENDS)
def test_non_finite_inputs_poison_the_integer_accumulators(backend, engine):
    """Round 6: the criterion's sums and the forward BatchNorm statistics are fixed-point INTEGER accumulators; a NaN / Inf partial has no integer image (an
    unguarded fptosi is a garbage finite number that would be folded into the loss / into run_mean and run_var).  Both raise a poison word instead and the consumers turn it
    into NaN -- what the reference's float sums do on divergence.  (a) a NaN class logit -> NaN loss items; (b) an Inf activation in a training-mode Conv unit -> non-finite
    running statistics for that layer, finite ones for a clean rerun."""
    from yolosharp_amd.model import Yolov8, v8DetectionLoss
    k = KAT
    bx, sc, batch = _case()
    m = Yolov8(engine, nc=k["nc"], size="n", height=k["H"], width=k["W"], max_batch=k["B"], dtype="f32")
    sc2 = sc.copy(); sc2[0, 1, 3] = np.nan
    m.set_preds({"boxes": bx, "scores": sc2})
    _, items = v8DetectionLoss(m)(None, batch)
    assert np.isnan(items[1]), items                       # the class item (and the total) carry the NaN; nothing is silently finite
    m.set_preds({"boxes": bx, "scores": sc})
    _, items = v8DetectionLoss(m)(None, batch)             # the accumulators are cleared per call: a clean step after a poisoned one is clean
    assert np.allclose(items, k["items"], rtol=1e-3, atol=1e-5), items
    m.close()
    # (b) BatchNorm statistics of a whole model forward (BN_ATOMIC path): an Inf in the image
    B, H, W = 2, 64, 64
    m = Yolov8(engine, nc=80, size="n", height=H, width=W, max_batch=B, dtype="bf16")
    m.init_weights(3); m.train()
    x = np.random.default_rng(0).random((B, 3, H, W), dtype=np.float32)
    xi = x.copy(); xi[0, 1, 5, 7] = np.inf
    m.forward(xi, fetch=False)
    sd = m.state_dict()
    assert not np.isfinite(sd["model.0.bn.running_mean"]).all() or not np.isfinite(sd["model.0.bn.running_var"]).all()
    m.close()
    m = Yolov8(engine, nc=80, size="n", height=H, width=W, max_batch=B, dtype="bf16")
    m.init_weights(3); m.train(); m.forward(x, fetch=False)
    sd = m.state_dict()
    assert all(np.isfinite(v).all() for kk, v in sd.items() if "running" in kk)
    m.close()
