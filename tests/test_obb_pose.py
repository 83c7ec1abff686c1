"""Obb / Pose heads: the cv4 towers, dist2rbox / kpts_decode and the rotated predict path.
Oracle = oracle/yolo_oracle.py (Head.cs:376-606, Tal.cs:389-408, Loss.cs:486-684,870-1071, Tal.cs:260-310).  fp32 tolerance 1e-3.  Forward /
predict plus the criteria v8OBBLoss / v8PoseLoss and the full backward."""
import numpy as np
import pytest
import torch

from conftest import BACKENDS
from oracle import yolo_oracle as O
from test_model import relerr


def make_ref(cls, nc, size, seed=0, **kw):
    torch.manual_seed(seed)
    ref = cls(nc=nc, size=size, **kw)
    for mod in ref.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.normal_(0, 0.1)
            mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
    return ref


def _load(engine, ref, cls, nc, size, B, H, W, dtype="f32", **kw):
    m = cls(engine, nc=nc, size=size, height=H, width=W, max_batch=B, dtype=dtype, **kw)
    info = m.tensor_info()
    assert [n for n, s, p in info if p] == [k for k, _ in ref.named_parameters()]              # cv2.*, cv3.*, dfl, cv4.*
    assert {n: tuple(s) for n, s, p in info} == {k: (tuple(v.shape) if v.dim() else (1,)) for k, v in ref.state_dict().items()}
    assert m.num_params() == sum(p.numel() for k, p in ref.named_parameters() if "dfl" not in k)   # flat trainable length, pad rows excluded
    sd = {k: v.detach().numpy() for k, v in ref.state_dict().items()}
    m.load_state_dict(sd)
    back = m.state_dict()
    assert all(np.array_equal(back[k].reshape(-1), sd[k].reshape(-1)) for k in sd)                                   # padded towers expose the reference rows
    return m


def _head_parity(engine, task, family, size, B, H, W, tol, nc=5, **kw):
    from yolosharp_amd import model as M
    name = f"Yolov{family}{task}"
    ref = make_ref(getattr(O, name), nc, size, **kw)
    m = _load(engine, ref, getattr(M, name), nc, size, B, H, W, **kw)
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(3))
    extra = "angle" if task == "Obb" else "kpts"
    # ---- eval: pred = cat(decoded boxes, class probabilities, angle | decoded keypoints)
    m.eval(); ref.eval()
    inf, preds = m.forward(x.numpy())
    with torch.no_grad():
        rinf, rpreds = ref(x)
    for key in ("boxes", "scores", extra):
        assert preds[key].shape == tuple(rpreds[key].shape)
        assert relerr(preds[key], rpreds[key]) < tol, key
    assert inf["boxes"].shape == tuple(rinf["boxes"].shape)
    r = rinf["boxes"].numpy()
    assert relerr(inf["boxes"][:, :4], r[:, :4]) < tol
    assert relerr(inf["boxes"][:, 4:4 + nc], r[:, 4:4 + nc]) < tol
    assert relerr(inf["boxes"][:, 4 + nc:], r[:, 4 + nc:]) < tol
    if task == "Obb":                                              # angle range of Head.cs:429
        a = inf["boxes"][:, 4 + nc]
        assert a.min() >= -np.pi / 4 - 1e-6 and a.max() <= 3 * np.pi / 4 + 1e-6
    # ---- train mode: batch statistics through the (padded) towers; the head returns preds only (Head.cs:103-106)
    m.train(); ref.train()
    inf, preds = m.forward(x.numpy())
    _, rpreds = ref(x)
    assert inf is None
    for key in ("boxes", "scores", extra):
        assert relerr(preds[key], rpreds[key].detach()) < tol, key
    rs, ms = ref.state_dict(), m.state_dict()                     # running statistics of every BatchNorm, incl. the cv4 towers
    for k in rs:
        if "running" in k:
            assert np.allclose(ms[k], rs[k].numpy(), rtol=1e-3, atol=1e-5), k
    return m, ref, x


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("task", ["Obb", "Pose"])
def test_yolov8n_head_f32(backend, engine, task):
    m, _, _ = _head_parity(engine, task, 8, "n", 2, 64, 64, 1e-3)      # Pose n: tower width 51 (padded to the 16-byte unit inside)
    m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_yolov8n_pose_two_dim_keypoints(backend, engine):
    m, _, _ = _head_parity(engine, "Pose", 8, "n", 2, 64, 96, 1e-3, nc=1, kpt_num=5, kpt_dim=2)
    m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_detection_criterion_is_refused(backend, engine):
    """OBB / Pose models never train on the plain detection criterion: it is not their loss (axis-aligned assigner, no angle /
    keypoint terms)."""
    from yolosharp_amd import YsError
    from yolosharp_amd.model import Yolov8Obb, Yolov8Pose, v8DetectionLoss
    batch = {"batch_idx": np.zeros(1, np.float32), "cls": np.zeros(1, np.float32), "bboxes": np.array([[0.5, 0.5, 0.2, 0.2]], np.float32)}
    for cls, msg in ((Yolov8Obb, "ys_loss_obb"), (Yolov8Pose, "ys_loss_pose")):
        m = cls(engine, nc=3, size="n", height=32, width=32, max_batch=1, dtype="f32")
        m.init_weights(1)
        m.forward(np.zeros((1, 3, 32, 32), np.float32), fetch=False)
        with pytest.raises(YsError, match=msg):
            v8DetectionLoss(m)(None, batch)
        m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gpu"])
@pytest.mark.parametrize("task,family,size", [("Obb", 8, "s"), ("Obb", 11, "s"), ("Pose", 8, "s"), ("Pose", 11, "m")])
def test_head_full_resolution_f32(backend, engine, task, family, size):
    m, _, _ = _head_parity(engine, task, family, size, 2, 640, 640, 1e-3, nc=15 if task == "Obb" else 1)
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gpu"])
def test_other_sizes_f32(backend, engine):
    """Pose l: cv4 towers are 64 wide (no padding branch); Obb m: c4 = 48."""
    for task, size in (("Pose", "l"), ("Obb", "m")):
        m, _, _ = _head_parity(engine, task, 8, size, 2, 320, 320, 1e-3, nc=2)
        m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gpu"])
@pytest.mark.parametrize("task", ["Obb", "Pose"])
def test_head_bf16_tracks_f32(backend, engine, task):
    from yolosharp_amd import model as M
    nc, B, H, W = 4, 2, 320, 320
    ref = make_ref(getattr(O, f"Yolov8{task}"), nc, "s")
    m = _load(engine, ref, getattr(M, f"Yolov8{task}"), nc, "s", B, H, W, dtype="bf16")
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(5))
    m.eval(); ref.eval()
    inf, preds = m.forward(x.numpy())
    with torch.no_grad():
        rinf, rpreds = ref(x)
    for key in ("boxes", "scores", "angle" if task == "Obb" else "kpts"):
        assert relerr(preds[key], rpreds[key]) < 5e-2, key
    assert relerr(inf["boxes"][:, :4], rinf["boxes"][:, :4]) < 5e-2
    m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_obb_predict_rotated_nms(backend, engine):
    """Predict path of an OBB model (Predictor.cs / Ops.cs:308-370 rotated branch): eval forward -> rotated NMS, rows compared
    with the oracle's NMS over the ORACLE's forward (so the decode, the layout and the selection are all checked)."""
    from yolosharp_amd.model import Yolov8Obb
    nc, B, H, W = 3, 2, 64, 64
    ref = make_ref(O.Yolov8Obb, nc, "n", seed=2)
    with torch.no_grad():                                          # give the class logits a spread so that some anchors pass conf
        for seq in ref.model[-1].cv3:
            seq[2].bias.add_(1.0); seq[2].weight.mul_(8.0)
    m = _load(engine, ref, Yolov8Obb, nc, "n", B, H, W)
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(7))
    m.eval(); ref.eval()
    inf, _ = m.forward(x.numpy())
    with torch.no_grad():
        rinf, _ = ref(x)
    got, gkeep = engine.non_max_suppression(inf["boxes"], conf_thres=0.5, iou_thres=0.3, nc=nc, rotated=True)
    want, wkeep = O.non_max_suppression_rotated(rinf["boxes"], conf_thres=0.5, iou_thres=0.3, nc=nc)
    assert sum(len(w) for w in want) > 4
    for g, w, gk, wk in zip(got, want, gkeep, wkeep):
        assert np.array_equal(gk, wk.numpy())
        assert g.shape == tuple(w.shape) and np.allclose(g, w.numpy(), rtol=1e-3, atol=1e-3)
    m.close()


# ----------------------------------------------------------------------------- v8PoseLoss (Loss.cs:870-1071) + backward
def _pose_train_parity(engine, family, size, B, H, W, tol, tol_grad, kpt_num=17, kpt_dim=3, nc=1, kmax=6):
    from yolosharp_amd import model as M
    name = f"Yolov{family}Pose"
    ref = make_ref(getattr(O, name), nc, size, kpt_num=kpt_num, kpt_dim=kpt_dim)
    m = _load(engine, ref, getattr(M, name), nc, size, B, H, W, kpt_num=kpt_num, kpt_dim=kpt_dim)
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(3))
    batch = O.synthetic_batch(B, H, W, nc, seed=1, kmax=kmax)
    batch["keypoints"] = O.synthetic_keypoints(batch, kpt_num, kpt_dim)
    nb = {k: v.numpy() for k, v in batch.items()}
    m.train(); ref.train()
    _, preds = m.forward(x.numpy())
    _, rpreds = ref(x)
    assert relerr(preds["kpts"], rpreds["kpts"].detach()) < tol
    rpreds["kpts"].retain_grad()
    loss, items = M.v8PoseLoss(m)(None, nb)
    rloss, ritems = O.v8PoseLoss(nc, kpt_num, kpt_dim)(rpreds, batch)
    assert items.shape == (5,) and float(ritems[1]) > 0 and (kpt_dim == 2 or float(ritems[2]) > 0)
    assert np.allclose(items, ritems.numpy(), rtol=1e-3, atol=1e-5), (items, ritems)
    assert np.allclose(loss, rloss.detach().numpy(), rtol=1e-3, atol=1e-4)
    rloss.sum().backward()
    r = rpreds["kpts"].grad.numpy()
    g = m.get_output("dkpts")
    assert np.abs(r).max() > 0 and np.abs(g - r).max() <= tol_grad * np.abs(r).max()
    m.zero_grad(); m.backward()
    grads = m.grads()
    gscale = max(float(p.grad.abs().max()) for _, p in ref.named_parameters() if p.grad is not None)
    for pname, p in ref.named_parameters():
        if p.grad is None:
            continue
        r = p.grad.numpy()
        assert np.abs(grads[pname] - r).max() <= tol_grad * np.abs(r).max() + 1e-6 * gscale, pname
    return m, ref


@pytest.mark.parametrize("backend", BACKENDS)
def test_yolov8n_pose_loss_backward_f32(backend, engine):
    """COCO layout (17 x 3, OKS sigmas); the n model's cv4 towers are 51 wide -> padded inside, gradients listed for the 51 rows."""
    m, _ = _pose_train_parity(engine, 8, "n", 2, 64, 64, 1e-3, 2e-3)
    m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_pose_loss_two_dim_keypoints_from_preds(backend, engine):
    """kpt_dim = 2 (no visibility term, sigmas = 1/K, Loss.cs:905,1051,1062) on caller-supplied preds (Loss.cs:411)."""
    from yolosharp_amd.model import Yolov8Pose, v8PoseLoss
    nc, K, D, B, H, W = 2, 5, 2, 2, 64, 64
    m = Yolov8Pose(engine, nc=nc, size="n", height=H, width=W, max_batch=B, dtype="f32", kpt_num=K, kpt_dim=D)
    A = m.A
    g = torch.Generator().manual_seed(11)
    rp = {"boxes": torch.randn(B, 64, A, generator=g), "scores": torch.randn(B, nc, A, generator=g) - 2.0,
          "kpts": (torch.randn(B, K * D, A, generator=g) * 0.5).requires_grad_(True),
          "feats": [torch.zeros(B, 1, H // s, W // s) for s in (8, 16, 32)]}
    batch = O.synthetic_batch(B, H, W, nc, seed=4, kmax=5)
    batch["keypoints"] = O.synthetic_keypoints(batch, K, D)
    m.set_preds({k: v.detach().numpy() for k, v in rp.items() if k != "feats"})
    loss, items = v8PoseLoss(m)(None, {k: v.numpy() for k, v in batch.items()})
    rloss, ritems = O.v8PoseLoss(nc, K, D)(rp, batch)
    assert float(ritems[1]) > 0 and float(ritems[2]) == 0 and items[2] == 0
    assert np.allclose(items, ritems.numpy(), rtol=1e-3, atol=1e-5), (items, ritems)
    rloss.sum().backward()
    r = rp["kpts"].grad.numpy()
    assert np.abs(m.get_output("dkpts") - r).max() <= 1e-3 * np.abs(r).max()
    m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_pose_loss_refuses_labels_out_of_collate_order(backend, engine):
    """Keypoint rows are addressed by a label's rank within its image (Loss.cs:1040-1071 `arange - offsets[batch_idx]`), which is
    only defined for labels grouped by image in collate order: host labels that are not are refused (round-2 advisor finding)."""
    from yolosharp_amd import YsError
    from yolosharp_amd.model import Yolov8Pose, v8PoseLoss
    m = Yolov8Pose(engine, nc=1, size="n", height=32, width=32, max_batch=2, dtype="f32")
    m.init_weights(3)
    m.forward(np.random.default_rng(0).random((2, 3, 32, 32), np.float32), fetch=False)
    bb = np.array([[0.5, 0.5, 0.4, 0.4]] * 3, np.float32)
    kp = np.random.default_rng(1).random((3, 17, 3)).astype(np.float32)
    good = {"batch_idx": np.array([0, 0, 1], np.float32), "cls": np.zeros(3, np.float32), "bboxes": bb, "keypoints": kp}
    _, items = v8PoseLoss(m)(None, good)
    assert np.isfinite(items).all()
    bad = dict(good, batch_idx=np.array([1, 0, 0], np.float32))
    with pytest.raises(YsError) as ei:
        v8PoseLoss(m)(None, bad)
    assert ei.value.status == 1 and "collate order" in str(ei.value)
    m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_pose_loss_without_labels(backend, engine):
    """No foreground anchor: pose = kobj = 0, zero keypoint gradients (the `fg_mask.sum() > 0` guard, Loss.cs:945)."""
    from yolosharp_amd.model import Yolov8Pose, v8PoseLoss
    m = Yolov8Pose(engine, nc=1, size="n", height=32, width=32, max_batch=1, dtype="f32")
    m.init_weights(3)
    m.forward(np.random.default_rng(0).random((1, 3, 32, 32), np.float32), fetch=False)
    e = np.zeros((0,), np.float32)
    loss, items = v8PoseLoss(m)(None, {"batch_idx": e, "cls": e, "bboxes": e.reshape(0, 4), "keypoints": np.zeros((0, 17, 3), np.float32)})
    assert items[1] == 0 and items[2] == 0 and items[3] > 0
    assert not m.get_output("dkpts").any()
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gpu"])
def test_yolov11s_pose_loss_backward_full_resolution_f32(backend, engine):
    # nc = 4: with a single class and random weights the class-logit gradient is nearly constant over the 20x20 maps, the BatchNorm
    # backward of model.7-9 cancels to ~1e-8 and fp32 summation order alone moves those tensors by ~5e-3 (tools/dev/pose_grad_diag.py)
    m, _ = _pose_train_parity(engine, 11, "s", 2, 640, 640, 1e-3, 2e-3, kmax=12, nc=4)
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gpu"])
def test_yolov8s_pose_bf16_train_steps(backend, engine):
    """bf16 training steps of a Pose model: finite items, keypoint terms of the first step near the fp32 oracle's, loss decreasing."""
    from yolosharp_amd.model import Yolov8Pose, v8PoseLoss, AMPWrapper
    nc, B, H, W = 1, 4, 320, 320
    ref = make_ref(O.Yolov8Pose, nc, "s")
    m = _load(engine, ref, Yolov8Pose, nc, "s", B, H, W, dtype="bf16")
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(3))
    batch = O.synthetic_batch(B, H, W, nc, seed=1, kmax=6)
    batch["keypoints"] = O.synthetic_keypoints(batch)
    nb = {k: v.numpy() for k, v in batch.items()}
    ref.train()
    _, rpreds = ref(x)
    _, ritems = O.v8PoseLoss(nc)(rpreds, batch)
    amp = AMPWrapper(m, lr=2e-3)
    crit = v8PoseLoss(m)
    hist = []
    for _ in range(6):
        loss, items = amp.TrainStep(x.numpy(), nb, crit)
        assert np.isfinite(items).all()
        hist.append(float(loss.sum()))
    first = hist[0] / B
    assert abs(first - float(ritems.sum())) < 0.1 * float(ritems.sum())
    assert hist[-1] < hist[0]
    m.close()


# ----------------------------------------------------------------------------- v8OBBLoss (Loss.cs:486-684) + backward
def _obb_train_parity(engine, family, size, B, H, W, tol, tol_grad, nc=6, kmax=6):
    from yolosharp_amd import model as M
    name = f"Yolov{family}Obb"
    ref = make_ref(getattr(O, name), nc, size)
    m = _load(engine, ref, getattr(M, name), nc, size, B, H, W)
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(3))
    batch = O.synthetic_obb_batch(B, H, W, nc, seed=1, kmax=kmax)
    nb = {k: v.numpy() for k, v in batch.items()}
    m.train(); ref.train()
    _, preds = m.forward(x.numpy())
    _, rpreds = ref(x)
    assert relerr(preds["angle"], rpreds["angle"].detach()) < tol
    for k in ("angle_raw", "boxes", "scores"):
        rpreds[k].retain_grad()
    loss, items = M.v8OBBLoss(m)(None, nb)
    rloss, ritems = O.v8OBBLoss(nc)(rpreds, batch)
    assert items.shape == (4,) and float(ritems[0]) > 0 and float(ritems[3]) > 0
    assert np.allclose(items, ritems.numpy(), rtol=1e-3, atol=1e-5), (items, ritems)
    assert np.allclose(loss, rloss.detach().numpy(), rtol=1e-3, atol=1e-4)
    rloss.sum().backward()
    for key, rk in (("dangle", "angle_raw"), ("dboxes", "boxes"), ("dscores", "scores")):
        r = rpreds[rk].grad.numpy()
        g = m.get_output(key)
        assert np.abs(r).max() > 0 and np.abs(g - r).max() <= tol_grad * np.abs(r).max(), key
    m.zero_grad(); m.backward()
    grads = m.grads()
    gscale = max(float(p.grad.abs().max()) for _, p in ref.named_parameters() if p.grad is not None)
    for pname, p in ref.named_parameters():
        if p.grad is None:
            continue
        r = p.grad.numpy()
        assert np.abs(grads[pname] - r).max() <= tol_grad * np.abs(r).max() + 1e-6 * gscale, pname
    return m


@pytest.mark.parametrize("backend", BACKENDS)
def test_yolov8n_obb_loss_backward_f32(backend, engine):
    m = _obb_train_parity(engine, 8, "n", 2, 64, 64, 1e-3, 2e-3)
    m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_obb_loss_from_preds_assignment_and_items(backend, engine):
    """Caller-supplied preds (Loss.cs:411) at 128 px with many oriented labels: thin boxes are filtered (< 2 px) or widened
    (< stride[0]); loss items and head gradients against the oracle."""
    from yolosharp_amd.model import Yolov8Obb, v8OBBLoss
    nc, B, H, W = 4, 2, 128, 128
    m = Yolov8Obb(engine, nc=nc, size="n", height=H, width=W, max_batch=B, dtype="f32")
    A = m.A
    g = torch.Generator().manual_seed(5)
    rp = {"boxes": torch.randn(B, 64, A, generator=g).requires_grad_(True), "scores": (torch.randn(B, nc, A, generator=g) - 1.0).requires_grad_(True),
          "feats": [torch.zeros(B, 1, H // s, W // s) for s in (8, 16, 32)]}
    raw = (torch.randn(B, 1, A, generator=g)).requires_grad_(True)
    rp["angle"] = (raw.sigmoid() - 0.25) * np.pi
    batch = O.synthetic_obb_batch(B, H, W, nc, seed=2, kmax=10)
    m.set_preds({k: v.detach().numpy() for k, v in rp.items() if k != "feats"})
    loss, items = v8OBBLoss(m)(None, {k: v.numpy() for k, v in batch.items()})
    rloss, ritems = O.v8OBBLoss(nc)(rp, batch)
    assert np.allclose(items, ritems.detach().numpy(), rtol=1e-3, atol=1e-5), (items, ritems)
    rloss.sum().backward()
    for key, t in (("dangle", raw), ("dboxes", rp["boxes"]), ("dscores", rp["scores"])):
        r = t.grad.numpy()
        assert np.abs(m.get_output(key) - r).max() <= 1e-3 * np.abs(r).max(), key
    m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_obb_loss_without_labels(backend, engine):
    from yolosharp_amd.model import Yolov8Obb, v8OBBLoss
    m = Yolov8Obb(engine, nc=2, size="n", height=32, width=32, max_batch=1, dtype="f32")
    m.init_weights(3)
    m.forward(np.random.default_rng(0).random((1, 3, 32, 32), np.float32), fetch=False)
    e = np.zeros((0,), np.float32)
    loss, items = v8OBBLoss(m)(None, {"batch_idx": e, "cls": e, "bboxes": e.reshape(0, 5)})
    assert items[0] == 0 and items[2] == 0 and items[3] == 0 and items[1] > 0
    assert not m.get_output("dangle").any()
    tiny = {"batch_idx": np.zeros(1, np.float32), "cls": np.zeros(1, np.float32), "bboxes": np.array([[0.5, 0.5, 0.5, 0.03, 0.3]], np.float32)}
    _, items2 = v8OBBLoss(m)(None, tiny)                           # 0.03 * 32 px < 2 px: the only label is filtered (Loss.cs:563)
    assert np.array_equal(items, items2)
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gpu"])
def test_yolov11s_obb_loss_backward_full_resolution_f32(backend, engine):
    m = _obb_train_parity(engine, 11, "s", 2, 640, 640, 1e-3, 2e-3, nc=15, kmax=12)
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gpu"])
def test_yolov8s_obb_bf16_train_steps(backend, engine):
    from yolosharp_amd.model import Yolov8Obb, v8OBBLoss, AMPWrapper
    nc, B, H, W = 15, 4, 320, 320
    ref = make_ref(O.Yolov8Obb, nc, "s")
    m = _load(engine, ref, Yolov8Obb, nc, "s", B, H, W, dtype="bf16")
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(3))
    batch = O.synthetic_obb_batch(B, H, W, nc, seed=1, kmax=6)
    nb = {k: v.numpy() for k, v in batch.items()}
    ref.train()
    _, rpreds = ref(x)
    _, ritems = O.v8OBBLoss(nc)(rpreds, batch)
    amp = AMPWrapper(m, lr=2e-3)
    crit = v8OBBLoss(m)
    hist = []
    for _ in range(6):
        loss, items = amp.TrainStep(x.numpy(), nb, crit)
        assert np.isfinite(items).all()
        hist.append(float(loss.sum()))
    assert abs(hist[0] / B - float(ritems.sum())) < 0.1 * float(ritems.sum())
    assert hist[-1] < hist[0]
    m.close()


# ----------------------------------------------------------------------------- PoseDetector.Val: Metrics.kpt_iou (Metrics.cs:186-212)
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("K,D", [(17, 3), (5, 2)])
def test_kpt_iou(backend, engine, K, D):
    g = torch.Generator().manual_seed(K)
    n, m = 7, 23
    k1 = torch.cat((torch.rand(n, K, 2, generator=g) * 640, torch.randint(0, 3, (n, K, 1), generator=g).float()), -1)
    k1[2, :, 2] = 0                                                # an object with no labelled keypoint: 0 / eps
    k2 = torch.cat((k1[torch.randint(0, n, (m,), generator=g), :, :2] + torch.randn(m, K, 2, generator=g) * 12, torch.rand(m, K, 1, generator=g)), -1)[..., :D]
    area = torch.rand(n, generator=g) * 4e4 * 0.53 + 50
    want = O.kpt_iou(k1, k2, area).numpy()
    got = engine.kpt_iou(k1.numpy(), k2.numpy(), area.numpy())
    assert got.shape == (n, m) and np.allclose(got, want, rtol=1e-4, atol=1e-6)
    assert want.max() > 0.5 and not got[2].any()
    tp = engine.match_predictions(np.zeros(m, np.float32), np.zeros(n, np.float32), got)      # the second match of PoseDetector.cs:158
    assert np.array_equal(tp, O.match_predictions(torch.zeros(m), torch.zeros(n), torch.from_numpy(want)).numpy())
    assert engine.kpt_iou(np.zeros((0, K, 3), np.float32), k2.numpy(), np.zeros(0, np.float32)).shape == (0, m)


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gpu"])
@pytest.mark.parametrize("task", ["Obb", "Pose"])
def test_fp8_mode_train_steps(backend, engine, task):
    """dtype = fp8 on the extra heads: the wide 3x3 layers run the fp8 MFMA kernels, the cv4 towers (51-wide padded inputs, 1-channel
    output) stay bf16; items finite and close to the bf16 engine's on the first step, loss falling."""
    from yolosharp_amd import model as M
    nc, B, H, W = (15, 4, 320, 320) if task == "Obb" else (1, 4, 320, 320)
    x = np.random.default_rng(3).random((B, 3, H, W), dtype=np.float32)
    tb = O.synthetic_obb_batch(B, H, W, nc, seed=1, kmax=6) if task == "Obb" else O.synthetic_batch(B, H, W, nc, seed=1, kmax=6)
    if task == "Pose":
        tb["keypoints"] = O.synthetic_keypoints(tb)
    nb = {k: v.numpy() for k, v in tb.items()}
    first = {}
    for dt in ("bf16", "fp8"):
        m = getattr(M, f"Yolov8{task}")(engine, nc=nc, size="s", height=H, width=W, max_batch=B, dtype=dt)
        m.init_weights(5)
        amp = M.AMPWrapper(m, lr=2e-3)
        crit = (M.v8OBBLoss if task == "Obb" else M.v8PoseLoss)(m)
        hist = []
        for _ in range(5):
            loss, items = amp.TrainStep(x, nb, crit)
            assert np.isfinite(items).all(), (dt, items)
            hist.append(items)
        first[dt] = hist[1]                       # step 2: the first step of the fp8 engine still runs bf16 kernels (scale bootstrap)
        assert hist[-1].sum() < hist[0].sum(), (dt, hist[0], hist[-1])
        m.close()
    assert np.allclose(first["fp8"], first["bf16"], rtol=0.1, atol=0.05), first
