"""Obb / Pose heads: the cv4 towers, dist2rbox / kpts_decode and the rotated predict path.
Oracle = oracle/yolo_oracle.py (Head.cs:376-606, Tal.cs:389-408).  fp32 tolerance 1e-3; forward / predict only."""
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
def test_criterion_is_refused(backend, engine):
    """OBB / Pose models are forward / predict only: the detection criterion refuses them instead of training the wrong loss."""
    from yolosharp_amd import YsError
    from yolosharp_amd.model import Yolov8Obb, v8DetectionLoss
    m = Yolov8Obb(engine, nc=3, size="n", height=32, width=32, max_batch=1, dtype="f32")
    m.init_weights(1)
    m.forward(np.zeros((1, 3, 32, 32), np.float32), fetch=False)
    batch = {"batch_idx": np.zeros(1, np.float32), "cls": np.zeros(1, np.float32), "bboxes": np.array([[0.5, 0.5, 0.2, 0.2]], np.float32)}
    with pytest.raises(YsError, match="OBB criterion is not built"):
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
