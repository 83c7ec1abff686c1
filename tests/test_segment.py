"""Segment rows: Proto + ConvTranspose2d phases, cv4 mask-coefficient towers,
v8SegmentationLoss with crop_mask, Ops.process_mask.  Oracle = oracle/yolo_oracle.py.  fp32 tolerance 1e-3."""
import numpy as np
import pytest
import torch

from conftest import BACKENDS
from oracle import yolo_oracle as O
from test_model import relerr


def make_ref(family, nc, size, seed=0):
    torch.manual_seed(seed)
    ref = (O.Yolov8Segment if family == 8 else O.Yolov11Segment)(nc=nc, size=size)
    for mod in ref.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.normal_(0, 0.1)
            mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
    return ref


def _segment_parity(engine, family, size, B, H, W, tol_fwd, tol_grad, cpu_crop=False, full_backward=True):
    from yolosharp_amd.model import Yolov8Segment, Yolov11Segment, v8SegmentationLoss
    nc = 80
    ref = make_ref(family, nc, size)
    m = (Yolov8Segment if family == 8 else Yolov11Segment)(engine, nc=nc, size=size, height=H, width=W, max_batch=B, dtype="f32")
    info = m.tensor_info()
    assert [n for n, s, p in info if p] == [k for k, _ in ref.named_parameters()]      # cv2, cv3, dfl, proto(cv1,cv2,cv3,upsample), cv4
    assert {n: tuple(s) for n, s, p in info} == {k: (tuple(v.shape) if v.dim() else (1,)) for k, v in ref.state_dict().items()}
    sd = {k: v.detach().numpy() for k, v in ref.state_dict().items()}
    m.load_state_dict(sd)
    back = m.state_dict()
    k = [n for n in sd if n.endswith("proto.upsample.weight")][0]
    assert np.array_equal(back[k], sd[k])                                               # [Cin][Cout][2][2] <-> phase-major round trip
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(3))
    batch = O.synthetic_batch(B, H, W, nc, seed=1, kmax=6)
    batch["masks"] = O.synthetic_masks(batch, B, H // 4, W // 4)
    nb = {k: v.numpy() for k, v in batch.items()}
    # ---- eval: pred = cat(decode, raw mask coefficients) (Head.cs:309-313)
    m.eval(); ref.eval()
    inf, preds = m.forward(x.numpy())
    with torch.no_grad():
        rinf, rpreds = ref(x)
    for key in ("boxes", "scores", "mask_coefficient", "proto"):
        assert relerr(preds[key], rpreds[key]) < tol_fwd, key
    assert inf["boxes"].shape == (B, 4 + nc + 32, m.A)
    assert relerr(inf["boxes"][:, :4 + nc], rinf["boxes"][:, :4 + nc]) < tol_fwd
    assert relerr(inf["boxes"][:, 4 + nc:], rinf["boxes"][:, 4 + nc:]) < tol_fwd
    # ---- train: loss items (box, seg, cls, dfl, semseg) and every gradient
    m.train(); ref.train()
    _, preds = m.forward(x.numpy())
    _, rpreds = ref(x)
    for key in ("boxes", "scores", "mask_coefficient", "proto"):
        assert relerr(preds[key], rpreds[key]) < tol_fwd, key
    rpreds["mask_coefficient"].retain_grad(); rpreds["proto"].retain_grad()
    loss, items = v8SegmentationLoss(m, cpu_crop_branch=cpu_crop)(None, nb)
    rloss, ritems = O.v8SegmentationLoss(nc, cpu_crop_branch=cpu_crop)(rpreds, batch)
    assert items.shape == (5,) and float(ritems[1]) > 0
    assert np.allclose(items, ritems.numpy(), rtol=1e-3, atol=1e-5), (items, ritems)
    assert np.allclose(loss, rloss.detach().numpy(), rtol=1e-3, atol=1e-4)
    rloss.sum().backward()
    for key in ("mask_coefficient", "proto"):
        r = rpreds[key].grad.numpy()
        gq = m.get_output("d" + key)
        assert np.abs(gq - r).max() <= tol_grad * np.abs(r).max(), key
    if not full_backward:      # the crop branch only changes the mask term: its gradients w.r.t. the head outputs were checked above
        m.close()
        return
    m.zero_grad(); m.backward()
    grads = m.grads()
    gscale = max(float(p.grad.abs().max()) for _, p in ref.named_parameters() if p.grad is not None)
    for name, p in ref.named_parameters():
        if p.grad is None:
            continue
        r = p.grad.numpy()
        assert np.abs(grads[name] - r).max() <= tol_grad * np.abs(r).max() + 1e-6 * gscale, name
    m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_yolov8n_segment_f32(backend, engine):
    _segment_parity(engine, 8, "n", 2, 64, 64, 1e-3, 1e-3)


@pytest.mark.parametrize("backend", BACKENDS)
def test_yolov8n_segment_cpu_crop_branch(backend, engine):
    """Ops.crop_mask's CPU-only integer-truncation branch (Ops.cs:421-435) is what a CPU run of the reference computes."""
    _segment_parity(engine, 8, "n", 2, 64, 64, 1e-3, 2e-3, cpu_crop=True, full_backward=False)


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gpu"])
def test_yolov11s_segment_full_resolution_f32(backend, engine):
    _segment_parity(engine, 11, "s", 2, 640, 640, 1e-3, 2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gpu"])
def test_yolov8s_segment_bf16_train_step(backend, engine):
    from yolosharp_amd.model import Yolov8Segment, v8SegmentationLoss
    B, H, W, nc = 8, 640, 640, 80
    x = np.random.default_rng(0).random((B, 3, H, W), dtype=np.float32)
    tb = O.synthetic_batch(B, H, W, nc, seed=1)
    tb["masks"] = O.synthetic_masks(tb, B, H // 4, W // 4)
    batch = {k: v.numpy() for k, v in tb.items()}
    items = {}
    for dt in ("bf16", "f32"):
        m = Yolov8Segment(engine, nc=nc, size="s", height=H, width=W, max_batch=B, dtype=dt)
        m.init_weights(4); m.train()
        m.forward(x, fetch=False)
        _, it = v8SegmentationLoss(m)(None, batch)
        m.zero_grad(); m.backward(); m.adamw_step([1e-3] * 3)
        items[dt] = it
        m.close()
    assert np.all(np.isfinite(items["bf16"])) and np.allclose(items["bf16"], items["f32"], rtol=5e-2, atol=1e-3), items


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("upsample", [False, True])
@pytest.mark.parametrize("cpu_crop", [False, True])
def test_process_mask(backend, engine, upsample, cpu_crop):
    g = torch.Generator().manual_seed(5)
    nm, mh, mw, ih, iw, n = 32, 24, 40, 96, 160, 7
    protos = torch.randn(nm, mh, mw, generator=g)
    coef = torch.randn(n, nm, generator=g)
    xy = torch.rand(n, 2, generator=g) * torch.tensor([iw * 0.6, ih * 0.6])
    wh = torch.rand(n, 2, generator=g) * torch.tensor([iw * 0.4, ih * 0.4]) + 2.0
    boxes = torch.cat((xy, xy + wh), 1)
    ref = O.process_mask(protos, coef, boxes, (ih, iw), upsample=upsample, cpu_branch=cpu_crop).numpy().astype(bool)
    out = engine.process_mask(protos.numpy(), coef.numpy(), boxes.numpy(), (ih, iw), upsample=upsample, cpu_crop_branch=cpu_crop)
    assert out.shape == ref.shape
    # '> 0' on a float sum: allow sign flips only where the oracle's value is at rounding level
    mism = out != ref
    if mism.any():
        vals = O.crop_mask(coef.matmul(protos.view(nm, -1)).view(-1, mh, mw),
                           boxes * torch.tensor([mw / iw, mh / ih, mw / iw, mh / ih]), cpu_crop)
        if upsample:
            vals = torch.nn.functional.interpolate(vals[None], size=(ih, iw), mode="bilinear", align_corners=False)[0]
        assert float(vals.numpy()[mism].__abs__().max()) < 1e-4
    assert mism.mean() < 1e-3
    assert engine.process_mask(protos.numpy(), np.zeros((0, nm), np.float32), np.zeros((0, 4), np.float32), (ih, iw)).shape == (0, mh, mw)
