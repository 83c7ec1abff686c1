"""Detect / Segment as standalone modules (`ys_head_*`, yolosharp_amd/heads.py) against the oracle's Head.cs restatement:
training outputs, the criterion on the handle, every parameter gradient and the gradients w.r.t. the three input feature maps;
eval decode; caller-supplied output gradients (ys_head_set_grads)."""
import numpy as np
import pytest
import torch

from conftest import BACKENDS
from oracle import yolo_oracle as O


def _feats(B, ch, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, c, H // s, W // s, generator=g) for c, s in zip(ch, (8, 16, 32))]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("legacy", [True, False])
def test_detect_head_module(backend, engine, legacy):
    from yolosharp_amd.heads import Detect
    from yolosharp_amd.model import v8DetectionLoss
    B, H, W, nc, ch = 2, 64, 64, 80, (64, 128, 256)
    torch.manual_seed(3)
    ref = O.Detect(nc=nc, ch=ch, legacy=legacy)
    for mod in ref.modules():                                   # non-trivial BN state so that eval differs from train
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 1.5); mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.uniform_(-0.3, 0.3)
    m = Detect(engine, nc=nc, ch=ch, legacy=legacy, height=H, width=W, max_batch=B, dtype="f32")
    sd = {k: v.detach().numpy() for k, v in ref.state_dict().items()}
    info = m.tensor_info()                                      # module-relative names, parameters in registration order (Head.cs:47-56)
    assert [n for n, s_, p in info if p] == [k for k, _ in ref.named_parameters()]
    assert set(n for n, s_, p in info) == set(sd.keys())
    m.load_state_dict(sd)
    xs = _feats(B, ch, H, W, 5)
    for x in xs:
        x.requires_grad_(True)
    batch = O.synthetic_batch(B, H, W, nc, seed=7, kmax=5)
    ref.train(); m.train()
    _, rpreds = ref(xs)
    _, preds = m([x.detach().numpy() for x in xs])
    for k in ("boxes", "scores"):
        assert np.abs(preds[k] - rpreds[k].detach().numpy()).max() <= 1e-3 * np.abs(rpreds[k].detach().numpy()).max(), k
    rloss, ritems = O.v8DetectionLoss(nc)(rpreds, batch)
    _, items = v8DetectionLoss(m)(None, {k: v.numpy() for k, v in batch.items()})
    assert np.allclose(items, ritems.detach().numpy(), rtol=1e-3, atol=1e-5), (items, ritems)
    rloss.sum().backward()
    m.zero_grad()
    dx = m.backward()
    for a, x in zip(dx, xs):
        r = x.grad.numpy()
        assert np.abs(a - r).max() <= 2e-3 * np.abs(r).max() + 1e-7
    g = m.grads()
    for name, p in ref.named_parameters():
        if p.grad is None:
            continue                                            # dfl.conv.weight is frozen (Head.cs:221)
        r = p.grad.numpy()
        assert np.abs(g[name] - r).max() <= 2e-3 * np.abs(r).max() + 1e-7, name
    # caller-supplied output gradients reproduce the criterion's
    dx2 = m.backward({"boxes": m.get_output("dboxes"), "scores": m.get_output("dscores")})
    for a, b in zip(dx, dx2):
        assert np.abs(a - b).max() <= 1e-5 * np.abs(a).max() + 1e-9
    # eval: decoded predictions (Head.cs:204-223)
    ref.eval(); m.eval()
    with torch.no_grad():
        rinf, _ = ref([x.detach() for x in xs])
    inf, _ = m([x.detach().numpy() for x in xs])
    r = rinf["boxes"].numpy()
    assert np.abs(inf["boxes"] - r).max() <= 1e-3 * np.abs(r).max()
    m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_segment_head_module(backend, engine):
    from yolosharp_amd.heads import Segment
    from yolosharp_amd.model import v8SegmentationLoss
    B, H, W, nc, ch = 2, 64, 64, 80, (64, 128, 256)
    torch.manual_seed(4)
    ref = O.Segment(nc=nc, nm=32, npr=ch[0], ch=ch, legacy=True)
    m = Segment(engine, nc=nc, ch=ch, legacy=True, height=H, width=W, max_batch=B, dtype="f32")
    sd = {k: v.detach().numpy() for k, v in ref.state_dict().items()}
    info = m.tensor_info()
    assert [n for n, s_, p in info if p] == [k for k, _ in ref.named_parameters()]
    assert set(n for n, s_, p in info) == set(sd.keys())
    m.load_state_dict(sd)
    xs = _feats(B, ch, H, W, 6)
    for x in xs:
        x.requires_grad_(True)
    batch = O.synthetic_batch(B, H, W, nc, seed=8, kmax=4)
    batch["masks"] = O.synthetic_masks(batch, B, H // 4, W // 4)
    ref.train(); m.train()
    _, rpreds = ref(xs)
    _, preds = m([x.detach().numpy() for x in xs])
    for k in ("boxes", "scores", "mask_coefficient", "proto"):
        r = rpreds[k].detach().numpy()
        assert np.abs(preds[k] - r).max() <= 1e-3 * np.abs(r).max(), k
    rloss, ritems = O.v8SegmentationLoss(nc)(rpreds, batch)
    _, items = v8SegmentationLoss(m)(None, {k: v.numpy() for k, v in batch.items()})
    assert np.allclose(items, ritems.detach().numpy(), rtol=2e-3, atol=1e-5), (items, ritems)
    rloss.sum().backward()
    m.zero_grad()
    dx = m.backward()
    for a, x in zip(dx, xs):
        r = x.grad.numpy()
        assert np.abs(a - r).max() <= 3e-3 * np.abs(r).max() + 1e-7
    g = m.grads()
    for name in ("proto.cv1.conv.weight", "proto.upsample.weight", "cv4.1.2.bias", "cv2.0.0.conv.weight", "cv3.2.1.bn.weight"):
        r = dict(ref.named_parameters())[name].grad.numpy()
        assert np.abs(g[name] - r).max() <= 3e-3 * np.abs(r).max() + 1e-7, name
    m.close()
