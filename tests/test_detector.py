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
    loss_items, (P, R, m50, m5095) = det.Val([d], conf_thres=0.001)
    assert loss_items.shape == (3,) and np.all(np.isfinite(loss_items))
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
