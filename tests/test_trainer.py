"""LR schedule / warm-up plumbing (YoloBaseTaskModel.cs:142,160-183,303-319,492-536) and the Trainer loop on the engine."""
import math

import numpy as np
import pytest

from yolosharp_amd import trainer as T


def test_lr_fit_and_lambdas():
    assert T.lr_fit(80) == round(0.002 * 5 / 84, 6) == 0.000119
    lin = T.lr_lambda(1.0, 0.01, 100)
    assert lin(0) == 1.0 and abs(lin(100) - 0.01) < 1e-12 and abs(lin(50) - 0.505) < 1e-12 and lin(150) == 0.01
    cos = T.one_cycle(1.0, 0.01, 100)
    assert cos(0) == 1.0 and abs(cos(100) - 0.01) < 1e-12 and abs(cos(50) - 0.505) < 1e-12
    assert T.interp(5, [0, 10], [0.1, 0.3]) == pytest.approx(0.2) and T.interp(-1, [0, 10], [0.1, 0.3]) == 0.1
    assert T.interp(10, [0, 10], [0.1, 0.3]) == 0.3 and T.interp(4, [0, 4, 10], [0.0, 7.0, 9.0]) == 7.0
    with pytest.raises(ValueError):
        T.interp(1, [0, 1], [0])


def test_warmup_and_scheduler_sequence():
    """nb = 40 iterations / epoch, 3 warm-up epochs -> nw = max(120, 100) = 120; epochs are 1-based so ni starts at nb."""
    s = T.LrSchedule(nc=80, epochs=10, nb=40, lrf=0.01)
    lr0 = 0.000119
    assert s.nw == 120 and s.lrs == [lr0] * 3
    l = s.begin_iteration(1, 0)                      # ni = 40
    d1 = lr0 * T.lr_lambda(1.0, 0.01, 10)(1)
    assert l[0] == pytest.approx(0.1 + 40 / 120 * (d1 - 0.1)) and l[1] == l[2] == pytest.approx(40 / 120 * d1)
    s.begin_iteration(1, 39); s.end_epoch()
    assert s.lrs == [pytest.approx(lr0 * 0.901)] * 3                      # LambdaLR after one step: lambda(1)
    l = s.begin_iteration(2, 40)                     # ni = 120 = nw: last warm-up iteration lands on the target
    assert l == [pytest.approx(lr0 * T.lr_lambda(1.0, 0.01, 10)(2))] * 3
    l2 = s.begin_iteration(3, 1)                     # ni = 121 > nw: the warm-up no longer touches the rates
    assert l2 == l
    s.end_epoch()
    assert s.lrs == [pytest.approx(lr0 * T.lr_lambda(1.0, 0.01, 10)(2))] * 3
    # short runs: nw is at least 100 iterations
    assert T.LrSchedule(80, 3, nb=5).nw == 100


@pytest.mark.gpu
def test_trainer_two_epochs_gpu(tmp_path):
    """Train loop on the device: warm-up rates reach AdamW, the loss goes down on a fixed synthetic batch, Val runs,
    best.bin / last.bin load back."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import yolo_oracle as O
    from yolosharp_amd import Engine, weights_bin
    from yolosharp_amd.model import Yolov8
    eng = Engine(0)
    B, H, W, nc = 8, 128, 128, 80
    m = Yolov8(eng, nc=nc, size="n", height=H, width=W, max_batch=B, dtype="bf16")
    m.init_weights(1)
    rng = np.random.default_rng(0)
    data = {k: v.numpy() for k, v in O.synthetic_batch(B, H, W, nc, seed=1, kmax=4).items()}
    data["images"] = rng.random((B, 3, H, W), dtype=np.float32)
    tr = T.Trainer(m, epochs=2, nb=6, out_dir=str(tmp_path), lr0=2e-3, warmup_bias_lr=2e-3)
    hist = tr.fit(lambda: [data] * 6, lambda: [data])
    assert len(hist) == 2 and np.all(np.isfinite(hist[1]["train_loss"]))
    assert hist[1]["train_loss"].sum() < hist[0]["train_loss"].sum()
    assert len(hist[1]["metrics"]) == 4
    sd, code = weights_bin.read_bin(os.path.join(str(tmp_path), "weights", "last.bin"))
    assert code == weights_bin.FLOAT32 and "model.22.dfl.conv.weight" in sd
    m.close()
