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


@pytest.mark.gpu
@pytest.mark.parametrize("task", ["obb", "pose", "segment"])
def test_trainer_other_tasks_gpu(tmp_path, task):
    """The same loop with the task's own criterion and validator (Obber / PoseDetector / Segmenter): loss items of the task's
    length, falling on a fixed batch; Val returns the box metrics (+ mask / pose metrics)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import yolo_oracle as O
    from yolosharp_amd import Engine
    from yolosharp_amd import model as M
    eng = Engine(0)
    B, H, W = 8, 128, 128
    nc = {"obb": 15, "pose": 1, "segment": 80}[task]
    m = {"obb": M.Yolov8Obb, "pose": M.Yolov8Pose, "segment": M.Yolov8Segment}[task](eng, nc=nc, size="n", height=H, width=W, max_batch=B, dtype="bf16")
    m.init_weights(1)
    if task == "obb":
        tb = O.synthetic_obb_batch(B, H, W, nc, seed=1, kmax=4)
    else:
        tb = O.synthetic_batch(B, H, W, nc, seed=1, kmax=4)
        if task == "pose":
            tb["keypoints"] = O.synthetic_keypoints(tb)
        else:
            tb["masks"] = O.synthetic_masks(tb, B, H // 4, W // 4)
    data = {k: v.numpy() for k, v in tb.items()}
    data["images"] = np.random.default_rng(0).random((B, 3, H, W), dtype=np.float32)
    tr = T.Trainer(m, epochs=2, nb=6, out_dir=str(tmp_path), lr0=2e-3, warmup_bias_lr=2e-3)
    hist = tr.fit(lambda: [data] * 6, lambda: [data])
    n_items = {"obb": 4, "pose": 5, "segment": 5}[task]
    assert hist[0]["train_loss"].shape == (n_items,) and np.all(np.isfinite(hist[1]["train_loss"]))
    # 12 warm-up steps at <= 12 % of lr0: the plumbing is what is tested here (descent itself: test_obb_pose / test_segment train steps)
    assert hist[1]["train_loss"].sum() < 1.01 * hist[0]["train_loss"].sum()
    assert hist[1]["val_loss"].shape == (n_items,) and len(hist[1]["metrics"]) == 4
    assert ("metrics2" in hist[1]) == (task != "obb")
    assert os.path.exists(os.path.join(str(tmp_path), "weights", "last.bin"))
    m.close()


def test_empty_batches_do_not_advance_the_warmup_index():
    """TrainEpoch's `continue` on an empty batch skips its i++ (YoloBaseTaskModel.cs:322-325,349); the epoch returns the SUM of
    the per-step loss items, and zeros when no step ran."""
    class FakeAmp:
        def __init__(self):
            self.seen = []
            self.lrs = None

        def TrainStep(self, images, data, crit):
            self.seen.append(list(self.lrs))
            return np.ones(3, np.float32), np.array([1.0, 2.0, 3.0], np.float32)

    tr = T.Trainer.__new__(T.Trainer)
    tr.amp, tr.crit = FakeAmp(), object()
    tr.sched = T.LrSchedule(nc=80, epochs=10, nb=4, lrf=0.01)
    full = {"batch_idx": np.zeros(2, np.float32), "images": np.zeros((1, 3, 32, 32), np.float32)}
    empty = {"batch_idx": np.zeros(0, np.float32), "images": np.zeros((1, 3, 32, 32), np.float32)}
    out = tr.train_epoch([full, empty, empty, full], epoch=1)
    assert np.array_equal(out, np.array([2.0, 4.0, 6.0], np.float32)) and tr.steps_run == 2
    ref = T.LrSchedule(nc=80, epochs=10, nb=4, lrf=0.01)
    assert tr.amp.seen == [ref.begin_iteration(1, 0), ref.begin_iteration(1, 1)]      # second trained batch is i = 1, not 3
    assert np.array_equal(tr.train_epoch([empty, empty], epoch=2), np.zeros(3, np.float32))


def test_reference_param_groups_step_bn_twice(emu_lib_path):
    """param_groups = "reference": the optimizer groups exactly as written in YoloBaseTaskModel.cs:144-151 (every BatchNorm
    weight / bias listed in two groups, one shared AdamW state) against the oracle's literal group loop, two optimizer steps
    with distinct per-group learning rates; the default disjoint mode differs from it only on BatchNorm parameters."""
    import torch
    from oracle import yolo_oracle as O
    from yolosharp_amd import Engine
    from yolosharp_amd.model import Yolov8, v8DetectionLoss
    from test_model import make_ref
    eng = Engine(lib_path=emu_lib_path)
    B, H, W, nc = 2, 32, 32, 80
    ref = make_ref(seed=9).train()
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(1))
    batch = O.synthetic_batch(B, H, W, nc, seed=2, kmax=3)
    nb = {k: v.numpy() for k, v in batch.items()}
    lrs = [3e-3, 1e-3, 2e-3]
    models = {}
    for mode in ("reference", "disjoint"):
        m = Yolov8(eng, nc=nc, size="n", height=H, width=W, max_batch=B, dtype="f32")
        m.load_state_dict({k: v.detach().numpy() for k, v in ref.state_dict().items()})
        m.set_param_groups(mode)
        models[mode] = m
    params = {n: p.detach().clone() for n, p in ref.named_parameters()}
    state = {}
    for step in range(2):
        for m in models.values():
            m.train(); m.forward(x.numpy(), fetch=False); v8DetectionLoss(m)(None, nb); m.zero_grad(); m.backward()
        grads = {k: torch.from_numpy(v) for k, v in models["reference"].grads().items()}     # same gradients feed both optimizers
        O.adamw_step_reference_groups(params, grads, state, lrs)
        for m in models.values():
            m.adamw_step(lrs)
        if step == 0:      # keep the two engines on the same weights for the second step's gradients
            models["disjoint"].load_state_dict({k: v for k, v in models["reference"].state_dict().items() if "running" not in k and "num_batches" not in k}, strict=False)
        got = models["reference"].state_dict()
        for name, p in params.items():
            if name not in grads:
                continue
            g = grads[name].numpy()
            big = np.abs(g) > 1e-2 * np.abs(g).max()          # elements whose Adam direction is not rounding noise
            d = np.abs(got[name] - p.numpy())
            assert d[big].max(initial=0.0) <= 5e-6 + 2e-4 * np.abs(p.numpy()).max(), (step, name, d[big].max())
        assert state["model.0.bn.weight"][2] == 2 * (step + 1) and state["model.0.conv.weight"][2] == step + 1
    a, b = models["reference"].state_dict(), models["disjoint"].state_dict()
    assert np.abs(a["model.0.bn.weight"] - b["model.0.bn.weight"]).max() > 1e-4          # two updates vs one
    with pytest.raises(Exception):
        models["reference"].set_param_groups("disjoint")                                   # not after the optimizer has stepped
    for m in models.values():
        m.close()
