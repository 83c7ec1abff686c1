"""Fused BN-backward reduction (round 3; csrc/conv_epi.h BnRedSeg, csrc/model.hip plan_bnred).

The dgrad launch that completes the gradient dz of a BN Conv unit's output also produces that unit's sums
sum(du), sum(du * y) (du = dz * SiLU'(BN(y))) in its epilogue, replacing chan_reduce_kernel's pass over dz and y.  The two
forms differ only in fp32 summation order, so:
  * layers whose upstream is identical in both modes (the Detect towers' `.1` units: dz comes straight from the `.2` dgrad of the
    loss gradient) must agree to fp32 rounding;
  * every other gradient agrees within the bf16 noise two equivalent bf16 backward passes show at this size.
Oracle parity of the fused path itself is covered by the bf16 suites (test_model / test_configs run with the default = fused)."""
import os

import numpy as np
import pytest
import torch

from conftest import BACKENDS
from oracle import yolo_oracle as O


def _grads(engine, mode, fam, size, x, batch, sd, capfd=None):
    from yolosharp_amd.model import Yolov8, Yolov11, v8DetectionLoss
    engine.set_option("BNRED", int(mode)); engine.set_option("BNRED_LOG", 1)
    try:
        M = Yolov8 if fam == "8" else Yolov11
        m = M(engine, nc=80, size=size, height=x.shape[2], width=x.shape[3], max_batch=x.shape[0], dtype="bf16")
        m.load_state_dict(sd)
        m.train(); m.forward(x, fetch=False)
        _, items = v8DetectionLoss(m)(None, batch)
        m.zero_grad(); m.backward()
        g = m.grads()
        m.close()
    finally:
        engine.unset_option("BNRED"); engine.unset_option("BNRED_LOG")
    return g, items


@pytest.mark.parametrize("backend", BACKENDS)
def test_fused_bn_backward_reduction_matches_the_separate_pass(backend, engine, capfd):
    B, H, W = 2, 64, 64
    torch.manual_seed(0)
    ref = O.Yolov8(nc=80, size="n")
    sd = {k: v.detach().numpy() for k, v in ref.state_dict().items()}
    x = torch.rand(B, 3, H, W).numpy()
    batch = {k: v.numpy() for k, v in O.synthetic_batch(B, H, W, 80, seed=1, kmax=6).items()}
    g0, it0 = _grads(engine, "0", "8", "n", x, batch, sd)
    capfd.readouterr()
    g1, it1 = _grads(engine, "1", "8", "n", x, batch, sd)
    err = capfd.readouterr().err
    # the plan: all but the SPPF units (first reader = max-pool) and the unit feeding only the up-sample are fused
    line = [l for l in err.splitlines() if "fused BN-backward reduction:" in l]
    assert line, err
    nf = int(line[0].split("reduction:")[1].split("of")[0])
    assert nf >= 50, line[0]
    assert np.array_equal(it0, it1)                       # forward and loss are untouched
    tight = [k for k in g0 if k.startswith("model.22.") and ".1.bn." in k]
    assert len(tight) == 12
    for k in tight:                                       # identical dz in both modes: only the summation order differs
        a, b = g0[k], g1[k]
        assert np.abs(a - b).max() <= 2e-5 * np.abs(a).max() + 1e-12, (k, np.abs(a - b).max(), np.abs(a).max())
    for k in g0:                                          # everything downstream: bf16 noise of two equivalent backward passes
        a, b = g0[k], g1[k]
        scale = np.abs(a).max()
        if scale < 0.05:                                  # cancellation-level tensors (e.g. SPPF cv1's dbeta at 2x2 pixels)
            continue
        assert np.abs(a - b).max() <= 0.06 * scale, (k, np.abs(a - b).max(), scale)


@pytest.mark.parametrize("backend", BACKENDS)
def test_fused_reduction_with_shortcut_blocks_and_two_sources(backend, engine):
    """YOLOv11s: C3k2 / C3k (shortcut Bottlenecks: the residual-gradient accumulation moves from the reduction pass to the apply
    pass), units whose channels are completed by two different dgrad launches (C2f-style cv1: chunk a by cv2, chunk b by m.0.cv1),
    depthwise and attention readers (not fused).  Same criteria as above on the Detect towers."""
    B, H, W = 2, 64, 64
    torch.manual_seed(1)
    ref = O.Yolov11(nc=80, size="s")
    sd = {k: v.detach().numpy() for k, v in ref.state_dict().items()}
    x = torch.rand(B, 3, H, W).numpy()
    batch = {k: v.numpy() for k, v in O.synthetic_batch(B, H, W, 80, seed=2, kmax=6).items()}
    g0, it0 = _grads(engine, "0", "11", "s", x, batch, sd)
    g1, it1 = _grads(engine, "1", "11", "s", x, batch, sd)
    assert np.array_equal(it0, it1)
    tight = [k for k in g0 if k.startswith("model.23.cv2.") and ".1.bn." in k]
    assert len(tight) == 6
    for k in tight:
        a, b = g0[k], g1[k]
        assert np.abs(a - b).max() <= 2e-5 * np.abs(a).max() + 1e-12, (k, np.abs(a - b).max(), np.abs(a).max())
    for k in g0:
        a, b = g0[k], g1[k]
        scale = np.abs(a).max()
        if scale < 0.05:
            continue
        assert np.abs(a - b).max() <= 0.08 * scale, (k, np.abs(a - b).max(), scale)
