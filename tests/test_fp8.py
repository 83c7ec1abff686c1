"""fp8 convolution path (BASELINE config 5: "fp8 MFMA implicit-GEMM conv path"; recipe in csrc/f8.hip).

Parity statement for a reduced-precision mode: the HIP kernel must equal an fp32 convolution of the SAME quantised operands
(e4m3 weights / activations, e5m2 gradients, per-tensor scales) up to fp32 accumulation order and the bf16 rounding of the
stored result.  The quantiser below is an independent numpy restatement of OCP e4m3fn / e5m2 round-to-nearest-even with
saturation; the hardware conversion instructions were checked against the same rules on the MI355X (tools/probe/probe_f8.hip)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import BACKENDS


def bf16(x):
    return torch.from_numpy(np.ascontiguousarray(x, np.float32)).bfloat16().float().numpy()


def quant(x, mant_bits, min_exp, maxv):
    """Dequantised values of RNE rounding to a float format with `mant_bits` mantissa bits, smallest normal exponent `min_exp`
    (subnormals below it) and saturation at +-maxv."""
    a = np.minimum(np.abs(x.astype(np.float64)), maxv)
    _, ex = np.frexp(a)                      # a = m * 2^ex, m in [0.5, 1)
    e = np.maximum(ex - 1, min_exp)
    step = np.ldexp(1.0, e - mant_bits)
    q = np.round(a / step) * step            # np.round: half to even
    return (np.sign(x) * q).astype(np.float32)


def q_e4m3(x):
    return quant(x, 3, -6, 448.0)


def q_e5m2(x):
    return quant(x, 2, -14, 57344.0)


def test_quantiser_known_values():
    """The rules measured on the hardware (probe_f8): ties to even, subnormals, saturation (the kernels clamp before converting)."""
    x = np.array([0, 1, 1.0625, 1.125, 1.1875, 3.9, 447, 464, 500, 1e6, -1e6, 0.0146, 0.001953125, 0.0009765625, 0.0012, -0.3, 0.017], np.float32)
    want8 = np.array([0, 1, 1, 1.125, 1.25, 4, 448, 448, 448, 448, -448, 0.013671875, 0.001953125, 0, 0.001953125, -0.3125, 0.017578125], np.float32)
    assert np.array_equal(q_e4m3(x), want8)
    x5 = np.array([1.125, 1.1875, 447, 60000, 1e6, 1e-5, 7.6e-6, 0.017, 0.0012], np.float32)
    want5 = np.array([1, 1.25, 448, 57344, 57344, 1.52587890625e-05, 0, 0.015625, 0.001220703125], np.float32)
    assert np.array_equal(q_e5m2(x5), want5)


CASES = [
    # B, Cin, H, W, Cout, k, s
    (2, 64, 12, 12, 64, 3, 1),       # resident weights, two K-steps of 128 + tail (K = 576)
    (1, 32, 20, 40, 48, 3, 1),       # multi-tile image, K = 288 (tail pieces), Cout not a multiple of 32
    (2, 128, 9, 11, 128, 3, 2),      # stride 2, odd sizes
    (1, 256, 6, 6, 80, 1, 1),        # 1x1, K = 256
    (1, 320, 8, 8, 96, 3, 1),        # streamed weights (K = 2880 per row)
    (1, 160, 10, 12, 256, 3, 1),     # blocked-GEMM fp8 kernel both ways (conv_gemm.hip): forward K = 1440, dgrad K = 2304 (e5m2 operand)
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", range(len(CASES)))
def test_conv_forward_fp8_matches_quantised_fp32(backend, engine, case):
    B, Cin, H, W, Cout, k, s = CASES[case]
    g = torch.Generator().manual_seed(100 + case)
    x = torch.randn(B, Cin, H, W, generator=g).numpy() * 1.7
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).numpy()
    y = engine.conv_bn_act(x, w, k, s, bn=None, bias=np.zeros(Cout, np.float32), act=False, training=True, dtype="fp8")
    xb, wb = bf16(x), bf16(w)
    sx = np.float32(0.5) * np.float32(448.0) / np.abs(xb).max()
    sw = np.float32(448.0) / np.abs(w).max()
    xq, wq = q_e4m3(xb * sx), q_e4m3(wb * sw)
    deq = np.float32(1.0) / (sx * sw)
    ref = (F.conv2d(torch.from_numpy(xq).double(), torch.from_numpy(wq).double(), None, stride=s, padding=k // 2) * float(deq)).float().numpy()
    scale = np.abs(ref).max()
    assert np.abs(y - ref).max() <= 6e-3 * scale, (np.abs(y - ref).max(), scale)        # bf16 rounding of the stored result (2^-9)
    # and the quantisation itself is what separates it from the bf16 kernel: a few percent, not more
    plain = F.conv2d(torch.from_numpy(xb), torch.from_numpy(wb), None, stride=s, padding=k // 2).numpy()
    rel = np.linalg.norm(y - plain) / np.linalg.norm(plain)
    assert 1e-3 < rel < 8e-2, rel


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [0, 2, 3, 4, 5])
def test_conv_dgrad_fp8_matches_quantised_fp32(backend, engine, case):
    """Input gradient on the fp8 kernel (dy -> e5m2, flipped / transposed weights -> e4m3; stride 2 as four phase convolutions);
    the weight gradient keeps the bf16 operands."""
    from yolosharp_amd import _lib
    B, Cin, H, W, Cout, k, s = CASES[case]
    g = torch.Generator().manual_seed(200 + case)
    x = bf16(torch.randn(B, Cin, H, W, generator=g).numpy())
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).numpy()
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    dy = bf16(torch.randn(B, Cout, Ho, Wo, generator=g).numpy() * 1e-3)
    dx = np.zeros_like(x); dw = np.zeros_like(w)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    _lib.check(engine.lib, engine.lib.ys_conv_bwd(engine.ctx, 2, vp(x), B, Cin, H, W, vp(np.ascontiguousarray(w)), Cout, k, s, vp(dy), vp(dx), vp(dw)))
    sg = np.float32(8192.0) / np.abs(dy).max()
    sw = np.float32(448.0) / np.abs(w).max()
    dyq, wq = q_e5m2(dy * sg), q_e4m3(bf16(w) * sw)
    xt = torch.zeros(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(xt, torch.from_numpy(wq).double(), None, stride=s, padding=k // 2).backward(torch.from_numpy(dyq).double())
    ref = (xt.grad / float(sg * sw)).float().numpy()
    if Cout % 32 == 0:
        assert np.abs(dx - ref).max() <= 6e-3 * np.abs(ref).max(), (np.abs(dx - ref).max(), np.abs(ref).max())
    # weight gradient: bf16 operands, fp32 accumulation (unchanged by the mode)
    xt2 = torch.from_numpy(x).requires_grad_(False)
    wt = torch.from_numpy(bf16(w)).requires_grad_(True)
    F.conv2d(xt2, wt, None, stride=s, padding=k // 2).backward(torch.from_numpy(dy))
    assert np.abs(dw - wt.grad.numpy()).max() <= 1e-3 * np.abs(wt.grad.numpy()).max()


def _f8_launches(engine, fn):
    """Run fn() with the per-launch profile on; returns (#fp8 conv launches, #bf16 P2 launches)."""
    import tempfile
    engine.kernel_profile(True)
    fn()
    engine.synchronize()
    with tempfile.NamedTemporaryFile(suffix=".csv", delete=False) as f:
        path = f.name
    engine.kernel_profile_dump(path)
    engine.kernel_profile(False)
    txt = open(path).read()
    os.remove(path)
    return txt.count(",p2f8 ") + txt.count(",gemmf8 "), txt.count(",p2 ") + txt.count(",gemm ")


@pytest.mark.parametrize("backend", BACKENDS)
def test_model_fp8_mode(backend, engine):
    """dtype="fp8" end to end (YOLOv8s, 64x64): the first step has no recorded maxima and runs the bf16 kernels -- bit-identical
    to the bf16 model; from the second step on the eligible layers run the fp8 kernel (forward and dgrad), the loss stays close
    to the bf16 model's on the same weights, gradients keep their direction, and the step is deterministic."""
    from oracle import yolo_oracle as O
    from yolosharp_amd.model import Yolov8, v8DetectionLoss
    B, H, W, nc = 2, 64, 64, 80
    SZ = "s" if backend == "gpu" else "n"      # the interpreter runs the n graph (same layer classes, a third of the time)
    x = np.random.default_rng(0).random((B, 3, H, W), dtype=np.float32)
    batch = {k: v.numpy() for k, v in O.synthetic_batch(B, H, W, nc, seed=1, kmax=5).items()}
    ms = {}
    # the fp8 mode keeps one BN-backward reduction pass per unit (its kernels have no fused variant, csrc/model.hip plan_bnred):
    # the bf16 twin is built the same way, so that "first step = bf16 path" can be stated bit for bit
    # Same for the Detect towers: the fp8 mode runs cv2[i][0] / cv3[i][0] as separate launches (c3 = 80 is not a multiple of the fp8
    # K unit) and does not group the levels, so the twin is built with YS_HEAD_FUSE=0 / YS_GROUP=0.  (A fused 144-wide launch and two
    # 64- / 80-wide ones reduce the BatchNorm statistics over different register tiles: equal in exact arithmetic, not bit for bit.)
    # Round 5: ... and with the statistics as rows + bn_finalize (BN_ATOMIC=0), which the fp8 mode keeps (its apply pass also writes the e4m3 image).
    with engine.options(BNRED=0, HEAD_FUSE=0, GROUP=0, BN_ATOMIC=0):       # (read when a model is created; BN_ATOMIC: the fp8 mode keeps statistics rows + bn_finalize)
        for dt in ("fp8", "bf16"):
            m = Yolov8(engine, nc=nc, size=SZ, height=H, width=W, max_batch=B, dtype=dt)
            m.init_weights(3); m.train()
            ms[dt] = (m, v8DetectionLoss(m))

    def step(dt, update=True):
        m, crit = ms[dt]
        m.forward(x, fetch=False)
        _, items = crit(None, batch)
        m.zero_grad(); m.backward()
        g = m.grads()
        if update:
            m.adamw_step([1e-3] * 3)
        return items, g

    (i8, g8), (ib, gb) = step("fp8"), step("bf16")
    assert np.array_equal(i8, ib) and all(np.array_equal(g8[k], gb[k]) for k in gb), "the scale-less first step must be the bf16 path"
    n8, nb = _f8_launches(engine, lambda: step("fp8", update=False))
    assert n8 >= 30 and nb > 0, (n8, nb)                       # eligible layers on the fp8 kernel, the rest (stem, 16-channel tails) on bf16
    (i8, g8), (ib, gb) = step("fp8"), step("bf16")
    # (at 64x64 the box / dfl terms rest on a handful of foreground anchors whose assignment flips under small perturbations)
    assert np.all(np.isfinite(i8)) and np.allclose(i8, ib, rtol=1.5e-1), (i8, ib)
    num = sum(float((g8[k].ravel() * gb[k].ravel()).sum()) for k in gb)
    den = np.sqrt(sum(float((g8[k] ** 2).sum()) for k in gb) * sum(float((gb[k] ** 2).sum()) for k in gb))
    # 4-pixel deep maps at 64 x 64: fp8 quantisation noise against the bf16 twin of the SAME routing (built under BNRED=0 / HEAD_FUSE=0 / GROUP=0 above) measures
    # 0.64 - 0.72 here depending on the kernel routing of the round (0.697 in round 5); the GPU test below holds 0.9 on the classification branch at a realistic size
    assert num / den > 0.62, num / den
    # eval forward uses the recorded activation scales
    ms["fp8"][0].eval(); ms["bf16"][0].eval()
    p8 = ms["fp8"][0].forward(x)[0]["boxes"]; pb = ms["bf16"][0].forward(x)[0]["boxes"]
    assert np.corrcoef(p8.ravel(), pb.ravel())[0, 1] > 0.99
    for m, _ in ms.values():
        m.close()


@pytest.mark.gpu
def test_fp8_tracks_bf16_at_size():
    """YOLOv8s 320x320 B=8 on the device: fp8 steps against bf16 steps from the same initial weights -- loss within a few percent,
    gradient direction kept, loss decreases under AdamW; and the BASELINE config-5 graph (YOLOv8x) takes an fp8 step at 640x640."""
    from oracle import yolo_oracle as O
    from yolosharp_amd import Engine
    from yolosharp_amd.model import Yolov8, v8DetectionLoss
    eng = Engine(0)
    B, H, W, nc = 8, 320, 320, 80
    x = np.random.default_rng(0).random((B, 3, H, W), dtype=np.float32)
    batch = {k: v.numpy() for k, v in O.synthetic_batch(B, H, W, nc, seed=1).items()}
    hist = {}
    for dt in ("fp8", "bf16"):
        m = Yolov8(eng, nc=nc, size="s", height=H, width=W, max_batch=B, dtype=dt)
        m.init_weights(3); m.train()
        crit = v8DetectionLoss(m)
        rec = []
        for it in range(5):
            m.forward(x, fetch=False); _, items = crit(None, batch); m.zero_grad(); m.backward()
            rec.append((items.copy(), m.grads() if it == 1 else None))
            m.adamw_step([1e-3] * 3)
        hist[dt] = rec
        m.close()
    # step 0: no recorded maxima yet -> bf16 kernels.  Bit-equal through round 4; since round 5 the bf16 model sums its BatchNorm statistics as fixed-point integer
    # atomics (model.hip bn_atomic) while the fp8 model keeps the per-workgroup float rows (its quantising apply pass wants the finalized coefficients earlier): the same
    # numbers added in a different order and precision, 1e-4 .. 6e-4 relative on the loss items
    assert np.allclose(hist["fp8"][0][0], hist["bf16"][0][0], rtol=2e-3), (hist["fp8"][0][0], hist["bf16"][0][0])
    for it in range(1, 5):
        assert np.allclose(hist["fp8"][it][0], hist["bf16"][it][0], rtol=5e-2), (it, hist["fp8"][it][0], hist["bf16"][it][0])
    g8, gb = hist["fp8"][1][1], hist["bf16"][1][1]
    num = sum(float((g8[k].ravel() * gb[k].ravel()).sum()) for k in gb)
    den = np.sqrt(sum(float((g8[k] ** 2).sum()) for k in gb) * sum(float((gb[k] ** 2).sum()) for k in gb))
    # Direction over all parameters.  The box branch dominates what is lost: at random init the task-aligned assignment flips anchors
    # under any perturbation of the predictions, so the box-branch gradients of two runs differ however small the kernel error is
    # (measured with tools/dev/f8_cos_diag.py: Detect.cv3 0.93-0.99, Detect.cv2 ~0.4, backbone 0.55-0.7, overall 0.86-0.94 depending
    # on how many layers run in fp8); the kernels themselves are pinned against fp32 on the quantised operands above.
    assert num / den > 0.8, num / den
    cls = [k for k in gb if ".cv3." in k and k.endswith("conv.weight")]
    ncls = sum(float((g8[k].ravel() * gb[k].ravel()).sum()) for k in cls)
    dcls = np.sqrt(sum(float((g8[k] ** 2).sum()) for k in cls) * sum(float((gb[k] ** 2).sum()) for k in cls))
    assert ncls / dcls > 0.9, ncls / dcls          # the classification branch (BCE over every anchor) does not depend on the flips
    assert hist["fp8"][4][0].sum() < hist["fp8"][1][0].sum()
    m = Yolov8(eng, nc=nc, size="x", height=640, width=640, max_batch=2, dtype="fp8")
    m.init_weights(1); m.train()
    crit = v8DetectionLoss(m)
    xx = np.random.default_rng(1).random((2, 3, 640, 640), dtype=np.float32)
    bb = {k: v.numpy() for k, v in O.synthetic_batch(2, 640, 640, nc, seed=2).items()}
    for it in range(3):
        m.forward(xx, fetch=False); _, items = crit(None, bb); m.zero_grad(); m.backward(); m.adamw_step([1e-4] * 3)
        assert np.all(np.isfinite(items)), (it, items)
    n8, nb = _f8_launches(eng, lambda: (m.forward(xx, fetch=False), crit(None, bb), m.zero_grad(), m.backward()))
    assert n8 > 40, (n8, nb)
    m.close()
