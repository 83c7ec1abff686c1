"""Per-block entry points (ys_block_create / forward / backward behind the reference's Modules.* surface):
each block handle vs the oracle module of the same constructor arguments -- state_dict names and registration order,
train-mode forward (batch statistics + running-stat update), eval forward, dx and every parameter gradient.
fp32 tolerance 1e-3 (north_star); bf16 runs the same cases at bf16 tolerance on the GPU."""
import numpy as np
import pytest
import torch

from conftest import BACKENDS
from oracle import yolo_oracle as O
from test_model import relerr

# (name, oracle ctor, engine ctor kwargs, c1, H, W)
CASES = {
    "conv3s2": (lambda: O.Conv(8, 16, 3, 2), "Conv", dict(c1=8, c2=16, k=3, s=2), 8, 12, 16),
    "conv1_noact": (lambda: O.Conv(16, 8, 1, 1, act=False), "Conv", dict(c1=16, c2=8, k=1, s=1, act=False), 16, 8, 8),
    "bottleneck": (lambda: O.Bottleneck(16, 16, True, e=0.5), "Bottleneck", dict(c1=16, c2=16, shortcut=True, e=0.5), 16, 8, 8),
    "c2f_n2_sc": (lambda: O.C2f(16, 16, 2, True), "C2f", dict(c1=16, c2=16, n=2, shortcut=True), 16, 8, 12),
    "c2f_n1": (lambda: O.C2f(24, 16, 1, False), "C2f", dict(c1=24, c2=16, n=1, shortcut=False), 24, 8, 8),
    "c3k2_bneck": (lambda: O.C3k2(16, 32, 1, False, 0.5), "C3k2", dict(c1=16, c2=32, n=1, c3k=False, e=0.5), 16, 8, 8),
    "c3k2_c3k": (lambda: O.C3k2(16, 32, 1, True, 0.5), "C3k2", dict(c1=16, c2=32, n=1, c3k=True, e=0.5), 16, 8, 8),
    "sppf": (lambda: O.SPPF(16, 16), "SPPF", dict(c1=16, c2=16), 16, 10, 10),
    "c2psa": (lambda: O.C2PSA(128, 128, 1), "C2PSA", dict(c1=128, c2=128, n=1), 128, 4, 4),
    "proto": (lambda: O.Proto(16, 16, 8), "Proto", dict(c1=16, c_=16, c2=8), 16, 6, 8),
}
EMU_CASES = ["conv3s2", "bottleneck", "c2f_n2_sc", "c3k2_c3k", "sppf", "proto"]


def _randomise(ref, seed):
    torch.manual_seed(seed)
    for mod in ref.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.normal_(0, 0.1)
            mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)


def _block_parity(engine, case, dtype, B, tol_fwd, tol_grad, scale=1):
    from yolosharp_amd import blocks
    make_ref, cls, kw, c1, H, W = CASES[case]
    H, W = H * scale, W * scale
    torch.manual_seed(11)
    ref = make_ref()
    _randomise(ref, 5)
    blk = getattr(blocks, cls)(engine, **kw, height=H, width=W, max_batch=B, dtype=dtype)
    info = blk.tensor_info()
    assert [n for n, s, p in info if p] == [k for k, _ in ref.named_parameters()]
    assert {n: tuple(s) for n, s, p in info} == {k: (tuple(v.shape) if v.dim() else (1,)) for k, v in ref.state_dict().items()}
    blk.load_state_dict({k: v.detach().numpy() for k, v in ref.state_dict().items()})
    x = torch.randn(B, c1, H, W, generator=torch.Generator().manual_seed(3))
    # ---- eval (running statistics)
    blk.eval(); ref.eval()
    with torch.no_grad():
        ry = ref(x)
    y = blk.forward(x.numpy())
    assert y.shape == tuple(ry.shape)
    assert relerr(y, ry) < tol_fwd
    with pytest.raises(RuntimeError):          # backward needs a training-mode forward
        blk.backward(np.zeros_like(y))
    # ---- train: forward, running-stat update, dx and parameter gradients
    blk.train(); ref.train()
    xr = x.clone().requires_grad_(True)
    ry = ref(xr)
    y = blk.forward(x.numpy())
    assert relerr(y, ry) < tol_fwd
    dy = torch.randn(ry.shape, generator=torch.Generator().manual_seed(4))
    ry.backward(dy)
    blk.zero_grad()
    dx = blk.backward(dy.numpy())
    assert relerr(dx, xr.grad) < tol_grad
    g = blk.grads()
    # gradients that are analytically zero (e.g. SPPF.cv1.bn.bias in train mode: a per-channel shift commutes with the max
    # pools and is removed by cv2's batch statistics) are compared on the scale of the largest gradient, not their own noise
    gmax = max(float(p.grad.abs().max()) for _, p in ref.named_parameters())
    def gerr(a, b):
        b = b.numpy()
        return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-3 * gmax))
    worst = max((gerr(g[k], p.grad), k) for k, p in ref.named_parameters())
    assert worst[0] < tol_grad, worst
    sd = blk.state_dict()
    for k, v in ref.state_dict().items():
        if "running" in k or "num_batches" in k:
            assert np.allclose(sd[k], v.numpy(), rtol=1e-3 if dtype == "f32" else 2e-2, atol=1e-4 if dtype == "f32" else 2e-2), k
    # a second backward without a new forward still works and ACCUMULATES (autograd semantics of .grad)
    blk.backward(dy.numpy(), need_dx=False)
    g2 = blk.grads()
    k0 = next(iter(g))
    assert relerr(g2[k0], 2 * g[k0]) < 1e-3 if dtype == "f32" else True
    blk.close()


@pytest.mark.parametrize("case", EMU_CASES)
@pytest.mark.parametrize("backend", ["emu"])
def test_block_parity_f32_emu(engine, backend, case):
    _block_parity(engine, case, "f32", 2, 1e-3, 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("backend", ["gpu"])
def test_block_parity_f32_gpu(engine, backend, case):
    _block_parity(engine, case, "f32", 3, 1e-3, 1e-3, scale=2)


@pytest.mark.gpu
# (SPPF is left to the f32 run: on bf16 activations the 5x5 max pools meet rounding ties, so gradients are routed to different
#  -- equally valid -- positions than in the fp32 oracle and a pointwise dx comparison is meaningless)
@pytest.mark.parametrize("case", ["conv3s2", "bottleneck", "c2f_n2_sc", "c2f_n1", "c3k2_c3k", "proto", "c2psa"])
@pytest.mark.parametrize("backend", ["gpu"])
def test_block_parity_bf16_gpu(engine, backend, case):
    _block_parity(engine, case, "bf16", 4, 4e-2, 8e-2, scale=2)


# ---- bf16 against the ROUNDING-MATCHED oracle (tests/bf16_ref.py: fp32 arithmetic with a bf16 round trip at every point where the engine
# stores a tensor).  This is the tight bf16 parity statement: a block's train-mode forward is bit-identical to that reference up to
# isolated one-ulp flips (summation order inside a convolution), its input gradient is within a few bf16 ulps (the engine adds several
# consumers' contributions into a bf16 buffer one at a time -- 25-65 % of the elements round differently by one ulp -- and BatchNorm
# backward amplifies that through the block), and the parameter gradients (fp32) agree to cosine 0.9998 / 1.5 % of their maximum.
# A whole YOLOv8n at random initialisation amplifies one-ulp flips by ~2x per block (measured: running-mean updates differ by 1e-5 after
# model.2 and 2e-2 after model.21, head outputs by 7 % rms), so tight model-level bf16 bounds do not exist; the block level is where the
# implementation can be pinned.
BF16_CASES = {
    "conv3": (lambda: O.Conv(32, 32, 3, 1), "Conv", dict(c1=32, c2=32, k=3, s=1), 32),
    "conv3s2": (lambda: O.Conv(32, 64, 3, 2), "Conv", dict(c1=32, c2=64, k=3, s=2), 32),
    "conv1_noact": (lambda: O.Conv(64, 32, 1, 1, act=False), "Conv", dict(c1=64, c2=32, k=1, s=1, act=False), 64),
    "bneck_sc": (lambda: O.Bottleneck(32, 32, True, e=0.5), "Bottleneck", dict(c1=32, c2=32, shortcut=True, e=0.5), 32),
    "c2f_sc": (lambda: O.C2f(32, 32, 2, True), "C2f", dict(c1=32, c2=32, n=2, shortcut=True), 32),
    "c2f": (lambda: O.C2f(48, 32, 1, False), "C2f", dict(c1=48, c2=32, n=1, shortcut=False), 48),
    "sppf": (lambda: O.SPPF(32, 32), "SPPF", dict(c1=32, c2=32), 32),
    "c2f_wide": (lambda: O.C2f(256, 256, 1, True), "C2f", dict(c1=256, c2=256, n=1, shortcut=True), 256),   # blocked-GEMM kernels (>= 128 channels)
    # halo-patch kernel (conv_halo.h) in every epilogue class: 160 -> 160 3x3 Bottleneck convolutions of a YOLOv8x-like C2f -- training forward (statistics),
    # dgrad with the fused BN-backward reduction, dgrad accumulating into a view that already holds the shortcut's gradient; 2.5 chunks of 64 channels
    "c2f_halo": (lambda: O.C2f(320, 320, 1, True), "C2f", dict(c1=320, c2=320, n=1, shortcut=True), 320),
}


def _block_bf16_matched(engine, case, B, H, W):
    from yolosharp_amd import blocks
    import bf16_ref as R
    make_ref, cls, kw, c1 = BF16_CASES[case]
    torch.manual_seed(11)
    ref = make_ref()
    _randomise(ref, 5)
    blk = getattr(blocks, cls)(engine, **kw, height=H, width=W, max_batch=B, dtype="bf16")
    blk.load_state_dict({k: v.detach().numpy() for k, v in ref.state_dict().items()})
    x = torch.randn(B, c1, H, W, generator=torch.Generator().manual_seed(3))
    blk.train(); ref.train()
    y = blk.forward(x.numpy())
    xr = R.bf16r(x).requires_grad_(True)
    with R.bf16_storage(ref):
        ry = ref(R.rste(xr))
    rya = ry.detach().numpy()
    flips = float((y != rya).mean())
    single_layer = cls == "Conv"                # a flipped value moves what is computed from it: deeper blocks carry more (GPU, 80 x 80, B = 8: C2f(n=2) 1.0 %)
    flip_cap = 5e-3 if single_layer else (1e-1 if c1 >= 128 else 3e-2)   # K = 2304 accumulations flip more often (256-channel C2f on the GPU: 5.8 %)
    if case == "c2f_halo":
        flip_cap = 0.2     # the halo-patch kernel walks K as (64-channel chunk, tap, unit-permuted K-step): not the oracle's order even on the interpreter (11 % measured)
    assert flips <= flip_cap, ("forward differs from the rounding-matched oracle in more than isolated flips", flips)
    R.check_elem(y, rya, case + " forward", max_out=flip_cap, out_mult=4.0)      # and a flip is one ulp, not garbage
    dy = R.bf16r(torch.randn(ry.shape, generator=torch.Generator().manual_seed(4)))
    blk.zero_grad()
    dx = blk.backward(dy.numpy())
    ry.backward(dy)
    single = cls == "Conv"
    R.check_elem(dx, xr.grad.numpy(), case + " dx", max_out=2e-3 if single else 0.15, out_mult=4.0 if single else 16.0)
    g = blk.grads()
    gmax = max(float(p.grad.abs().max()) for _, p in ref.named_parameters())
    for k, p in ref.named_parameters():
        a, b = g[k].astype(np.float64).ravel(), p.grad.numpy().astype(np.float64).ravel()
        assert np.abs(a - b).max() <= 1.5e-2 * np.abs(b).max() + 2e-3 * gmax, (k, np.abs(a - b).max(), np.abs(b).max(), gmax)
        if np.abs(b).max() > 1e-2 * gmax:
            assert float(a @ b) / np.sqrt(float(a @ a) * float(b @ b)) > 0.9998, k
    sd = blk.state_dict()
    for k, v in ref.state_dict().items():
        if "running" in k:
            assert np.allclose(sd[k], v.numpy(), rtol=2e-4, atol=2e-5), k     # statistics of bit-identical y: fp32 summation order only
    blk.close()


@pytest.mark.parametrize("case", ["conv3", "conv3s2", "conv1_noact", "bneck_sc", "c2f_sc", "c2f", "sppf", "c2f_halo"])
@pytest.mark.parametrize("backend", ["emu"])
def test_block_bf16_rounding_matched_emu(engine, backend, case):
    if case == "c2f_halo":
        _block_bf16_matched(engine, case, 1, 20, 18)     # 2 x 2 ragged 16 x 16 tiles
    else:
        _block_bf16_matched(engine, case, 2, 24, 24)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(BF16_CASES))
@pytest.mark.parametrize("backend", ["gpu"])
def test_block_bf16_rounding_matched_gpu(engine, backend, case):
    """The same statement on the MI355X at sizes where the persistent grids walk several tiles per workgroup (80 x 80, B = 8; the
    256-channel case at 20 x 20 is a P5 C2f of YOLOv8n)."""
    if case == "c2f_wide":
        _block_bf16_matched(engine, case, 8, 20, 20)
    elif case == "c2f_halo":
        _block_bf16_matched(engine, case, 8, 80, 80)     # 200 tiles x 1 channel tile: one tile per workgroup ...
        with engine.options(HALO_MAX_GRID=48):           # ... and the tile stream (4-5 tiles per workgroup: no per-tile prologue, operands of the next tile land under the epilogue)
            _block_bf16_matched(engine, case, 8, 80, 80)
    else:
        _block_bf16_matched(engine, case, 8, 80, 80)


@pytest.mark.parametrize("backend", ["emu"])
def test_c2psa_bf16_emu(engine, backend):
    """bf16 C2PSA (attention forward / backward, depthwise pe conv) against the fp32 oracle module; the fp32 runs above are the
    tight parity statement."""
    _block_parity(engine, "c2psa", "bf16", 2, 4e-2, 8e-2)


@pytest.mark.parametrize("backend", ["emu"])
def test_block_errors(engine, backend):
    """Unsupported geometry fails loudly with the reason (no silent substitution); model-only entry points reject block handles."""
    from yolosharp_amd import blocks
    with pytest.raises(RuntimeError, match="multiple of"):
        blocks.C2f(engine, 6, 16, 1, height=8, width=8, dtype="f32")
    with pytest.raises(RuntimeError, match="c1 == c2"):
        blocks.SPPF(engine, 16, 32, height=8, width=8, dtype="f32")
    with pytest.raises(RuntimeError, match="k=5"):
        blocks.Conv(engine, 8, 8, 5, 1, height=8, width=8, dtype="f32")
    blk = blocks.Conv(engine, 8, 8, 1, 1, height=4, width=4, dtype="f32")
    with pytest.raises(RuntimeError, match="block"):
        blk.get_output("boxes")
    with pytest.raises(RuntimeError, match="needs a training-mode"):
        blk.backward(np.zeros((0, 8, 4, 4), np.float32))
    blk.close()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("hw", [(4, 4), (10, 10), (9, 7), (20, 20)])
def test_c2psa_mfma_attention_tracks_scalar_kernels(engine, backend, hw):
    """Round 5: attn_fwd_mfma_kernel / attn_bwd_q_mfma_kernel (S = Q K^T, O = P V, dP = dO V^T, dq = dS K on the matrix cores; P / dS enter the second product rounded to
    bf16) against the scalar kernels (ATTN_MFMA=0) on the same C2PSA block: output, input gradient and every parameter gradient within bf16 rounding of the scalar
    result -- relative L2 <= 3e-2, max |diff| <= 6 % of the tensor's scale.  Measured against the fp32 oracle (4 x 4 / 10 x 10 tokens): both forms sit at the same
    distance from it (y 0.0082-0.0084 vs 0.0080-0.0083, dx 0.0110-0.0119 both, worst parameter gradient qkv.bn.weight 0.022-0.028 vs 0.020-0.027) and 0.7-2.1 % from each
    other: the bf16 storage of q / k / v / dO, not the rounding of P, sets the error.  Token counts below, at and across the 16- / 32-token tile edges; 400 tokens = the production shape."""
    from yolosharp_amd.blocks import C2PSA
    H, W = hw
    if backend == "emu" and H * W > 128:
        pytest.skip("400-token case: GPU only (the interpreter needs minutes)")
    B, C_ = 2, 128
    rng = np.random.default_rng(5)
    x = rng.standard_normal((B, C_, H, W), dtype=np.float32)
    dy = rng.standard_normal((B, C_, H, W), dtype=np.float32)
    res = []
    for mf in (1, 0):
        with engine.options(ATTN_MFMA=mf):
            m = C2PSA(engine, C_, C_, 1, height=H, width=W, max_batch=B, dtype="bf16")
            m.init_weights(7); m.train()
            y = m.forward(x); m.zero_grad(); dx = m.backward(dy)
            res.append((y, dx, m.grads()))
            m.close()
    (ya, da, ga), (yb, db, gb) = res
    def close(a, b, what):
        assert np.isfinite(a).all(), what
        sc = float(np.abs(b).max()) + 1e-12
        rel = float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-12))
        assert rel <= 3e-2 and float(np.abs(a - b).max()) <= 6e-2 * sc, (what, rel, float(np.abs(a - b).max()), sc)
    close(ya, yb, "y"); close(da, db, "dx")
    for k in ga:
        close(ga[k], gb[k], k)
