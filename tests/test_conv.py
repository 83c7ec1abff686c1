"""Conv unit parity (Convs.cs:36-62, Head.cs:47-50 plain Conv2d) against plain PyTorch fp32 ops, forward and backward."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import BACKENDS

FWD_CASES = [
    # B, Cin, H, W, Cout, k, s, bn, bias, act, train
    (2, 16, 12, 12, 32, 3, 1, True, False, True, True),
    (2, 3, 16, 16, 16, 3, 2, True, False, True, True),      # stem: Cin=3 padded to the fragment width
    (1, 48, 8, 8, 24, 1, 1, True, False, True, True),       # C2f cv2 of v8m-like widths (Cout not a multiple of 16)
    (2, 80, 6, 6, 80, 3, 1, True, False, True, False),      # eval: BN folded into the conv epilogue
    (2, 64, 5, 5, 80, 1, 1, False, True, False, True),      # Detect cv3[i][2]: plain conv + bias
    (1, 16, 9, 11, 16, 3, 2, True, False, True, True),      # odd spatial size, stride 2
    (1, 256, 4, 4, 128, 1, 1, True, False, False, True),    # SPPF.cv1: no activation (Block.cs:257)
    (1, 272, 4, 4, 16, 1, 1, True, False, True, True),      # K tail: Cin not a multiple of the k-step
    (2, 64, 5, 5, 7, 1, 1, False, True, False, True),       # nc not a multiple of 4 (masked stores)
    # multi-tile images with ragged tile edges (whole-Cin LDS patch kernel): resident and streamed weights, stride 2,
    # several output-channel tiles, Cin/8 odd (wider patch pitch)
    (2, 32, 20, 40, 48, 3, 1, True, False, True, True),
    (1, 128, 12, 20, 144, 3, 1, True, False, True, True),
    (2, 16, 18, 36, 16, 3, 2, True, False, True, True),
    (1, 40, 23, 17, 80, 3, 1, True, False, True, False),
    (3, 64, 21, 21, 64, 3, 2, False, True, False, True),
    # streamed weights through the LDS-DMA ring (conv.hip DMAW): training epilogue, a last weight group that is only partly real
    # (K = 720 -> 11.25 groups; K = 648 -> 10.125), an output-channel column whose rows run past Cout (96 = 64 + 32)
    (2, 80, 19, 13, 80, 3, 1, True, False, True, True),
    (1, 72, 10, 37, 96, 3, 1, True, False, True, True),
    # blocked-GEMM kernel of the wide layers (conv_gemm.hip; conftest lowers its size gates): every tile shape, ragged M and N,
    # K not a multiple of the K-tile (taps straddle tiles), stride 2, 1x1, eval epilogue
    (1, 128, 9, 11, 80, 3, 2, True, False, True, True),     # 256x80 tile
    (2, 160, 7, 9, 64, 3, 1, True, False, True, True),      # 256x64 tile, K = 1440
    (1, 256, 13, 10, 128, 1, 1, True, False, True, True),   # 128x128 tile, two M tiles
    (1, 136, 17, 9, 320, 3, 1, True, False, True, False),   # 128x160 tile x 2 channel tiles, eval
    (2, 128, 6, 6, 80, 1, 1, False, True, False, True),     # plain conv + bias
    # halo-patch kernel of the 3x3 stride-1 wide layers (conv_gemm.hip conv_halo_kernel): 16 x 16 pixel tiles with ragged edges in both directions,
    # several images / channel tiles, a last 64-channel chunk that is only partly real (Cin = 160, 136, 72), odd and even chunk counts, Cout past the
    # channel tile, eval epilogue, plain conv + bias
    (1, 160, 20, 33, 160, 3, 1, True, False, True, True),   # 2 x 3 tiles (ragged), 2.5 chunks, 160-wide channel tile
    (2, 128, 17, 16, 128, 3, 1, True, False, True, True),   # 128-wide channel tile, two images, two chunks
    (1, 136, 14, 40, 144, 3, 1, True, False, True, True),   # partial last chunk, Cout 144 of 160
    (1, 64, 33, 18, 320, 3, 1, True, False, True, False),   # one chunk, two channel tiles, eval BN + SiLU
    (2, 72, 15, 15, 128, 3, 1, False, True, False, True),   # plain conv + bias, one tile per image
]


def ref_forward(x, w, k, s, bn, bias, act, train, dtype):
    if dtype == "bf16":
        x, w = x.bfloat16().float(), w.bfloat16().float()
    y = F.conv2d(x, w, bias, stride=s, padding=k // 2)
    if bn is not None:
        if dtype == "bf16":
            y = y.bfloat16().float()
        y = F.batch_norm(y, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], training=train, momentum=0.03, eps=1e-3)
    if act:
        y = F.silu(y)
    return y


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", range(len(FWD_CASES)))
def test_conv_bn_act_forward(backend, engine, dtype, case):
    B, Cin, H, W, Cout, k, s, has_bn, has_bias, act, train = FWD_CASES[case]
    g = torch.Generator().manual_seed(case)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.2
    bn = None
    if has_bn:
        bn = {"weight": torch.rand(Cout, generator=g) + 0.5, "bias": torch.randn(Cout, generator=g) * 0.1,
              "running_mean": torch.randn(Cout, generator=g) * 0.1, "running_var": torch.rand(Cout, generator=g) + 0.5}
    bias = torch.randn(Cout, generator=g) if has_bias else None
    bn_np = {kk: v.numpy().copy() for kk, v in bn.items()} if bn else None
    ref = ref_forward(x, w, k, s, {kk: v.clone() for kk, v in bn.items()} if bn else None, bias, act, train, dtype)
    bn_ref = {kk: v.clone() for kk, v in bn.items()} if bn else None
    if bn and train:
        ref_forward(x, w, k, s, bn_ref, bias, act, train, dtype)     # updates running stats in place
    y = engine.conv_bn_act(x.numpy(), w.numpy(), k, s, bn=bn_np, bias=None if bias is None else bias.numpy(), act=act,
                           training=train, dtype=dtype)
    ref = ref.numpy()
    scale = np.abs(ref).max()
    if dtype == "f32":    # north-star tolerance 1e-3 (measured ~1e-6)
        assert np.abs(y - ref).max() <= 1e-3 * max(scale, 1.0), (np.abs(y - ref).max(), scale)
    else:
        # bf16, against the rounding-matched reference (operands and the raw conv output rounded where the engine rounds them): PER ELEMENT
        # |y - ref| <= 2^-7 |ref| + 2^-8 rms(ref) -- the result's own rounding plus one ulp at the tensor's scale for a flipped upstream
        # value.  (Rounds 1-3 allowed 2e-2 of the GLOBAL maximum, which a wrong halo column of small outputs would pass.)
        from bf16_ref import check_elem
        # (one element in 10^4 may sit up to 2x the bound: the fp32 summation order inside the convolution differs from the reference's -- the halo-patch kernel
        # walks K as (64-channel chunk, tap) -- and a flipped bf16 rounding of the raw output is then scaled by the batch statistics)
        check_elem(y, ref, "conv case %d" % case, max_out=1e-4, out_mult=2.0)
    if bn and train:
        rtol = 1e-4 if dtype == "f32" else 2e-2
        assert np.allclose(bn_np["running_mean"], bn_ref["running_mean"].numpy(), rtol=rtol, atol=rtol)
        assert np.allclose(bn_np["running_var"], bn_ref["running_var"].numpy(), rtol=rtol, atol=rtol)


BWD_CASES = [(2, 16, 8, 8, 32, 3, 1), (2, 16, 8, 8, 16, 1, 1), (1, 32, 9, 7, 80, 3, 2), (2, 3, 8, 8, 16, 3, 2),
             (2, 64, 6, 6, 80, 3, 2), (2, 80, 4, 4, 80, 1, 1), (1, 96, 6, 6, 64, 1, 1),
             # multi-tile images, ragged tile edges, several (cout, cin) channel tiles (LDS-tile wgrad kernel)
             (2, 32, 20, 40, 32, 3, 1), (1, 128, 12, 20, 144, 3, 1), (1, 256, 5, 5, 64, 1, 1), (2, 16, 18, 36, 16, 3, 2),
             (3, 48, 13, 17, 32, 1, 1),
             # dgrad of a streamed-weight layer (ring + both K-steps of a group per LDS wait), ragged tiles
             (2, 80, 11, 9, 80, 3, 1),
             # dgrad through the blocked-GEMM kernel (Cout of the layer = K of its dgrad): stride-1 3x3, 1x1, stride-2 phases
             (1, 64, 11, 13, 160, 3, 1), (2, 80, 9, 9, 256, 1, 1), (1, 64, 14, 10, 128, 3, 2),
             # wgrad through the blocked-GEMM kernel (conv_wgrad_gemm.hip): 160 / 128 tiles and both mixes, ragged channels, stride 2, 1x1
             (1, 160, 9, 11, 320, 3, 2), (2, 256, 6, 7, 128, 1, 1), (1, 136, 10, 9, 160, 3, 1), (1, 256, 7, 9, 152, 3, 1),
             # dgrad through the halo-patch kernel (gradient channels = its K): 160 -> 160 over ragged 16-row tiles, 128 -> 128 two images
             (1, 160, 19, 21, 160, 3, 1), (2, 128, 9, 17, 128, 3, 1)]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", range(len(BWD_CASES)))
def test_conv_backward(backend, engine, dtype, case):
    """dgrad (gather form, stride-2 included) and wgrad (deterministic split-K) vs torch autograd."""
    import ctypes as C
    B, Cin, H, W, Cout, k, s = BWD_CASES[case]
    g = torch.Generator().manual_seed(10 + case)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.2
    if dtype == "bf16":
        x, w = x.bfloat16().float(), w.bfloat16().float()
    x.requires_grad_(True); w.requires_grad_(True)
    y = F.conv2d(x, w, None, stride=s, padding=k // 2)
    dy = torch.randn(y.shape, generator=g)
    if dtype == "bf16":
        dy = dy.bfloat16().float()
    y.backward(dy)
    dx = np.zeros(x.shape, np.float32); dw = np.zeros(w.shape, np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    xn, wn, dyn = x.detach().numpy().copy(), w.detach().numpy().copy(), dy.numpy().copy()
    from yolosharp_amd import _lib
    _lib.check(engine.lib, engine.lib.ys_conv_bwd(engine.ctx, 1 if dtype == "bf16" else 0, vp(xn), B, Cin, H, W, vp(wn), Cout, k, s,
                                                 vp(dyn), vp(dx), vp(dw)))
    if dtype == "f32":
        assert np.abs(dx - x.grad.numpy()).max() <= 1e-4 * np.abs(x.grad.numpy()).max()
    else:                                    # bf16: dx is rounded to bf16 on store (per-element bound, as in the forward test); dw stays fp32
        from bf16_ref import check_elem
        check_elem(dx, x.grad.numpy(), "dgrad case %d" % case)
    assert np.abs(dw - w.grad.numpy()).max() <= 1e-4 * np.abs(w.grad.numpy()).max()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [15, 17, 18])
def test_wgrad_gemm_short_k_tile(backend, engine, case, monkeypatch):
    """The 32-pixel K-tile variant of the blocked-GEMM wgrad kernel (chosen when two 64-pixel stages do not fit the LDS share)."""
    with engine.options(WGEMM_KT=32):
        test_conv_backward(backend, engine, "bf16", case)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("grid", ["1", "2", "4"])
def test_halo_kernel_tile_stream(backend, engine, grid, monkeypatch):
    """conv_halo_kernel's workgroups walk their tiles as ONE tap stream: the last chunk of a tile requests the next tile's first patch and taps (no per-tile
    prologue), patch buffers alternate across the tile boundary for odd chunk counts.  Oracle-sized maps have fewer tiles than the chip has CUs, so the test caps the grid:
    1, 2 or 4 workgroups for 6 / 2 tiles (odd and even chunk counts, ragged edges), forward (statistics + BN) and dgrad."""
    with engine.options(HALO_MAX_GRID=int(grid)):
        for case in (len(FWD_CASES) - 5, len(FWD_CASES) - 4, len(FWD_CASES) - 2):
            test_conv_bn_act_forward(backend, engine, "bf16", case)
        test_conv_backward(backend, engine, "bf16", len(BWD_CASES) - 2)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("grid", ["0", "1", "3"])
def test_halo_kernel_16x8_tiles(backend, engine, grid):
    """conv_halo_kernel with 16 x 8-pixel tiles (MR = 4: patch 18 x 10, 25 request pieces per chunk, one patch request per tap, requests in every MFMA slot) -- what the
    plan picks for maps that 16 x 16 tiles cover badly (40 x 40).  HALO_MR4=2 forces it on the halo test shapes: forward (statistics + BN + SiLU), dgrad with and
    without accumulation / fused reduction, default grid and workgroups that walk several tiles (1 and 3 workgroups)."""
    with engine.options(HALO_MR4=2, HALO_MAX_GRID=int(grid)):
        for case in range(len(FWD_CASES) - 5, len(FWD_CASES)):
            test_conv_bn_act_forward(backend, engine, "bf16", case)
        for case in (len(BWD_CASES) - 2, len(BWD_CASES) - 1):
            test_conv_backward(backend, engine, "bf16", case)


@pytest.mark.parametrize("backend", BACKENDS)
def test_wide_layers_run_the_blocked_gemm_kernel(backend, engine, tmp_path):
    """The per-launch profile names the kernel a layer ran on: wide layers -> conv_gemm_kernel for forward and dgrad; narrower layers and the
    phase convolutions of a stride-2 dgrad with < 160 gradient channels stay on the whole-Cin patch kernel."""
    import ctypes as C
    from yolosharp_amd import _lib
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    g = torch.Generator().manual_seed(0)
    engine.kernel_profile(True)
    for (cin, cout, k, s) in [(128, 160, 3, 1), (32, 160, 3, 1), (64, 128, 3, 2)]:
        x = torch.randn(1, cin, 12, 12, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) * 0.1
        dy = torch.randn(1, cout, 12 // s, 12 // s, generator=g)
        engine.conv_bn_act(x.numpy(), w.numpy(), k, s, bn=None, bias=np.zeros(cout, np.float32), act=False, dtype="bf16")
        dx = np.zeros(x.shape, np.float32); dw = np.zeros(w.shape, np.float32)
        xn, wn, dyn = x.numpy().copy(), w.numpy().copy(), dy.numpy().copy()
        _lib.check(engine.lib, engine.lib.ys_conv_bwd(engine.ctx, 1, vp(xn), 1, cin, 12, 12, vp(wn), cout, k, s, vp(dyn), vp(dx), vp(dw)))
    path = tmp_path / "launches.csv"
    engine.kernel_profile_dump(path)
    engine.kernel_profile(False)
    labels = [l.split(",")[1] for l in open(path).read().splitlines()[1:] if l.startswith("conv_igemm")]
    assert sum(l.startswith("halo k33 s1 div1 cin128 cout160") for l in labels) == 1          # forward, 128 -> 160: the halo-patch form of the 3x3 stride-1 layers
    assert sum(l.startswith("halo k33 s1 div1 cin160 cout128") for l in labels) == 1          # its dgrad
    assert sum(l.startswith("gemm k33 s1 div1 cin160 cout32") for l in labels) == 0           # Cout < 64: not eligible
    assert sum(l.startswith("p2 k33 s1 div1 cin32 cout160") for l in labels) == 1             # narrow forward: patch kernel
    # stride-2 dgrad of 64 -> 128 (128 gradient channels < the kernel's 160-channel gate): all four phase convolutions on the patch kernel, as ONE grouped
    # launch when the group planner accepts the shapes (its label starts with the first member, the 2x2-tap phase), else one launch per phase
    assert not [l for l in labels if l.startswith("gemm k") and "cin128 cout64" in l]
    ph = [l for l in labels if "cin128 cout64" in l and l.split()[0] in ("p2", "p2grp4")]
    assert (len(ph) == 1 and ph[0].startswith("p2grp4 k22")) or sorted(l.split()[1] for l in ph) == ["k11", "k12", "k21", "k22"], ph
    wl = [l.split(",")[1] for l in open(path).read().splitlines()[1:] if l.startswith("conv_wgrad")]
    assert sum(l.startswith("wgemm k3 s1 cin128 cout160") for l in wl) == 1                   # both sides >= 128 channels
    assert sum(l.startswith("wgemm") for l in wl) == 1 and len(wl) == 3                        # the narrow layers keep the 9-wave kernel


@pytest.mark.parametrize("backend", [pytest.param("gpu", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("case", [(16, 800, 160, 160, 320, 1), (2, 400, 320, 320, 160, 1), (4, 64, 160, 160, 64, 3),
                                  # round 4: the two layers on which the packed-FP32 statistics of an unrolled epilogue differed in 3-6 of 6 reruns
                                  # (128 x 160 tile of the blocked-GEMM kernel, ~10 tiles per workgroup); and a YOLOv11m-seg (config 4) shape
                                  (16, 400, 320, 320, 160, 1), (16, 160, 320, 320, 160, 3), (32, 256, 80, 80, 256, 3)])
def test_wide_bn_layer_reruns_are_bit_identical(backend, engine, case):
    """Run-to-run bit identity of Conv + BN(train) + SiLU at sizes where a workgroup of the persistent grid walks several tiles
    (blocked-GEMM kernel: K not a multiple of the K-tile; the whole-Cin patch kernel for the third case).  Round 3: a batched form
    of the staged epilogue's store loop passed every parity test and produced BatchNorm sums that differed from run to run on
    exactly these layers (one pixel's vectors in ~10^6 read stale) -- caught by test_c5_v8x_1280_bs16_fp8_train_steps, which needs
    the whole YOLOv8x graph and fails on every run of such a build; this is the per-layer form (tools/dev/determinism_layer.py is the
    triage version).  On the build that had the defect the B = 16 case differed in 1 of 6 .. 11 of 11 reruns depending on the box and
    the B <= 4 cases in none (a workgroup has to walk ~6 tiles), so the first case runs 12 times; the model-level test stays the
    definitive one."""
    B, Cin, H, W, Cout, k = case
    rng = np.random.default_rng(3)
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) / np.sqrt(Cin * k * k)).astype(np.float32)
    outs = []
    for _ in range(10 if B >= 16 else 4):
        bn = {"weight": np.ones(Cout, np.float32), "bias": np.zeros(Cout, np.float32),
              "running_mean": np.zeros(Cout, np.float32), "running_var": np.ones(Cout, np.float32)}
        y = engine.conv_bn_act(x, w, k, 1, bn=bn, act=True, training=True, dtype="bf16")
        outs.append((y, bn["running_mean"].copy(), bn["running_var"].copy()))
    for y, rm, rv in outs[1:]:
        assert np.array_equal(rm, outs[0][1]) and np.array_equal(rv, outs[0][2]), "batch statistics differ between reruns"
        assert np.array_equal(y, outs[0][0]), "outputs differ between reruns"


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gpu"])
def test_halo_kernel_against_blocked_gemm_on_random_shapes(backend, engine):
    """conv_halo_kernel (GEMM_HALO=1) and conv_gemm_kernel (GEMM_HALO=0) on the same random 3x3 stride-1 problems -- ragged maps, one to many tiles per workgroup,
    partial last chunks (Cin mod 64 in 1..32 and above), 128- and 160-wide channel tiles, several batch sizes: forward with training-mode BatchNorm + SiLU (the
    statistics path), then dx / dw through ys_conv_bwd.  Two different summation orders of the same bf16 products: outputs within bf16 rounding of each other
    (max |diff| <= 2^-7 of the output scale), and each rerun of the halo launch bit-identical to itself (the races of round 5 showed as run-to-run differences)."""
    import ctypes as C
    from yolosharp_amd import _lib
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(11)
    shapes = []
    for _ in range(14):
        cin = int(rng.choice([64, 72, 80, 96, 128, 160, 192, 320]))
        cout = int(rng.choice([64, 80, 128, 144, 160, 320]))
        shapes.append((int(rng.integers(1, 5)), cin, int(rng.integers(16, 70)), int(rng.integers(16, 70)), cout))
    shapes += [(2, 160, 80, 80, 160), (1, 320, 48, 33, 320)]
    engine.kernel_profile(True)
    for (B, cin, H, W, cout) in shapes:
        x = rng.standard_normal((B, cin, H, W), dtype=np.float32)
        w = (rng.standard_normal((cout, cin, 3, 3), dtype=np.float32) / np.sqrt(9 * cin)).astype(np.float32)
        dy = rng.standard_normal((B, cout, H, W), dtype=np.float32)
        res = {}
        for halo in (1, 0, 1, 2, 2):           # 2: the halo kernel with 16 x 8-pixel tiles forced (twice: bit-identical reruns)
            with engine.options(GEMM_HALO=1 if halo else 0, HALO_MR4=2 if halo == 2 else 1, HALO_MIN_FILL=1, GEMM_MIN_M=1, GEMM_MIN_CIN=64):
                bn = {"weight": np.ones(cout, np.float32), "bias": np.zeros(cout, np.float32), "running_mean": np.zeros(cout, np.float32), "running_var": np.ones(cout, np.float32)}
                y = engine.conv_bn_act(x, w, 3, 1, bn=bn, act=True, training=True, dtype="bf16")
                y = y[0] if isinstance(y, tuple) else y
                dx = np.zeros(x.shape, np.float32); dw = np.zeros(w.shape, np.float32)
                _lib.check(engine.lib, engine.lib.ys_conv_bwd(engine.ctx, 1, vp(x), B, cin, H, W, vp(w), cout, 3, 1, vp(dy), vp(dx), vp(dw)))
            key = {0: "gemm", 1: "halo", 2: "halo8"}[halo]
            if key in res:                     # second halo run: bit-identical
                assert np.array_equal(res[key][0], y) and np.array_equal(res[key][1], dx), (B, cin, H, W, cout)
            res[key] = (np.asarray(y).copy(), dx.copy())
        for key, i, nm in (("halo", 0, "y"), ("halo", 1, "dx"), ("halo8", 0, "y8"), ("halo8", 1, "dx8")):
            a, b = res[key][i], res["gemm"][i]
            assert np.isfinite(a).all(), (nm, B, cin, H, W, cout)
            scale = float(np.abs(b).max()) + 1e-6
            assert float(np.abs(a - b).max()) <= scale * 2.0 ** -7 + 1e-6, (nm, B, cin, H, W, cout, float(np.abs(a - b).max()), scale)
    import os, tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "launches.csv")
        engine.kernel_profile_dump(path)
        labels = [l.split(",")[1] for l in open(path).read().splitlines()[1:] if l.startswith("conv_igemm")]
    engine.kernel_profile(False)
    n_halo = sum(l.startswith("halo k33") for l in labels)
    assert n_halo >= 48 and sum("tile8x16" in l for l in labels) >= 20, (n_halo, len(labels))     # the comparison is not gemm against gemm: most of these shapes (forward and / or dgrad) are halo-eligible
