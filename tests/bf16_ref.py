"""Rounding-matched bf16 reference (TEST INFRASTRUCTURE): the oracle's YOLOv8 graph evaluated in fp32 arithmetic with a bf16 round-trip
at every point where the engine's bf16 path stores a tensor -- so that a bf16 parity test compares like with like instead of granting
the engine "2-5 % of the global maximum" against an fp32 reference.

Where the engine rounds (csrc/conv.hip, conv_epi.h, elementwise.hip):
  * convolution operands: the input tensor and the weight shadow are bf16, products accumulate in fp32 (MFMA);
  * the raw convolution output y is stored bf16 BEFORE the batch statistics are taken (conv_epi.h: statistics of the rounded values);
  * BatchNorm + SiLU run in fp32 on y and the result z is stored bf16; a Bottleneck shortcut is added in the same pass (one rounding of
    x + z, elementwise.hip bn_act_apply);
  * the Detect towers' last 1x1 convolutions add their bias to the fp32 accumulators and store bf16 (`pd`, `ps`);
  * backward: every gradient tensor that mirrors one of those buffers is bf16 (dz, dy); weight / BatchNorm parameter gradients are fp32.
Torch's autograd computes the same chain when each stored tensor passes through `_RoundSTE` (forward: round to bf16; backward: round
the incoming gradient to bf16).  What is NOT matched: summation order inside a convolution / a statistic, and the order in which
several consumers' contributions are added into one gradient buffer (the engine rounds after every accumulate, autograd sums in fp32
and this reference rounds once) -- both show up as isolated one-ulp flips, which is what the per-element tolerance below allows.

Cites: Modules/Convs.cs:36-62 (Conv), Modules/Block.cs:572-608 (Bottleneck), Modules/Head.cs:35-53,71-106 (Detect).
"""
import contextlib

import numpy as np
import torch
import torch.nn.functional as F

from oracle import yolo_oracle as O


def bf16r(t):
    return t.bfloat16().float()


class _RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return bf16r(x)

    @staticmethod
    def backward(ctx, g):
        return bf16r(g)


rste = _RoundSTE.apply


def _conv_forward(self, x, fuse_residual=None):
    w = bf16r(self.conv.weight)
    y = F.conv2d(x, w, None, self.conv.stride, self.conv.padding, self.conv.dilation, self.conv.groups)
    y = rste(y)                                   # raw conv output stored bf16; the batch statistics see the rounded values
    z = self.act(self.bn(y))
    if fuse_residual is not None:
        z = fuse_residual + z                     # Bottleneck shortcut inside the apply pass: one rounding of the sum
    return rste(z)


def _bottleneck_forward(self, x):
    h = self.cv1(x)
    if self.add:
        return _conv_forward(self.cv2, h, fuse_residual=x)
    return self.cv2(h)


def _plain_conv_forward(self, x):
    """nn.Conv2d with bias at the end of a Detect tower (Head.cs:47-48): bias on the fp32 accumulators, bf16 store."""
    return rste(F.conv2d(x, bf16r(self.weight), self.bias, self.stride, self.padding))


def _attention_forward(self, x):
    """C2PSA attention (Block.cs:719-810) with the engine's storage points (csrc/model.hip add_c2psa, csrc/attn_dw.hip): q / k / v are the
    stored (bf16) qkv output; softmax probabilities are rounded to bf16 for the second product (the MFMA kernels' P operand); the attention
    output is stored bf16; `attention + pe(v)` is one rounding inside pe's BatchNorm + SiLU pass.  Returns the INPUT of proj: proj carries
    the PSABlock shortcut in its own apply pass (see _psablock_forward)."""
    B, C, H, W = x.shape
    N = H * W
    qkv = self.qkv(x)
    q, k, v = qkv.view(B, self.num_heads, self.key_dim * 2 + self.head_dim, N).split([self.key_dim, self.key_dim, self.head_dim], dim=2)
    attn = rste(((q.transpose(-2, -1) @ k) * self.scale).softmax(dim=-1))
    a = rste((v @ attn.transpose(-2, -1)).view(B, C, H, W))
    return _conv_forward(self.pe, v.reshape(B, C, H, W), fuse_residual=a)


def _psablock_forward(self, x):
    h = _attention_forward(self.attn, x)
    x = _conv_forward(self.attn.proj, h, fuse_residual=x if self.add else None)
    f = self.ffn[0](x)
    return _conv_forward(self.ffn[1], f, fuse_residual=x if self.add else None)


def _proto_forward(self, x):
    """Proto (Block.cs:51-84): the ConvTranspose2d adds its bias to the fp32 accumulators and stores bf16."""
    u = self.upsample
    up = rste(F.conv_transpose2d(self.cv1(x), bf16r(u.weight), u.bias, u.stride, u.padding))
    return self.cv3(self.cv2(up))


@contextlib.contextmanager
def bf16_storage(model):
    """Inside the context, `model(x)` (an oracle Yolov8 / Yolov11 detect or segment graph) follows the engine's bf16 storage points."""
    saved = (O.Conv.forward, O.Bottleneck.forward, O.PSABlock.forward, O.Proto.forward)
    O.Conv.forward = _conv_forward
    O.Bottleneck.forward = _bottleneck_forward
    O.PSABlock.forward = _psablock_forward
    O.Proto.forward = _proto_forward
    patched = []
    for mod in model.modules():
        if isinstance(mod, O.Detect):
            for seq in list(mod.cv2) + list(mod.cv3) + list(getattr(mod, "cv4", [])):
                last = seq[-1]
                if isinstance(last, torch.nn.Conv2d):
                    last.forward = _plain_conv_forward.__get__(last, type(last))
                    patched.append(last)
    try:
        yield
    finally:
        O.Conv.forward, O.Bottleneck.forward, O.PSABlock.forward, O.Proto.forward = saved
        for last in patched:
            del last.forward


def forward_bf16(model, x):
    with bf16_storage(model):
        return model(bf16r(x))


def elem_bound(ref, k_rel=2.0 ** -7, k_rms=2.0 ** -8):
    """Per-element bound |y - ref| <= 2^-7 |ref| + 2^-8 rms(ref): two bf16 ulps of the element itself plus one ulp at the tensor's scale
    (a one-ulp flip of an upstream stored value moves small outputs by an amount set by the tensor's scale, not by their own size)."""
    ref = np.asarray(ref, np.float64)
    return k_rel * np.abs(ref) + k_rms * np.sqrt(np.mean(ref * ref))


def check_elem(y, ref, what="", k_rel=2.0 ** -7, k_rms=2.0 ** -8, max_out=0.0, out_mult=8.0):
    """Assert the per-element bound.  `max_out` > 0 admits that fraction of elements up to out_mult x the bound (deep graphs: a flipped
    stored value propagates); nothing may exceed out_mult x."""
    y = np.asarray(y, np.float64); ref = np.asarray(ref, np.float64)
    err = np.abs(y - ref); bnd = elem_bound(ref, k_rel, k_rms)
    bad = err > bnd
    frac = float(bad.mean())
    worst = float((err / np.maximum(bnd, 1e-30)).max())
    assert frac <= max_out and worst <= (out_mult if max_out > 0 else 1.0), \
        "%s: %.3g of the elements beyond the per-element bound (allowed %.3g), worst %.2f x the bound" % (what, frac, max_out, worst)
    return frac, worst
