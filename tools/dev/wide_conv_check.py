"""GPU triage: bf16 Conv units at YOLOv8x / v11m widths vs torch fp32 on bf16-rounded operands (forward, dgrad, wgrad)."""
import ctypes as C, sys, os
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolosharp_amd import Engine, _lib
eng = Engine(0)
vp = lambda a: a.ctypes.data_as(C.c_void_p)
cases = [(1, 320, 40, 40, 320, 3, 1), (1, 640, 20, 20, 640, 3, 1), (1, 320, 40, 40, 640, 3, 2), (1, 960, 20, 20, 640, 1, 1),
         (1, 1280, 20, 20, 640, 1, 1), (1, 160, 64, 64, 160, 3, 1), (1, 80, 64, 64, 160, 3, 2), (2, 256, 24, 24, 256, 3, 1),
         (1, 512, 20, 20, 512, 3, 2), (1, 128, 40, 40, 256, 3, 2), (1, 640, 20, 20, 320, 1, 1), (1, 128, 40, 40, 128, 3, 1)]
for (B, Cin, H, W, Cout, k, s) in cases:
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
    y = F.conv2d(x, w, None, stride=s, padding=k // 2)
    bias = np.zeros(Cout, np.float32)
    yy = eng.conv_bn_act(x.numpy(), w.numpy(), k, s, bn=None, bias=bias, act=False, training=True, dtype="bf16")
    e_f = np.abs(yy - y.numpy()).max() / np.abs(y.numpy()).max()
    x.requires_grad_(True); w.requires_grad_(True)
    y = F.conv2d(x, w, None, stride=s, padding=k // 2)
    dy = torch.randn(y.shape, generator=g).bfloat16().float()
    y.backward(dy)
    dx = np.zeros(x.shape, np.float32); dw = np.zeros(w.shape, np.float32)
    xn, wn, dyn = x.detach().numpy().copy(), w.detach().numpy().copy(), dy.numpy().copy()
    _lib.check(eng.lib, eng.lib.ys_conv_bwd(eng.ctx, 1, vp(xn), B, Cin, H, W, vp(wn), Cout, k, s, vp(dyn), vp(dx), vp(dw)))
    e_dx = np.abs(dx - x.grad.numpy()).max() / np.abs(x.grad.numpy()).max()
    e_dw = np.abs(dw - w.grad.numpy()).max() / np.abs(w.grad.numpy()).max()
    print(f"B{B} cin{Cin} {H}x{W} cout{Cout} k{k} s{s}: fwd {e_f:.2e} dx {e_dx:.2e} dw {e_dw:.2e}", flush=True)
