"""Host-side mirror of the reference's block modules over the C ABI's per-block entry points.

  Conv        <- YoloSharp/Modules/Convs.cs:36-62     Conv(c1, c2, k, s, act)
  Bottleneck  <- YoloSharp/Modules/Block.cs:572-608   Bottleneck(c1, c2, shortcut, e)
  C2f         <- YoloSharp/Modules/Block.cs:371-399   C2f(c1, c2, n, shortcut)
  C3k2        <- YoloSharp/Modules/Block.cs:623-662   C3k2(c1, c2, n, c3k, e)
  SPPF        <- YoloSharp/Modules/Block.cs:236-285   SPPF(c1, c2)
  C2PSA       <- YoloSharp/Modules/Block.cs:664-810   C2PSA(c1, c2, n)
  Proto       <- YoloSharp/Modules/Block.cs:51-84     Proto(c1, c_, c2)

Each instance owns one `ys_block_create` handle: `forward(x)` is `Module<Tensor,Tensor>.forward` on an fp32 NCHW array,
`backward(dy)` is autograd's backward of that call (returns dx, accumulates the parameter gradients read by `grads()`).
state_dict names are module-relative ("cv1.conv.weight", "m.0.cv2.bn.running_var", ...) in TorchSharp's registration order.
The geometry (H, W, max_batch) is fixed at construction because the engine plans its buffers ahead of time.
"""
import ctypes as C

import numpy as np

from . import _lib
from .engine import Engine, _ptr
from .model import DTYPES, Yolov8

KINDS = {"Conv": 0, "Bottleneck": 1, "C2f": 2, "C3k2": 3, "SPPF": 4, "C2PSA": 5, "Proto": 6}


class _Block(Yolov8):
    KIND = None

    def _create(self, engine: Engine, c1, c2, height, width, max_batch, dtype, n=0, shortcut=False, c3k=False, e=0.0, k=1, s=1,
                act=True):
        self.engine, self.lib = engine, engine.lib
        self.c1, self.c2, self.height, self.width, self.max_batch, self.dtype = c1, c2, height, width, max_batch, dtype
        desc = _lib.BlockDesc(KINDS[self.KIND], c1, c2, n, int(bool(shortcut)), int(bool(c3k)), float(e), k, s, int(bool(act)),
                              height, width, max_batch, DTYPES[dtype])
        self.handle = C.c_void_p()
        _lib.check(self.lib, self.lib.ys_block_create(engine.ctx, C.byref(desc), C.byref(self.handle)))
        self.training, self._batch, self._info = True, 0, None
        self.nc, self.reg_max, self.A = 0, 1, 0
        shp = (C.c_int32 * 3)()
        _lib.check(self.lib, self.lib.ys_block_output_shape(self.handle, shp))
        self.out_shape = tuple(int(v) for v in shp)

    def forward(self, x):
        x = np.ascontiguousarray(x, np.float32)
        B = x.shape[0]
        assert x.shape == (B, self.c1, self.height, self.width), x.shape
        y = np.empty((B,) + self.out_shape, np.float32)
        _lib.check(self.lib, self.lib.ys_block_forward(self.handle, _ptr(x), 0, B, _ptr(y)))
        self._batch = B
        return y

    __call__ = forward

    def backward(self, dy, need_dx=True):
        dy = np.ascontiguousarray(dy, np.float32)
        assert dy.shape == (self._batch,) + self.out_shape, dy.shape
        dx = np.empty((self._batch, self.c1, self.height, self.width), np.float32) if need_dx else None
        _lib.check(self.lib, self.lib.ys_block_backward(self.handle, _ptr(dy), 0, _ptr(dx) if need_dx else None))
        return dx

    def grads(self):
        out = {}
        for name, shape, is_param in self.tensor_info():
            if is_param:
                a = np.empty(shape, np.float32)
                _lib.check(self.lib, self.lib.ys_model_get_grad(self.handle, name.encode(), _ptr(a), a.size))
                out[name] = a
        return out


class Conv(_Block):
    KIND = "Conv"

    def __init__(self, engine, c1, c2, k=1, s=1, act=True, *, height, width, max_batch=1, dtype="bf16"):
        self._create(engine, c1, c2, height, width, max_batch, dtype, k=k, s=s, act=act)


class Bottleneck(_Block):
    KIND = "Bottleneck"

    def __init__(self, engine, c1, c2, shortcut=True, e=0.5, *, height, width, max_batch=1, dtype="bf16"):
        self._create(engine, c1, c2, height, width, max_batch, dtype, shortcut=shortcut, e=e)


class C2f(_Block):
    KIND = "C2f"

    def __init__(self, engine, c1, c2, n=1, shortcut=False, *, height, width, max_batch=1, dtype="bf16"):
        self._create(engine, c1, c2, height, width, max_batch, dtype, n=n, shortcut=shortcut)


class C3k2(_Block):
    KIND = "C3k2"

    def __init__(self, engine, c1, c2, n=1, c3k=False, e=0.5, *, height, width, max_batch=1, dtype="bf16"):
        self._create(engine, c1, c2, height, width, max_batch, dtype, n=n, c3k=c3k, e=e)


class SPPF(_Block):
    KIND = "SPPF"

    def __init__(self, engine, c1, c2, *, height, width, max_batch=1, dtype="bf16"):
        self._create(engine, c1, c2, height, width, max_batch, dtype)


class C2PSA(_Block):
    KIND = "C2PSA"

    def __init__(self, engine, c1, c2, n=1, *, height, width, max_batch=1, dtype="bf16"):
        self._create(engine, c1, c2, height, width, max_batch, dtype, n=n)


class Proto(_Block):
    KIND = "Proto"

    def __init__(self, engine, c1, c_=256, c2=32, *, height, width, max_batch=1, dtype="bf16"):
        self._create(engine, c1, c2, height, width, max_batch, dtype, n=c_)
