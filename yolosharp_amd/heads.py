"""Host-side mirror of the reference's head modules as STANDALONE modules over `ys_head_*`.

  Detect   <- YoloSharp/Modules/Head.cs:8-236    Detect(nc, ch, legacy)
  Segment  <- YoloSharp/Modules/Head.cs:238-374  Segment(nc, nm = 32, npr = ch[0], ch, legacy)

`forward([p3, p4, p5])` is `Module<Tensor[], ...>.forward` on fp32 NCHW arrays and returns (inference, preds) like the reference:
training -> (None, preds) with preds = {"boxes" [B, 4*reg_max, A], "scores" [B, nc, A] (+ "mask_coefficient", "proto")};
eval -> ({"boxes": [B, 4+nc(+nm), A]}, preds).  The criteria of yolosharp_amd.model work on the handle unchanged
(v8DetectionLoss(head)(None, batch)), `backward()` then returns [dp3, dp4, dp5]; `backward(dpreds)` takes caller gradients instead.
state_dict names are module-relative ("cv2.0.0.conv.weight", "cv3.2.2.bias", "dfl.conv.weight", "proto.cv1.conv.weight", ...).
"""
import ctypes as C

import numpy as np

from . import _lib
from .engine import Engine, _ptr
from .model import DTYPES, Yolov8


class Detect(Yolov8):
    TASK, NM = 0, 0

    def __init__(self, engine: Engine, nc=80, ch=(64, 128, 256), legacy=True, height=640, width=640, max_batch=1, dtype="f32", reg_max=16):
        self.engine, self.lib = engine, engine.lib
        self.nc, self.reg_max, self.ch = nc, reg_max, tuple(int(c) for c in ch)
        self.height, self.width, self.max_batch, self.dtype = height, width, max_batch, dtype
        desc = _lib.HeadDesc(8 if legacy else 11, self.TASK, nc, reg_max, (C.c_int32 * 3)(*self.ch), height, width, max_batch, DTYPES[dtype], 0, 0)
        self.handle = C.c_void_p()
        _lib.check(self.lib, self.lib.ys_head_create(engine.ctx, C.byref(desc), C.byref(self.handle)))
        self.A = self.lib.ys_model_num_anchors(self.handle)
        self.training, self._batch, self._info = True, 0, None
        self.shapes = [(c, height // s, width // s) for c, s in zip(self.ch, (8, 16, 32))]

    def forward(self, x, fetch=True):
        xs = [np.ascontiguousarray(a, np.float32) for a in x]
        B = xs[0].shape[0]
        for a, shp in zip(xs, self.shapes):
            assert a.shape == (B,) + shp, (a.shape, shp)
        ptrs = (C.c_void_p * 3)(*[a.ctypes.data for a in xs])
        _lib.check(self.lib, self.lib.ys_head_forward(self.handle, ptrs, 0, B))
        self._batch = B
        if not fetch:
            return None, None
        preds = {"boxes": self.get_output("boxes"), "scores": self.get_output("scores")}
        if self.NM:
            preds["mask_coefficient"] = self.get_output("mask_coefficient")
            preds["proto"] = self.get_output("proto")
        if self.training:
            return None, preds
        inf = {"boxes": self.get_output("pred")}
        if self.NM:
            inf["proto"] = preds["proto"]
        return inf, preds

    __call__ = forward

    def backward(self, dpreds=None, need_dx=True):
        """Autograd of the last training forward.  dpreds = None: the gradients the criterion left on the outputs; else a dict with
        the gradients of preds["boxes"], ["scores"] (, ["mask_coefficient"], ["proto"])."""
        if dpreds is not None:
            g = [np.ascontiguousarray(dpreds[k], np.float32) if k in dpreds else None for k in ("boxes", "scores", "mask_coefficient", "proto")]
            _lib.check(self.lib, self.lib.ys_head_set_grads(self.handle, _ptr(g[0]), _ptr(g[1]), _ptr(g[2]) if g[2] is not None else None,
                                                             _ptr(g[3]) if g[3] is not None else None))
        dx = [np.empty((self._batch,) + shp, np.float32) for shp in self.shapes] if need_dx else None
        ptrs = (C.c_void_p * 3)(*[a.ctypes.data for a in dx]) if need_dx else None
        _lib.check(self.lib, self.lib.ys_head_backward(self.handle, 0, ptrs))
        return dx


class Segment(Detect):
    TASK, NM = 1, 32
