"""Host-side mirror of the reference's operator surface over the C ABI (include/yolosharp_hip.h).

The reference is C# (TorchSharp); there is no dotnet toolchain in the build image, so the host side
is written in Python with the reference's names and argument meaning:

  Ops.non_max_suppression  <- YoloSharp/Utils/Ops.cs:239-371
  Convs.Conv.forward       <- YoloSharp/Modules/Convs.cs:36-62
  Yolov8 / v8DetectionLoss / AMPWrapper.TrainStep  (see model.py)

Everything numeric runs in libyolosharp_hip.so (hand-written HIP, gfx950).  numpy arrays cross the
boundary as plain fp32 host pointers; torch is not involved in the data path.
"""
import ctypes as C
import sys
import numpy as np

from . import _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


_DTYPE_CODES = {"f32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "fp8": 2, "float8": 2, 0: 0, 1: 1, 2: 2}   # same names as model.DTYPES


def _dtype_code(dtype):
    try:
        return _DTYPE_CODES[dtype]
    except (KeyError, TypeError):
        raise ValueError("dtype %r: expected one of f32/float32, bf16/bfloat16, fp8/float8 (or the codes 0, 1, 2)" % (dtype,))


class Engine:
    """One device context (ys_ctx): owns a HIP stream; not thread-safe (callers serialise)."""

    def __init__(self, device=0, lib_path=None, stream=None):
        self.lib = _lib.load(lib_path)
        self.ctx = C.c_void_p()
        if stream is None:
            _lib.check(self.lib, self.lib.ys_ctx_create(device, C.byref(self.ctx)))
        else:
            _lib.check(self.lib, self.lib.ys_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(self.ctx)))
        self.device = device
        self.stream = stream          # the caller's HIP stream the context runs on (None: a stream the library created for itself)

    def close(self):
        if self.ctx:
            self.lib.ys_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        if sys is None or sys.is_finalizing():      # interpreter teardown: the HIP runtime may already be gone and destruction order is arbitrary
            return
        try:
            self.close()
        except Exception:
            pass

    @property
    def is_device_build(self):
        return bool(self.lib.ys_is_device_build())

    def synchronize(self):
        _lib.check(self.lib, self.lib.ys_ctx_synchronize(self.ctx))

    def profile(self, enable=True):
        _lib.check(self.lib, self.lib.ys_ctx_profile_enable(self.ctx, int(enable)))

    def last_ms(self, name):
        ms = C.c_float()
        _lib.check(self.lib, self.lib.ys_ctx_last_ms(self.ctx, name.encode(), C.byref(ms)))
        return ms.value

    def kernel_profile(self, enable=True):
        _lib.check(self.lib, self.lib.ys_ctx_kernel_profile(self.ctx, int(enable)))

    def kernel_profile_read(self, name):
        n, ms = C.c_int32(), C.c_float()
        _lib.check(self.lib, self.lib.ys_ctx_kernel_profile_read(self.ctx, name.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def kernel_profile_dump(self, path):
        _lib.check(self.lib, self.lib.ys_ctx_kernel_profile_dump(self.ctx, str(path).encode()))

    # ---- device memory helpers (bench keeps inputs resident in HBM)
    def malloc(self, nbytes):
        p = C.c_void_p()
        _lib.check(self.lib, self.lib.ys_device_malloc(self.ctx, nbytes, C.byref(p)))
        return p

    def free(self, p):
        _lib.check(self.lib, self.lib.ys_device_free(self.ctx, p))

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(arr.nbytes)
        _lib.check(self.lib, self.lib.ys_memcpy_h2d(self.ctx, p, _ptr(arr), arr.nbytes))
        return p

    def from_device(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        _lib.check(self.lib, self.lib.ys_memcpy_d2h(self.ctx, _ptr(out), p, out.nbytes))
        return out

    # ---- tuning / routing options of the library (process-wide; include/yolosharp_hip.h ys_set_option)
    def set_option(self, key, value):
        _lib.check(self.lib, self.lib.ys_set_option(str(key).encode(), float(value)))

    def unset_option(self, key):
        _lib.check(self.lib, self.lib.ys_unset_option(str(key).encode()))

    def get_option(self, key):
        """The table's entry for `key` (None = not set: the built-in default applies)."""
        v, s = C.c_double(), C.c_int()
        _lib.check(self.lib, self.lib.ys_get_option(str(key).encode(), C.byref(v), C.byref(s)))
        return float(v.value) if s.value else None

    def options(self, **kw):
        """Context manager: set the options for the body, then put back what was there before -- a value set earlier or seeded from the environment at load
        (tests/conftest.py lowers size gates that way), or nothing (tests: `with engine.options(BNRED=0): ...`)."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            old = {k: self.get_option(k) for k in kw}
            for k, v in kw.items():
                self.set_option(k, v)
            try:
                yield self
            finally:
                for k, v in old.items():
                    if v is None:
                        self.unset_option(k)
                    else:
                        self.set_option(k, v)
        return cm()

    # ---- data-parallel exchange through the C ABI (RCCL inside the library; bench.py uses torch.distributed instead)
    def dist_unique_id(self):
        buf = (C.c_ubyte * 128)()
        _lib.check(self.lib, self.lib.ys_dist_unique_id(buf))
        return bytes(buf)

    def dist_init(self, rank, world, unique_id):
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        _lib.check(self.lib, self.lib.ys_dist_init(self.ctx, rank, world, buf))

    def dist_destroy(self):
        _lib.check(self.lib, self.lib.ys_dist_destroy(self.ctx))

    # ---- validation (Detector.cs:103-120): box_iou + match_predictions per image, batched on the device
    def box_iou(self, box1, box2, eps=1e-7):
        """Metrics.box_iou (Metrics.cs:16-34): xyxy [n,4] x [m,4] -> [n,m] fp32."""
        b1 = np.ascontiguousarray(box1, np.float32).reshape(-1, 4)
        b2 = np.ascontiguousarray(box2, np.float32).reshape(-1, 4)
        out = np.zeros((b1.shape[0], b2.shape[0]), np.float32)
        _lib.check(self.lib, self.lib.ys_box_iou(self.ctx, _ptr(b1), b1.shape[0], _ptr(b2), b2.shape[0], eps, 0, _ptr(out)))
        return out

    def kpt_iou(self, kpt1, kpt2, area, eps=1e-7):
        """Metrics.kpt_iou (Metrics.cs:186-212): gt keypoints [n,K,3] x predicted keypoints [m,K,D], area [n] -> OKS [n,m] fp32."""
        k1 = np.ascontiguousarray(kpt1, np.float32)
        k2 = np.ascontiguousarray(kpt2, np.float32)
        ar = np.ascontiguousarray(area, np.float32).reshape(-1)
        n, K = k1.shape[0], k1.shape[1]
        assert k1.shape == (n, K, 3) and k2.ndim == 3 and k2.shape[1] == K and ar.shape[0] == n, (k1.shape, k2.shape, ar.shape)
        out = np.zeros((n, k2.shape[0]), np.float32)
        _lib.check(self.lib, self.lib.ys_kpt_iou(self.ctx, _ptr(k1), n, _ptr(k2), k2.shape[0], _ptr(ar), K, k2.shape[2], eps, 0, _ptr(out)))
        return out

    def val_match(self, rows, count, batch, img_w, img_h):
        """rows [B,max_det,6+extra] / count [B]: the padded NMS outputs (x1,y1,x2,y2,conf,cls,...); batch = collate dict
        (batch_idx, cls, bboxes normalised cxcywh).  Returns a list of bool [count[b], 10] (match_predictions per image)."""
        rows = np.ascontiguousarray(rows, np.float32)
        count = np.ascontiguousarray(count, np.int32)
        B, max_det, stride = rows.shape
        bi = np.ascontiguousarray(np.asarray(batch["batch_idx"], np.float32).reshape(-1))
        cl = np.ascontiguousarray(np.asarray(batch["cls"], np.float32).reshape(-1))
        bb = np.ascontiguousarray(np.asarray(batch["bboxes"], np.float32).reshape(-1, 4))
        cor = np.zeros((B, max_det, 10), np.uint8)
        _lib.check(self.lib, self.lib.ys_val_match_batched(self.ctx, _ptr(rows), _ptr(count), 0, B, max_det, stride, _ptr(bi), _ptr(cl),
                                                           _ptr(bb), bi.shape[0], float(img_w), float(img_h), _ptr(cor)))
        return [cor[b, :count[b]].astype(bool) for b in range(B)]

    def mask_iou(self, gt_ids, nl, pred_masks, eps=1e-7):
        """Metrics.mask_iou (Metrics.cs:120-125) on (gt_ids == k+1), k < nl, vs pred_masks bool/uint8 [n, h, w] -> [nl, n] fp32."""
        ids = np.ascontiguousarray(gt_ids, np.float32).reshape(-1)
        pm = np.ascontiguousarray(np.asarray(pred_masks).astype(np.uint8)).reshape(-1, ids.shape[0]) if len(pred_masks) else np.zeros((0, ids.shape[0]), np.uint8)
        out = np.zeros((nl, pm.shape[0]), np.float32)
        _lib.check(self.lib, self.lib.ys_mask_iou(self.ctx, _ptr(ids), nl, _ptr(pm), pm.shape[0], ids.shape[0], eps, 0, _ptr(out)))
        return out

    def match_predictions(self, pred_classes, true_classes, iou):
        """YoloBaseTaskModel.match_predictions (:377-446): iou [nl, n] -> bool [n, 10]."""
        pc = np.ascontiguousarray(pred_classes, np.float32).reshape(-1)
        tc = np.ascontiguousarray(true_classes, np.float32).reshape(-1)
        iou = np.ascontiguousarray(iou, np.float32).reshape(tc.shape[0], pc.shape[0])
        cor = np.zeros((pc.shape[0], 10), np.uint8)
        _lib.check(self.lib, self.lib.ys_match_predictions(self.ctx, _ptr(pc), pc.shape[0], _ptr(tc), tc.shape[0], _ptr(iou), 0, _ptr(cor)))
        return cor.astype(bool)

    # ---- Augment.LetterBox / Augment.Rectangle (Data/Augment.cs:698-857)
    def letterbox(self, img, resized_width=640, resized_height=640, color=114, rectangle_shape=None):
        """img: uint8 or float32 [C,h,w].  LetterBox(resized_width, resized_height, color) -> (out [C,H,W], pad_l, pad_u); with
        rectangle_shape=(w, h) it is Augment.Rectangle (fit box = the resized shape, canvas = the rectangle shape)."""
        x = np.ascontiguousarray(img)
        is_float = x.dtype != np.uint8
        if is_float:
            x = np.ascontiguousarray(x, np.float32)
        Cc, h, w = x.shape
        ow, oh = rectangle_shape if rectangle_shape is not None else (resized_width, resized_height)
        out = np.empty((Cc, oh, ow), x.dtype)
        pl, pu = C.c_int32(), C.c_int32()
        _lib.check(self.lib, self.lib.ys_letterbox(self.ctx, _ptr(x), int(is_float), 0, Cc, h, w, resized_width, resized_height, ow, oh, int(color),
                                                   _ptr(out), C.byref(pl), C.byref(pu)))
        return out, pl.value, pu.value

    # ---- Ops.process_mask (Ops.cs:462-489)
    def process_mask(self, protos, masks_in, bboxes, shape, upsample=False, cpu_crop_branch=False):
        """protos [nm,mh,mw], masks_in [n,nm], bboxes [n,4] xyxy (image pixels), shape=(ih,iw) -> bool [n,oh,ow]."""
        protos = np.ascontiguousarray(protos, np.float32)
        masks_in = np.ascontiguousarray(masks_in, np.float32).reshape(-1, protos.shape[0])
        bboxes = np.ascontiguousarray(bboxes, np.float32).reshape(-1, 4)
        n, (nm, mh, mw), (ih, iw) = masks_in.shape[0], protos.shape, shape
        assert bboxes.shape[0] == n
        out = np.zeros((n, ih, iw) if upsample else (n, mh, mw), np.uint8)
        _lib.check(self.lib, self.lib.ys_process_mask(self.ctx, _ptr(protos), _ptr(masks_in), _ptr(bboxes), 0, n, nm, mh, mw,
                                                      ih, iw, int(bool(upsample)), int(bool(cpu_crop_branch)), _ptr(out)))
        return out.astype(bool)

    # ---- Ops.non_max_suppression (Ops.cs:239-371)
    def non_max_suppression(self, prediction, conf_thres=0.25, iou_thres=0.45, agnostic=False, max_det=300, nc=0,
                            max_time_img=0.05, max_nms=30000, max_wh=7680, in_place=True, rotated=False,
                            end2end=False):
        """prediction: float32 ndarray [B, 4+nc+extra, A] (xywh, probabilities).  Returns (output, keepi):
        lists of [n_i, 6+extra] float32 rows (x1,y1,x2,y2,conf,cls,extra) and [n_i] int64 anchor indices.
        Like the reference, `prediction[:, 0:4]` is converted to xyxy in place when in_place=True, the
        `agnostic` flag is accepted but ignored (Ops.cs:345), and invalid thresholds raise (YsError status 1).
        rotated=True (Ops.cs:286,349-353): oriented boxes, angle = last channel, boxes stay xywh, Ops.nms_rotated's
        "any earlier box overlaps" rule on Metrics.batch_probiou."""
        if end2end:
            raise NotImplementedError("end2end NMS is outside the hot path (SURVEY.md 8a)")
        pred = prediction if in_place else prediction.copy()
        if pred.dtype != np.float32 or not pred.flags["C_CONTIGUOUS"]:
            raise TypeError("prediction must be a C-contiguous float32 array [B, C, A]")
        B, Cc, A = pred.shape
        ncc = int(nc) if nc else Cc - 4
        extra = Cc - 4 - ncc
        rows = np.zeros((B, max_det, 6 + extra), np.float32)
        keep = np.zeros((B, max_det), np.int64)
        cnt = np.zeros((B,), np.int32)
        fn = self.lib.ys_nms_rotated_batched if rotated else self.lib.ys_nms_batched
        _lib.check(self.lib, fn(self.ctx, _ptr(pred), 0, B, Cc, A, conf_thres, iou_thres, max_det,
                                int(nc), max_nms, max_wh, _ptr(rows), _ptr(keep), _ptr(cnt)))
        output = [rows[b, :cnt[b]].copy() for b in range(B)]
        keepi = [keep[b, :cnt[b]].copy() for b in range(B)]
        return output, keepi

    # ---- Metrics.probiou / batch_probiou (Metrics.cs:137-177, 223-258): oriented boxes xywhr
    def probiou(self, obb1, obb2, CIoU=False, eps=1e-7):
        o1 = np.ascontiguousarray(obb1, np.float32).reshape(-1, 5); o2 = np.ascontiguousarray(obb2, np.float32).reshape(-1, 5)
        if o1.shape != o2.shape:
            raise ValueError("probiou: obb1 and obb2 must have the same shape [N, 5]")
        out = np.zeros((o1.shape[0],), np.float32)
        _lib.check(self.lib, self.lib.ys_probiou(self.ctx, _ptr(o1), _ptr(o2), 0, o1.shape[0], int(bool(CIoU)), eps, _ptr(out)))
        return out

    def batch_probiou(self, obb1, obb2, eps=1e-7):
        o1 = np.ascontiguousarray(obb1, np.float32).reshape(-1, 5); o2 = np.ascontiguousarray(obb2, np.float32).reshape(-1, 5)
        out = np.zeros((o1.shape[0], o2.shape[0]), np.float32)
        _lib.check(self.lib, self.lib.ys_batch_probiou(self.ctx, _ptr(o1), o1.shape[0], _ptr(o2), o2.shape[0], 0, eps, _ptr(out)))
        return out

    def nms_device(self, pred_dev, B, Cc, A, conf_thres, iou_thres, max_det, nc, rows_dev, keep_dev, cnt_dev,
                   max_nms=30000, max_wh=7680):
        """Device-resident variant (no copies, asynchronous): all pointers are HIP device pointers."""
        _lib.check(self.lib, self.lib.ys_nms_batched(self.ctx, pred_dev, 1, B, Cc, A, conf_thres, iou_thres, max_det,
                                                     nc, max_nms, max_wh, rows_dev, keep_dev, cnt_dev))

    # ---- Convs.Conv.forward (Convs.cs:36-62) / plain Conv2d with bias (Head.cs:47-50)
    def conv_bn_act(self, x, weight, k, s, bn=None, bias=None, act=True, training=True, dtype="f32"):
        """x [B,Cin,H,W], weight [Cout,Cin,k,k] float32.  bn = dict(weight, bias, running_mean, running_var)
        (running stats are updated in place when training).  Returns y [B,Cout,Ho,Wo] float32."""
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(weight, np.float32)
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        p = k // 2
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        y = np.empty((B, Cout, Ho, Wo), np.float32)
        dt = _dtype_code(dtype)
        g = b = rm = rv = None
        if bn is not None:
            g = np.ascontiguousarray(bn["weight"], np.float32)
            b = np.ascontiguousarray(bn["bias"], np.float32)
            rm, rv = bn["running_mean"], bn["running_var"]
            assert rm.dtype == np.float32 and rv.dtype == np.float32
        bs = np.ascontiguousarray(bias, np.float32) if bias is not None else None
        _lib.check(self.lib, self.lib.ys_conv_bn_act_fwd(self.ctx, dt, _ptr(x), B, Cin, H, W, _ptr(w), Cout, k, s,
                                                         _ptr(g), _ptr(b), _ptr(rm), _ptr(rv), _ptr(bs), int(act),
                                                         int(training), _ptr(y)))
        return y
