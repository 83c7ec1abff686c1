"""Host-side mirror of the reference's model / loss / step-executor surface over the C ABI.

  Yolov8            <- YoloSharp/Models/Yolo.cs:10-135  (Module<Tensor,(inference,preds)>; train()/eval();
                       state_dict names "model.{i}...."; forward(x NCHW fp32))
  v8DetectionLoss   <- YoloSharp/Utils/Loss.cs:328-484  (forward(preds, batch) -> (loss*B [3], loss_detach [3]))
  AMPWrapper        <- YoloSharp/Utils/Amp.cs:187-286,338-373 (TrainStep / Step / Evaluate)

All arithmetic happens in libyolosharp_hip.so; these classes only marshal plain fp32 arrays and keep the
reference's call order (forward -> criterion -> backward -> optimizer.step -> zero_grad).
"""
import ctypes as C
import sys

import numpy as np

from . import _lib
from .engine import Engine, _ptr

SIZES = {"n": 0, "s": 1, "m": 2, "l": 3, "x": 4}
DTYPES = {"f32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "fp8": 2, "float8": 2}


class Yolov8:
    FAMILY = 8          # ys_family: Models/Yolo.cs:10-135
    TASK = 0            # ys_task: 0 detect, 1 segment, 2 obb, 3 pose
    NM = 0              # extra outputs per anchor after the class scores (Segment: 32 mask coefficients, Obb: 1 angle, Pose: nk)

    def __init__(self, engine: Engine, nc=80, reg_max=16, size="n", height=640, width=640, max_batch=1, dtype="bf16",
                 max_labels=0, kpt_num=17, kpt_dim=3):
        self.engine, self.lib = engine, engine.lib
        self.nc, self.reg_max, self.height, self.width, self.max_batch = nc, reg_max, height, width, max_batch
        self.dtype = dtype
        if self.TASK == 3:
            self.kpt_num, self.kpt_dim, self.NM = kpt_num, kpt_dim, kpt_num * kpt_dim
        desc = _lib.ModelDesc(self.FAMILY, SIZES[size], self.TASK, nc, reg_max, height, width, max_batch, DTYPES[dtype], max_labels,
                              kpt_num if self.TASK == 3 else 0, kpt_dim if self.TASK == 3 else 0)
        self.handle = C.c_void_p()
        _lib.check(self.lib, self.lib.ys_model_create(engine.ctx, C.byref(desc), C.byref(self.handle)))
        self.training = True
        self.A = self.lib.ys_model_num_anchors(self.handle)
        self._batch = 0
        self._info = None

    def close(self):
        if self.handle:
            # a model outlives its context only when the garbage collector finalises a reference cycle in arbitrary order (a failed test keeps both in
            # its traceback): destroying it then would walk freed context state -- the context's own destruction has released the device memory
            if getattr(self.engine, "ctx", None):
                self.lib.ys_model_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        if sys is None or sys.is_finalizing():      # interpreter teardown: the HIP runtime may already be gone and destruction order is arbitrary
            return
        try:
            self.close()
        except Exception:
            pass

    # ---- state_dict surface
    def tensor_info(self):
        if self._info is None:
            info = []
            n = self.lib.ys_model_num_tensors(self.handle)
            name = C.create_string_buffer(256)
            nd, isp = C.c_int32(), C.c_int32()
            shape = (C.c_int64 * 4)()
            for i in range(n):
                _lib.check(self.lib, self.lib.ys_model_tensor_info(self.handle, i, name, 256, C.byref(nd), shape, C.byref(isp)))
                info.append((name.value.decode(), tuple(shape[k] for k in range(nd.value)), bool(isp.value)))
            self._info = info
        return self._info

    def named_parameters(self):
        return [(n, s) for n, s, p in self.tensor_info() if p]

    def num_params(self):
        return int(self.lib.ys_model_num_params(self.handle))

    def state_dict(self):
        out = {}
        for name, shape, _ in self.tensor_info():
            a = np.empty(shape, np.float32)
            _lib.check(self.lib, self.lib.ys_model_get_tensor(self.handle, name.encode(), _ptr(a), a.size))
            out[name] = a
        return out

    def load_state_dict(self, sd, strict=True):
        for name, shape, _ in self.tensor_info():
            if name not in sd:
                if strict:
                    raise KeyError(name)
                continue
            a = np.ascontiguousarray(np.asarray(sd[name], dtype=np.float32).reshape(shape))
            _lib.check(self.lib, self.lib.ys_model_set_tensor(self.handle, name.encode(), _ptr(a), a.size))

    def grads(self):
        out = {}
        for name, shape, is_param in self.tensor_info():
            if not is_param or name.endswith("dfl.conv.weight"):
                continue
            a = np.empty(shape, np.float32)
            _lib.check(self.lib, self.lib.ys_model_get_grad(self.handle, name.encode(), _ptr(a), a.size))
            out[name] = a
        return out

    def init_weights(self, seed=0):
        _lib.check(self.lib, self.lib.ys_model_init_weights(self.handle, seed))

    def train(self, mode=True):
        self.training = bool(mode)
        _lib.check(self.lib, self.lib.ys_model_set_training(self.handle, int(self.training)))
        return self

    def eval(self):
        return self.train(False)

    # ---- forward (Yolo.cs:92-134)
    def forward_device(self, images_dev, batch):
        """images_dev: device pointer to fp32 NCHW [batch,3,H,W] (already resident in HBM). Asynchronous."""
        _lib.check(self.lib, self.lib.ys_model_forward(self.handle, images_dev, 1, batch))
        self._batch = batch

    def forward(self, x, fetch=True):
        """x: float32 ndarray [B,3,H,W].  Returns (inference, preds) like the reference: training -> (None, preds);
        eval -> ({"boxes": [B,4+nc,A]}, preds) with preds = {"boxes": [B,4*reg_max,A], "scores": [B,nc,A]}."""
        x = np.ascontiguousarray(x, np.float32)
        B = x.shape[0]
        assert x.shape == (B, 3, self.height, self.width), x.shape
        _lib.check(self.lib, self.lib.ys_model_forward(self.handle, _ptr(x), 0, B))
        self._batch = B
        if not fetch:
            return None, None
        preds = {"boxes": self.get_output("boxes"), "scores": self.get_output("scores")}
        if self.training:
            return None, preds
        return {"boxes": self.get_output("pred")}, preds

    __call__ = forward

    def forward_u8(self, images_u8):
        """uint8 [B,3,h,w] (0..255, h <= height, w <= width): padded with 114, / 255 and packed on the device (Detector.cs:31-41),
        then the eval / train forward.  Returns like forward()."""
        x = np.ascontiguousarray(images_u8, np.uint8)
        B, c, h, w = x.shape
        assert c == 3
        _lib.check(self.lib, self.lib.ys_model_forward_u8(self.handle, _ptr(x), 0, B, h, w))
        self._batch = B
        preds = {"boxes": self.get_output("boxes"), "scores": self.get_output("scores")}
        return (None, preds) if self.training else ({"boxes": self.get_output("pred")}, preds)

    def get_output(self, key):
        B = self._batch
        C_ = {"boxes": 4 * self.reg_max, "scores": self.nc, "pred": 4 + self.nc + self.NM, "dboxes": 4 * self.reg_max,
              "dscores": self.nc, "mask_coefficient": self.NM, "dmask_coefficient": self.NM, "angle": self.NM, "kpts": self.NM, "dkpts": self.NM, "dangle": self.NM}.get(key)
        if key in ("proto", "dproto"):
            a = np.empty((B, self.NM, self.height // 4, self.width // 4), np.float32)
        else:
            a = np.empty((B, C_, self.A), np.float32)
        _lib.check(self.lib, self.lib.ys_model_get_output(self.handle, key.encode(), _ptr(a), a.size))
        return a

    def set_preds(self, preds):
        """The criterion's `preds` supplied by the caller (Loss.cs:411): {"boxes": [B,4*reg_max,A], "scores": [B,nc,A]} (+
        "mask_coefficient" [B,nm,A], "proto" [B,nm,H/4,W/4] for Segment models) become the head outputs the loss reads."""
        bx = np.ascontiguousarray(preds["boxes"], np.float32)
        sc = np.ascontiguousarray(preds["scores"], np.float32)
        B = bx.shape[0]
        assert bx.shape == (B, 4 * self.reg_max, self.A) and sc.shape == (B, self.nc, self.A), (bx.shape, sc.shape)
        mc = None
        if self.TASK == 2:      # the engine keeps the angle head as logits: invert angle = (sigmoid(z) - 0.25) * pi (Head.cs:429)
            p = np.clip(np.asarray(preds["angle"], np.float64) / np.pi + 0.25, 1e-7, 1 - 1e-7)
            mc = np.ascontiguousarray(np.log(p / (1 - p)), np.float32)
        elif self.TASK in (1, 3):
            mc = np.ascontiguousarray(preds["kpts" if self.TASK == 3 else "mask_coefficient"], np.float32)
        pr = np.ascontiguousarray(preds["proto"], np.float32) if self.TASK == 1 else None
        _lib.check(self.lib, self.lib.ys_model_set_preds(self.handle, B, _ptr(bx), _ptr(sc), _ptr(mc), _ptr(pr)))
        self._batch = B

    def reserve_labels(self, per_image):
        _lib.check(self.lib, self.lib.ys_model_reserve_labels(self.handle, int(per_image)))

    def pred_device(self):
        p = C.c_void_p()
        _lib.check(self.lib, self.lib.ys_model_pred_device(self.handle, C.byref(p)))
        return p

    # ---- backward / optimizer (Amp.cs:338-373)
    def backward(self):
        _lib.check(self.lib, self.lib.ys_model_backward(self.handle))

    def backward_segment(self, seg):
        _lib.check(self.lib, self.lib.ys_model_backward_segment(self.handle, seg))

    def backward_segment_async(self, seg):
        """backward_segment without the wait for the weight-gradient stream at the segment's end (see segment_fence)."""
        _lib.check(self.lib, self.lib.ys_model_backward_segment_async(self.handle, seg))

    def segment_fence(self, seg, stream_ptr):
        """Makes the HIP stream `stream_ptr` (int, e.g. torch.cuda.Stream.cuda_stream) wait until segment `seg`'s gradients are complete."""
        _lib.check(self.lib, self.lib.ys_model_segment_fence(self.handle, seg, C.c_void_p(int(stream_ptr))))

    def num_segments(self):
        return self.lib.ys_model_backward_segments(self.handle)

    def segment_grad_range(self, seg):
        o, c = C.c_int64(), C.c_int64()
        _lib.check(self.lib, self.lib.ys_model_segment_grad_range(self.handle, seg, C.byref(o), C.byref(c)))
        return o.value, c.value

    def backward_allreduce(self):
        """Segmented backward with the RCCL SUM all-reduce of each finished segment overlapped with the next (needs
        Engine.dist_init); ends with the engine stream ordered after the last all-reduce."""
        _lib.check(self.lib, self.lib.ys_model_backward_allreduce(self.handle))

    def set_overlap(self, on=True):
        """Weight-gradient kernels on a second stream (default on); off for per-kernel profiling.  Results are identical."""
        _lib.check(self.lib, self.lib.ys_model_set_overlap(self.handle, int(bool(on))))

    def zero_grad(self):
        _lib.check(self.lib, self.lib.ys_model_zero_grad(self.handle))

    def grad_buffer(self):
        p, n = C.c_void_p(), C.c_int64()
        _lib.check(self.lib, self.lib.ys_model_grad_buffer(self.handle, C.byref(p), C.byref(n)))
        return p, n.value

    def param_buffer(self):
        p, n = C.c_void_p(), C.c_int64()
        _lib.check(self.lib, self.lib.ys_model_param_buffer(self.handle, C.byref(p), C.byref(n)))
        return p, n.value

    def set_param_groups(self, mode="disjoint"):
        """"disjoint" (default) or "reference" = the overlapping groups of YoloBaseTaskModel.cs:144-151 as written (BatchNorm
        parameters are stepped twice per optimizer step from one shared state)."""
        _lib.check(self.lib, self.lib.ys_optim_set_param_groups(self.handle, {"disjoint": 0, "reference": 1}[mode]))

    def adamw_step(self, lrs, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=5e-4):
        arr = (C.c_float * len(lrs))(*lrs)
        _lib.check(self.lib, self.lib.ys_optim_adamw_step(self.handle, arr, len(lrs), beta1, beta2, eps, weight_decay))


class Yolov11(Yolov8):
    """Models/Yolo.cs:200-258: C3k2 / C2PSA graph with Detect(legacy=false)."""
    FAMILY = 11


class _SegmentMixin:
    """Head.Segment (Head.cs:238-324): preds gain "mask_coefficient" [B,32,A] and "proto" [B,32,H/4,W/4]; the eval
    inference dict is {"boxes": [B,4+nc+32,A], "proto": ...} (Head.cs:303-313)."""
    TASK = 1
    NM = 32

    def forward(self, x, fetch=True):
        inf, preds = Yolov8.forward(self, x, fetch)
        if not fetch:
            return inf, preds
        preds["mask_coefficient"] = self.get_output("mask_coefficient")
        preds["proto"] = self.get_output("proto")
        if inf is not None:
            inf["proto"] = preds["proto"]
        return inf, preds

    __call__ = forward


class Yolov8Segment(_SegmentMixin, Yolov8):
    """Models/Yolo.cs:337-352."""


class Yolov11Segment(_SegmentMixin, Yolov11):
    """Models/Yolo.cs:354-370."""


class _ObbMixin:
    """Head.Obb (Head.cs:376-482): preds gain "angle" [B,1,A] = (sigmoid(cv4) - 0.25) * pi; the eval inference tensor is
    [B, 4+nc+1, A] = (xywh of dist2rbox * stride, class probabilities, angle) (Head.cs:411-418), the layout
    Engine.non_max_suppression(rotated=True) reads.  Criterion: v8OBBLoss below."""
    TASK = 2
    NM = 1

    def forward(self, x, fetch=True):
        inf, preds = Yolov8.forward(self, x, fetch)
        if fetch:
            preds["angle"] = self.get_output("angle")
        return inf, preds

    __call__ = forward


class _PoseMixin:
    """Head.Pose (Head.cs:484-606): preds gain the raw "kpts" [B,nk,A]; the eval inference tensor is [B, 4+nc+nk, A] with
    kpts_decode applied (Head.cs:590-605).  Criterion: v8PoseLoss below."""
    TASK = 3
    NM = 51

    def forward(self, x, fetch=True):
        inf, preds = Yolov8.forward(self, x, fetch)
        if fetch:
            preds["kpts"] = self.get_output("kpts")
        return inf, preds

    __call__ = forward


class Yolov8Obb(_ObbMixin, Yolov8):
    """Models/Yolo.cs:406-420."""


class Yolov11Obb(_ObbMixin, Yolov11):
    """Models/Yolo.cs:422-436."""


class Yolov8Pose(_PoseMixin, Yolov8):
    """Models/Yolo.cs:470-484."""


class Yolov11Pose(_PoseMixin, Yolov11):
    """Models/Yolo.cs:486-500."""


class v8DetectionLoss:
    """Loss.cs:328-484.  forward(preds, batch): `preds` is implicit (the model's last training forward stays on the
    device); batch = {"batch_idx": [N], "cls": [N], "bboxes": [N,4] normalised cxcywh} (YoloDataLoader.cs:18-44)."""

    def __init__(self, model: Yolov8):
        self.model, self.lib = model, model.lib

    def forward_device(self, bidx_dev, cls_dev, box_dev, n):
        _lib.check(self.lib, self.lib.ys_loss_detect(self.model.handle, bidx_dev, cls_dev, box_dev, n, 1))

    def forward(self, preds, batch, read=True):
        bi = np.ascontiguousarray(np.asarray(batch["batch_idx"], np.float32).reshape(-1))
        cl = np.ascontiguousarray(np.asarray(batch["cls"], np.float32).reshape(-1))
        bb = np.ascontiguousarray(np.asarray(batch["bboxes"], np.float32).reshape(-1, 4))
        n = bi.shape[0]
        _lib.check(self.lib, self.lib.ys_loss_detect(self.model.handle, _ptr(bi), _ptr(cl), _ptr(bb), n, 0))
        return self.read() if read else None

    __call__ = forward

    def read(self):
        items = (C.c_float * 3)()
        total = C.c_float()
        _lib.check(self.lib, self.lib.ys_loss_read(self.model.handle, items, C.byref(total)))
        loss_detach = np.array(list(items), np.float32)
        return loss_detach * self.model._batch, loss_detach      # (loss * batch_size, loss.detach()), Loss.cs:476


class v8SegmentationLoss(v8DetectionLoss):
    """Loss.cs:688-863.  batch additionally carries "masks" [B, H/4, W/4] (overlap-encoded instance ids,
    YoloDataset.cs:265-267).  Returns (loss*B [5], loss_detach [5]) in the order box, seg, cls, dfl, semseg.
    cpu_crop_branch selects Ops.crop_mask's CPU-only integer branch (Ops.cs:421-435) instead of the broadcast form."""

    def __init__(self, model, cpu_crop_branch=False):
        super().__init__(model)
        self.crop_mode = 1 if cpu_crop_branch else 0

    def forward_device(self, bidx_dev, cls_dev, box_dev, n, masks_dev):
        _lib.check(self.lib, self.lib.ys_loss_segment(self.model.handle, bidx_dev, cls_dev, box_dev, n, masks_dev, 1, self.crop_mode))

    def forward(self, preds, batch, read=True):
        bi = np.ascontiguousarray(np.asarray(batch["batch_idx"], np.float32).reshape(-1))
        cl = np.ascontiguousarray(np.asarray(batch["cls"], np.float32).reshape(-1))
        bb = np.ascontiguousarray(np.asarray(batch["bboxes"], np.float32).reshape(-1, 4))
        mk = np.ascontiguousarray(np.asarray(batch["masks"], np.float32))
        m = self.model
        assert mk.shape == (m._batch, m.height // 4, m.width // 4), mk.shape
        _lib.check(self.lib, self.lib.ys_loss_segment(m.handle, _ptr(bi), _ptr(cl), _ptr(bb), bi.shape[0], _ptr(mk), 0, self.crop_mode))
        return self.read() if read else None

    __call__ = forward

    N_ITEMS = 5

    def read(self):
        items = (C.c_float * self.N_ITEMS)()
        total = C.c_float()
        _lib.check(self.lib, self.lib.ys_loss_read_items(self.model.handle, items, self.N_ITEMS, C.byref(total)))
        loss_detach = np.array(list(items), np.float32)
        return loss_detach * self.model._batch, loss_detach


class v8OBBLoss(v8SegmentationLoss):
    """Loss.cs:486-684.  batch["bboxes"] is [N, 5] = normalised cx, cy, w, h + angle (radians).  Returns (loss*B [4],
    loss_detach [4]) in the order box, cls, dfl, angle."""
    N_ITEMS = 4

    def __init__(self, model):
        v8DetectionLoss.__init__(self, model)

    def forward_device(self, bidx_dev, cls_dev, box_dev, n):
        _lib.check(self.lib, self.lib.ys_loss_obb(self.model.handle, bidx_dev, cls_dev, box_dev, n, 1))

    def forward(self, preds, batch, read=True):
        bi = np.ascontiguousarray(np.asarray(batch["batch_idx"], np.float32).reshape(-1))
        cl = np.ascontiguousarray(np.asarray(batch["cls"], np.float32).reshape(-1))
        bb = np.ascontiguousarray(np.asarray(batch["bboxes"], np.float32).reshape(-1, 5))
        _lib.check(self.lib, self.lib.ys_loss_obb(self.model.handle, _ptr(bi), _ptr(cl), _ptr(bb), bi.shape[0], 0))
        return self.read() if read else None

    __call__ = forward


class v8PoseLoss(v8SegmentationLoss):
    """Loss.cs:870-1071.  batch additionally carries "keypoints" [N, kpt_num, kpt_dim] normalised (x, y[, visibility]).
    Labels must be grouped by image in collate order (batch_idx non-decreasing; the reference's _select_target_keypoints is only
    defined for that order) -- host labels are validated by the library.
    Returns (loss*B [5], loss_detach [5]) in the order box, pose, kobj, cls, dfl."""

    def __init__(self, model):
        v8DetectionLoss.__init__(self, model)

    def forward_device(self, bidx_dev, cls_dev, box_dev, n, kpts_dev):
        _lib.check(self.lib, self.lib.ys_loss_pose(self.model.handle, bidx_dev, cls_dev, box_dev, n, kpts_dev, 1))

    def forward(self, preds, batch, read=True):
        bi = np.ascontiguousarray(np.asarray(batch["batch_idx"], np.float32).reshape(-1))
        cl = np.ascontiguousarray(np.asarray(batch["cls"], np.float32).reshape(-1))
        bb = np.ascontiguousarray(np.asarray(batch["bboxes"], np.float32).reshape(-1, 4))
        m = self.model
        kp = np.ascontiguousarray(np.asarray(batch["keypoints"], np.float32))
        assert kp.shape == (bi.shape[0], m.kpt_num, m.kpt_dim), kp.shape
        _lib.check(self.lib, self.lib.ys_loss_pose(m.handle, _ptr(bi), _ptr(cl), _ptr(bb), bi.shape[0], _ptr(kp), 0))
        return self.read() if read else None

    __call__ = forward


class AMPWrapper:
    """Amp.cs:187-286: TrainStep = forward + criterion + Step (backward, optimizer.step, zero_grad).  The reference's
    half-precision branch never updates weights (SURVEY.md 5 'AMP quirk'); here bf16 keeps fp32 master weights inside
    the engine and the update is applied in every dtype."""

    def __init__(self, model: Yolov8, lr=None, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4, param_groups="disjoint"):
        self.model = model
        if param_groups != "disjoint":
            model.set_param_groups(param_groups)
        lr0 = round(0.002 * 5 / (4 + model.nc), 6) if lr is None else lr     # YoloBaseTaskModel.cs:142
        self.lrs = [lr0, lr0, lr0]                                            # ParamGroups[i].LearningRate
        self.betas, self.eps, self.wd = betas, eps, weight_decay

    def TrainStep(self, images, batch, loss_func: v8DetectionLoss):
        self.model.train()
        self.model.forward(images, fetch=False)
        loss, items = loss_func.forward(None, batch)
        self.Step()
        return loss, items

    def Step(self):
        self.model.backward()
        self.model.adamw_step(self.lrs, self.betas[0], self.betas[1], self.eps, self.wd)
        self.model.zero_grad()

    def Evaluate(self, images):
        self.model.eval()
        return self.model.forward(images)
