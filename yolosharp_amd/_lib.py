"""ctypes binding of the C ABI declared in include/yolosharp_hip.h.

The product library is yolosharp_amd/libyolosharp_hip.so (hipcc, gfx950).  There is NO CPU
fallback: if the library is missing, or no HIP device is present when a context is created,
loading/creation fails loudly.  Tests pass the path of the test-only interpreter build
(tools/hipemu/libyolosharp_emu.so) EXPLICITLY to exercise the kernel sources without a GPU; no environment
variable can redirect the default loader.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libyolosharp_hip.so")

c_float_p = C.POINTER(C.c_float)
c_i64_p = C.POINTER(C.c_int64)
c_i32_p = C.POINTER(C.c_int32)


class ModelDesc(C.Structure):
    _fields_ = [("family", C.c_int32), ("size", C.c_int32), ("task", C.c_int32), ("nc", C.c_int32),
                ("reg_max", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
                ("max_batch", C.c_int32), ("dtype", C.c_int32), ("max_labels", C.c_int32),
                ("kpt_num", C.c_int32), ("kpt_dim", C.c_int32)]


class BlockDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("c1", C.c_int32), ("c2", C.c_int32), ("n", C.c_int32), ("shortcut", C.c_int32),
                ("c3k", C.c_int32), ("e", C.c_float), ("k", C.c_int32), ("s", C.c_int32), ("act", C.c_int32),
                ("height", C.c_int32), ("width", C.c_int32), ("max_batch", C.c_int32), ("dtype", C.c_int32)]


class HeadDesc(C.Structure):
    _fields_ = [("family", C.c_int32), ("task", C.c_int32), ("nc", C.c_int32), ("reg_max", C.c_int32), ("ch", C.c_int32 * 3),
                ("height", C.c_int32), ("width", C.c_int32), ("max_batch", C.c_int32), ("dtype", C.c_int32),
                ("kpt_num", C.c_int32), ("kpt_dim", C.c_int32)]


# name -> (restype, argtypes); must list every symbol of include/yolosharp_hip.h
PROTOTYPES = {
    "ys_last_error": (C.c_char_p, []),
    "ys_version": (C.c_int, []),
    "ys_is_device_build": (C.c_int, []),
    "ys_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ys_ctx_create_on_stream": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "ys_ctx_destroy": (C.c_int, [C.c_void_p]),
    "ys_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "ys_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "ys_model_create": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]),
    "ys_model_destroy": (C.c_int, [C.c_void_p]),
    "ys_model_num_tensors": (C.c_int, [C.c_void_p]),
    "ys_model_tensor_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, c_i32_p, c_i64_p, c_i32_p]),
    "ys_model_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "ys_model_get_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "ys_model_get_grad": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "ys_model_init_weights": (C.c_int, [C.c_void_p, C.c_uint64]),
    "ys_model_set_training": (C.c_int, [C.c_void_p, C.c_int]),
    "ys_model_num_anchors": (C.c_int, [C.c_void_p]),
    "ys_model_num_params": (C.c_int64, [C.c_void_p]),
    "ys_model_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "ys_model_forward_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ys_model_get_output": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "ys_model_pred_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ys_loss_detect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "ys_loss_read": (C.c_int, [C.c_void_p, c_float_p, c_float_p]),
    "ys_loss_segment": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "ys_loss_obb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "ys_loss_pose": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "ys_loss_read_items": (C.c_int, [C.c_void_p, c_float_p, C.c_int, c_float_p]),
    "ys_val_match_batched": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "ys_kpt_iou": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "ys_box_iou": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "ys_process_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ys_model_backward": (C.c_int, [C.c_void_p]),
    "ys_model_backward_segments": (C.c_int, [C.c_void_p]),
    "ys_model_backward_segment": (C.c_int, [C.c_void_p, C.c_int]),
    "ys_model_backward_segment_async": (C.c_int, [C.c_void_p, C.c_int]),
    "ys_model_segment_fence": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ys_model_segment_grad_range": (C.c_int, [C.c_void_p, C.c_int, c_i64_p, c_i64_p]),
    "ys_model_zero_grad": (C.c_int, [C.c_void_p]),
    "ys_model_set_overlap": (C.c_int, [C.c_void_p, C.c_int]),
    "ys_head_create": (C.c_int, [C.c_void_p, C.POINTER(HeadDesc), C.POINTER(C.c_void_p)]),
    "ys_head_forward": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "ys_head_set_grads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ys_head_backward": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "ys_model_grad_buffer": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), c_i64_p]),
    "ys_model_param_buffer": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), c_i64_p]),
    "ys_set_option": (C.c_int, [C.c_char_p, C.c_double]),
    "ys_unset_option": (C.c_int, [C.c_char_p]),
    "ys_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "ys_dist_unique_id": (C.c_int, [C.c_void_p]),
    "ys_dist_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ys_dist_destroy": (C.c_int, [C.c_void_p]),
    "ys_dist_allreduce_grads": (C.c_int, [C.c_void_p, C.c_int]),
    "ys_dist_wait": (C.c_int, [C.c_void_p]),
    "ys_model_backward_allreduce": (C.c_int, [C.c_void_p]),
    "ys_optim_adamw_step": (C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]),
    "ys_nms_batched": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ys_nms_rotated_batched": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ys_probiou": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "ys_batch_probiou": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "ys_conv_bn_act_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ys_conv_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                             C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ys_mask_iou": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "ys_match_predictions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "ys_block_create": (C.c_int, [C.c_void_p, C.POINTER(BlockDesc), C.POINTER(C.c_void_p)]),
    "ys_block_output_shape": (C.c_int, [C.c_void_p, c_i32_p]),
    "ys_model_set_preds": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ys_model_reserve_labels": (C.c_int, [C.c_void_p, C.c_int]),
    "ys_optim_set_param_groups": (C.c_int, [C.c_void_p, C.c_int]),
    "ys_letterbox": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, c_i32_p, c_i32_p]),
    "ys_block_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ys_block_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ys_device_malloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "ys_device_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ys_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ys_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ys_ctx_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "ys_ctx_kernel_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "ys_ctx_kernel_profile_read": (C.c_int, [C.c_void_p, C.c_char_p, c_i32_p, c_float_p]),
    "ys_ctx_kernel_profile_dump": (C.c_int, [C.c_void_p, C.c_char_p]),
    "ys_ctx_last_ms": (C.c_int, [C.c_void_p, C.c_char_p, c_float_p]),
}


class YsError(RuntimeError):
    """Engine error. status 1 (invalid argument) mirrors the reference's ArgumentException."""

    def __init__(self, status, msg):
        super().__init__("yolosharp_hip status %d: %s" % (status, msg))
        self.status = status


_cache = {}


def load(path=None):
    path = path or DEFAULT_LIB   # explicit argument only: no environment override can point the product loader elsewhere
    path = os.path.abspath(path)
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise ImportError(
            "yolosharp_hip: native library %s not found. Build it with `python -m yolosharp_amd.build device` "
            "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback." % path)
    lib = C.CDLL(path)
    missing = []
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise ImportError("yolosharp_hip: %s does not export %s (ABI incomplete; rebuild)" % (path, ", ".join(missing)))
    _cache[path] = lib
    return lib


def check(lib, status):
    if status != 0:
        raise YsError(status, (lib.ys_last_error() or b"").decode("utf-8", "replace"))
