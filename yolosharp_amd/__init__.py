"""yolosharp_amd -- MI355X (gfx950) native engine for the YOLO hot path of IntptrMax/YoloSharp.

Only the hot path lives here: csrc/ (hand-written HIP kernels + the C ABI of include/yolosharp_hip.h)
and a thin host-side mirror of the reference's operator interface (engine.py, model.py, dist.py).
There is no CPU or PyTorch fallback: the native library must be built and a HIP device present.
"""
from ._lib import YsError, load  # noqa: F401
from .engine import Engine  # noqa: F401

__all__ = ["Engine", "YsError", "load"]
