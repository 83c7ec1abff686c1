"""The C-ABI library loads and exports every symbol include/yolosharp_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "yolosharp_hip.h")).read()
    return sorted(set(re.findall(r"YS_API\s+[\w\s\*]+?\b(ys_\w+)\s*\(", txt)))


def test_header_declares_symbols():
    syms = declared_symbols()
    assert len(syms) >= 35 and "ys_nms_batched" in syms and "ys_model_forward" in syms


def test_device_library_exports_every_declared_symbol():
    from yolosharp_amd import build
    path = build.build_device()           # hipcc cross-compiles for gfx950 without a GPU
    lib = ctypes.CDLL(path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.ys_is_device_build.restype = ctypes.c_int
    assert lib.ys_is_device_build() == 1


def test_binding_covers_header():
    from yolosharp_amd import _lib
    assert sorted(_lib.PROTOTYPES) == declared_symbols()


def test_no_cpu_fallback_without_device():
    """The product library must fail loudly when no HIP device exists (this container has none)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from yolosharp_amd import Engine, YsError
    with pytest.raises((YsError, OSError)):
        Engine()


def test_missing_library_fails_loudly(tmp_path):
    from yolosharp_amd import _lib
    with pytest.raises(ImportError):
        _lib.load(str(tmp_path / "nope.so"))
