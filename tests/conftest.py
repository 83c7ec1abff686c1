"""Test configuration.

Backends:
  "emu" -- the kernel sources compiled by g++ against tools/hipemu (a TEST-ONLY SIMT interpreter).  Lets the
           `-m "not gpu"` suite execute every kernel against the oracle in the GPU-less dev container.
  "gpu" -- the product library yolosharp_amd/libyolosharp_hip.so on a real MI355X (marked @pytest.mark.gpu).
Only tests/ (and smoke()/bench.py's cpu_baseline leg) may touch oracle/.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The fp8 mode keeps layers below 128 input channels on the bf16 kernel by default (they are HBM-bound: csrc/conv.hip); the tests
# lower the threshold so that the small shapes an oracle can check in seconds go through the fp8 kernel too.  Read once, at the
# first convolution plan of the process.
os.environ.setdefault("YS_F8_MIN_CIN", "32")
os.environ.setdefault("YS_F8_MIN_TAPS", "1")       # and 1x1 layers too (production keeps them in bf16: the quantisation pass costs more than it saves)
# The blocked-GEMM convolution kernel (csrc/conv_gemm.hip) takes layers with >= 128 input channels and >= 1024 output pixels; the
# tests drop the pixel gate so that oracle-sized shapes reach it.
os.environ.setdefault("YS_GEMM_MIN_M", "1")
os.environ.setdefault("YS_WGEMM_MIN_M", "1")      # same for its weight-gradient counterpart (csrc/conv_wgrad_gemm.hip)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The `-m "not gpu"` suite runs every kernel through the test interpreter (minutes per model-level test on one core): spread it over the
    cores with pytest-xdist when it is installed and the caller did not choose a worker count.  The `-m gpu` suite stays in ONE process:
    its tests share the device and some of them time kernels."""
    opt = config.option
    if not hasattr(opt, "numprocesses") or opt.numprocesses is not None or os.environ.get("YS_TEST_SERIAL") == "1":
        return None
    if "not gpu" not in (getattr(opt, "markexpr", "") or ""):
        return None
    if getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False):
        return None
    n = min(8, os.cpu_count() or 1)
    if n > 1:
        opt.numprocesses = n
    return None


BACKENDS = ["emu", pytest.param("gpu", marks=pytest.mark.gpu)]


@pytest.fixture(scope="session")
def emu_lib_path():
    from yolosharp_amd import build
    return build.build_emu()


@pytest.fixture(scope="session")
def oracle_lib():
    import ctypes
    from yolosharp_amd import build
    return ctypes.CDLL(build.build_oracle())


_engines = {}


@pytest.fixture
def engine(request, emu_lib_path):
    """Engine for the backend named by the test's `backend` parameter."""
    backend = request.getfixturevalue("backend")
    if backend not in _engines:
        from yolosharp_amd import Engine
        if backend == "emu":
            _engines[backend] = Engine(lib_path=emu_lib_path)
        else:
            eng = Engine()  # product library; raises loudly if missing or no HIP device
            assert eng.is_device_build, "GPU tests must run the hipcc-built library"
            _engines[backend] = eng
    return _engines[backend]
