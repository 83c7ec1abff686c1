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
os.environ.setdefault("YS_HALO_MIN_FILL", "1")   # its halo-patch form wants feature maps that fill 16 x 16 pixel tiles: the oracle-sized maps of the tests do not
os.environ.setdefault("YS_WGEMM_MIN_M", "1")      # same for its weight-gradient counterpart (csrc/conv_wgrad_gemm.hip)
# Inside a pytest-xdist worker the interpreter's OpenMP team (and torch's) is kept small: eight workers with a full team each oversubscribe the
# cores of the dev container ~8x (the whole `-m "not gpu"` suite: 13 min of wall time for 100 min of CPU time).
if os.environ.get("PYTEST_XDIST_WORKER"):
    os.environ.setdefault("OMP_NUM_THREADS", "2")
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
    if os.environ.get("PYTEST_XDIST_WORKER") or hasattr(config, "workerinput"):
        return None                                 # inside a worker: never start workers of its own
    if "not gpu" not in (getattr(opt, "markexpr", "") or ""):
        return None
    if getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False):
        return None
    n = min(4, max(1, (os.cpu_count() or 1) // 2))   # workers x 2 OpenMP threads each
    if n > 1 and _xdist_workers_start():
        opt.numprocesses = n
    return None


def _xdist_workers_start():
    """pytest-xdist starts its workers through execnet's popen gateway; in some sandboxes that bootstrap dies with a broken pipe and the
    run hangs for minutes before failing.  Probe it in a throw-away process first: if a gateway cannot echo one value within 40 s the suite
    runs in this process instead (about eight minutes)."""
    import subprocess
    import sys
    code = ("import execnet\n"
            "gw = execnet.makegateway('popen')\n"
            "ch = gw.remote_exec('channel.send(channel.receive() + 1)')\n"
            "ch.send(41)\n"
            "assert ch.receive(30) == 42\n"
            "gw.exit()\n")
    try:
        return subprocess.run([sys.executable, "-c", code], stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                              timeout=40).returncode == 0
    except Exception:
        return False


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
