"""Build the native pieces in-tree.

  build_device(): hipcc --offload-arch=gfx950 -> yolosharp_amd/libyolosharp_hip.so   (THE product)
  build_emu():    g++ + tools/hipemu          -> tools/hipemu/libyolosharp_emu.so   (test-only SIMT interpreter
                                                build of the same kernel sources; never loaded by the package)
  build_oracle(): gcc                         -> oracle/libys_oracle.so              (test-only checker)

hipcc cross-compiles for gfx950 without a GPU, so all three run in the CPU-only dev container.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yolosharp_amd", "csrc")
EMU = os.path.join(ROOT, "tools", "hipemu")
BUILD = os.path.join(ROOT, "build")

SOURCES = ["core.hip", "nms.hip", "conv.hip", "conv_gemm.hip", "conv_halo5.hip", "conv_halo4.hip", "conv_halo5m.hip", "conv_halo4m.hip", "conv_stem.hip", "conv_wgrad.hip", "conv_wgrad_gemm.hip", "elementwise.hip", "ops_api.hip", "loss.hip", "attn_dw.hip", "segloss.hip", "poseloss.hip", "valmetrics.hip", "dist.hip", "f8.hip", "model.hip"]
# bit-exact fp32 sections (NMS IoU arithmetic) must not be contracted into FMAs
NO_CONTRACT = {"nms.hip", "valmetrics.hip"}

LIB_DEVICE = os.path.join(ROOT, "yolosharp_amd", "libyolosharp_hip.so")
LIB_EMU = os.path.join(EMU, "libyolosharp_emu.so")
LIB_ORACLE = os.path.join(ROOT, "oracle", "libys_oracle.so")


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(ROOT, "include", "yolosharp_hip.h")]
    hs += [os.path.join(EMU, f) for f in os.listdir(EMU) if f.endswith(".h")]
    return hs


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + "\n")
        raise RuntimeError("build failed: " + " ".join(cmd[:3]))
    return r.stdout


def build_device(verbose=False, variant="dev", defines=(), out=None):
    """variant "dev" is the product.  Triage variants (e.g. build_device(variant="tl", defines=["-DYS_P2_TIMELINE"],
    out="build/libyolosharp_hip_tl.so")) compile the same sources with extra defines into their own object directory."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.join(BUILD, variant), exist_ok=True)
    hdrs = _headers()
    lib = out or LIB_DEVICE
    objs, jobs = [], []
    for s in _sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(BUILD, variant, s + ".o")
        objs.append(obj)
        if _newer(obj, [src] + hdrs):
            # -fno-slp-vectorize: no compiler-formed packed-FP32 (v_pk_*_f32 / v_pk_mov_b32) arithmetic in the kernels.  The staged convolution
            # epilogue's BatchNorm sums were not bit-stable from run to run when hipcc's SLP vectorizer packed them through v_pk_mov_b32
            # shuffles (round-4 bisection, profiles/README.md); single-issue FP32 is also what the CDNA4 guide prices lower next to MFMAs.
            # Measured neutral on the headline step (9.49-9.51 vs 9.48-9.52 ms, same box).
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fno-slp-vectorize",
                   "-fno-strict-aliasing", "-Wno-unused-result", "-I", CSRC, "-c", src, "-o", obj] + list(defines)
            if s in NO_CONTRACT:
                cmd.insert(4, "-ffp-contract=off")
            jobs.append(cmd)
    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(_run, jobs))
    if jobs or _newer(lib, objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    if verbose:
        print("built", lib, "(%d TU recompiled)" % len(jobs))
    return lib


def build_emu(verbose=False):
    os.makedirs(os.path.join(BUILD, "emu"), exist_ok=True)
    hdrs = _headers()
    objs, jobs = [], []
    srcs = [(os.path.join(CSRC, s), s) for s in _sources()] + [(os.path.join(EMU, "hip_emu.cpp"), "hip_emu.cpp")]
    for src, name in srcs:
        obj = os.path.join(BUILD, "emu", name + ".o")
        objs.append(obj)
        if _newer(obj, [src] + hdrs):
            jobs.append(["g++", "-O2", "-std=c++17", "-fPIC", "-fopenmp", "-fvisibility=hidden", "-fno-strict-aliasing", "-DYS_EMU_BUILD",
                         "-Wno-unused-result", "-I", CSRC, "-I", EMU, "-x", "c++", "-c", src, "-o", obj])
    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(_run, jobs))
    if jobs or _newer(LIB_EMU, objs):
        _run(["g++", "-shared", "-fPIC", "-fopenmp", "-o", LIB_EMU] + objs)
    if verbose:
        print("built", LIB_EMU, "(%d TU recompiled)" % len(jobs))
    return LIB_EMU


def build_oracle(verbose=False):
    src = os.path.join(ROOT, "oracle", "nms_ref.c")
    if _newer(LIB_ORACLE, [src]):
        _run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", LIB_ORACLE, src])
    if verbose:
        print("built", LIB_ORACLE)
    return LIB_ORACLE


def build_abi_smoke(verbose=False):
    """tests/c/abi_smoke.c (plain C consumer of include/yolosharp_hip.h) linked against the product library and against the
    test-only interpreter build: build/abi_smoke_dev, build/abi_smoke_emu."""
    src = os.path.join(ROOT, "tests", "c", "abi_smoke.c")
    hdr = os.path.join(ROOT, "include", "yolosharp_hip.h")
    os.makedirs(BUILD, exist_ok=True)
    outs = []
    for tag, lib in (("dev", LIB_DEVICE), ("emu", LIB_EMU)):
        exe = os.path.join(BUILD, "abi_smoke_" + tag)
        if os.path.exists(lib) and _newer(exe, [src, hdr, lib]):
            d, n = os.path.dirname(lib), os.path.basename(lib)[3:-3]
            _run(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                  "-L", d, "-l" + n, "-lm", "-Wl,-rpath,$ORIGIN/" + os.path.relpath(d, BUILD), "-Wl,--allow-shlib-undefined"])
        outs.append(exe)
        if verbose:
            print("built", exe)
    return outs


if __name__ == "__main__":
    what = sys.argv[1:] or ["device", "emu", "oracle"]
    if "device" in what:
        build_device(True)
    if "emu" in what:
        build_emu(True)
    if "oracle" in what:
        build_oracle(True)
    if "timeline" in what:
        build_device(True, variant="tl", defines=["-DYS_P2_TIMELINE"], out=os.path.join(BUILD, "libyolosharp_hip_tl.so"))
    if "epi0" in what:     # A/B: the LDS-staged 16-byte-store epilogue of rounds 1-2 instead of the direct one
        build_device(True, variant="epi0", defines=["-DYS_P2_EPI_DIRECT=0"], out=os.path.join(BUILD, "libyolosharp_hip_epi0.so"))
    if "p2ablate" in what:
        build_device(True, variant="p2abl", defines=["-DYS_P2_ABLATE"], out=os.path.join(BUILD, "libyolosharp_hip_p2abl.so"))
    if "ablate" in what:
        build_device(True, variant="abl", defines=["-DYS_GEMM_ABLATE"], out=os.path.join(BUILD, "libyolosharp_hip_abl.so"))
    if "abi" in what or len(sys.argv) == 1:
        build_abi_smoke(True)
