"""ISA guard (round-4 advisor finding): the convolution kernels' epilogues compute the BatchNorm statistics with single-issue FP32 VALU.  With hipcc's SLP
vectorizer the sums of some unrolled iterations went through v_pk_mov_b32 shuffles into v_pk_add_f32 / v_pk_fma_f32, and exactly those sums differed from run to run
(profiles/README.md, round 4); the sources are compiled with -fno-slp-vectorize since.  This test disassembles the gfx950 code objects of the product build and fails if a
packed-FP32 instruction reappears in a convolution kernel (a compiler upgrade, another pass forming them, a hand-written v_pk_*_f32) -- no GPU needed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
CONV_TUS = ["conv.hip", "conv_gemm.hip", "conv_halo4.hip", "conv_halo5.hip", "conv_halo4m.hip", "conv_halo5m.hip", "conv_stem.hip"]
PACKED = re.compile(r"\bv_pk_(add|fma|mul)_f32\b|\bv_pk_mov_b32\b")


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
@pytest.mark.parametrize("tu", CONV_TUS)
def test_no_packed_fp32_in_convolution_kernels(tu, tmp_path):
    from yolosharp_amd import build
    build.build_device()                                   # (no-op when the objects are current)
    obj = os.path.join(ROOT, "build", "dev", tu + ".o")
    assert os.path.exists(obj), obj
    local = str(tmp_path / "tu.o")
    shutil.copy(obj, local)
    subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dev = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert dev, "no gfx950 code object in " + tu
    dis = subprocess.run([OBJDUMP, "-d", str(tmp_path / dev[0])], check=True, stdout=subprocess.PIPE, text=True).stdout
    kernel, hits, n_kernels = None, {}, 0
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            kernel = m.group(1); n_kernels += 1
            continue
        if kernel and PACKED.search(line):
            hits[kernel] = hits.get(kernel, 0) + 1
    assert n_kernels > 0
    conv = {k: v for k, v in hits.items() if re.search(r"conv_(p2|gemm|halo|igemm|stem|3x3)|conv3x3", k)}
    assert not conv, "packed-FP32 instructions in convolution kernels: %s" % sorted(conv.items())[:8]
