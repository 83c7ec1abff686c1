#!/bin/bash
# triage: ISA of one conv_p2 register tile (seconds): tools/dev/isa_p2.sh <MR> <NR> <out.s> [extra -D...]
M=$1; N=$2; OUT=$3; shift 3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -I yolosharp_amd/csrc --cuda-device-only -S -DYS_P2_ONE -DYS_P2_ONE_M=$M -DYS_P2_ONE_N=$N "$@" yolosharp_amd/csrc/conv.hip -o $OUT 2>&1 | grep -E "error" -A5 | head -20
