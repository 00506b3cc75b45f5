#!/bin/bash
# The FP6-emulation build of the library (round 6): csrc/{rowops,gemm8,gemm8_narrow}.hip compiled with -DLR_EMU_FP6=1 (common.h
# emu_e2m3: the residuals of the default form snapped to the OCP MX e2m3 / 32-block grid before their e4m3 encoding, weights' twins
# included) and linked with the product's other objects  ->  tools/fp6/lib_emu_fp6.so (scratch: git-ignored, travels with gpurun).
#   tools/fp6/build_emu.sh && LLAVA_REWARD_HIP_LIB=tools/fp6/lib_emu_fp6.so python tools/fp6/emu_probe.py
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$R/llava-reward_amd/csrc/build
mkdir -p $B/emu
for s in rowops gemm8 gemm8_narrow; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DLR_EMU_FP6=1 -c $R/llava-reward_amd/csrc/$s.hip -o $B/emu/$s.o &
done
wait
objs=$(ls $B/*.o | grep -v "/rowops.o\|/gemm8.o\|/gemm8_narrow.o" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/fp6/lib_emu_fp6.so $objs $B/emu/rowops.o $B/emu/gemm8.o $B/emu/gemm8_narrow.o
echo $R/tools/fp6/lib_emu_fp6.so
