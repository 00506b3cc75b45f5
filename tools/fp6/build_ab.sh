#!/bin/bash
# The FP6 K-loop A/B build of the library (round 6): csrc/{gemm8,engine}.hip compiled with -DLR_FP6_AB=1 (kernel form F8 == 3 of
# gemm_bt8_kernel + the lr_op_gemm_bt_fp6ab entry) and linked with the product's other objects  ->  tools/fp6/lib_fp6ab.so (scratch).
#   tools/fp6/build_ab.sh && python tools/fp6/ab_bench.py
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$R/llava-reward_amd/csrc/build
mkdir -p $B/ab
for s in gemm8 engine; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DLR_FP6_AB=1 "$@" -c $R/llava-reward_amd/csrc/$s.hip -o $B/ab/$s.o &
done
wait
objs=$(ls $B/*.o | grep -v "/gemm8.o\|/engine.o" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/fp6/lib_fp6ab.so $objs $B/ab/gemm8.o $B/ab/engine.o
echo $R/tools/fp6/lib_fp6ab.so
