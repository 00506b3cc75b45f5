#!/bin/bash
# A diagnostic / A-B build of csrc/gemm8.hip (+ gemm8_narrow.hip) linked with the product's other objects:
#   tools/dbg/build_gemm_variant.sh <name> [-DLR_GEMM_OUTOP_UNROLL=8 ...]   ->  tools/dbg/lib_gemm_<name>.so
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$R/llava-reward_amd/csrc/build
n=$1; shift
mkdir -p $B/var_$n
for s in gemm8 gemm8_narrow; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c $R/llava-reward_amd/csrc/$s.hip -o $B/var_$n/$s.o 2>&1 | grep -B2 -A6 "error" || true &
done
wait
objs=$(ls $B/*.o | grep -v "/gemm8.o\|/gemm8_narrow.o" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/dbg/lib_gemm_$n.so $objs $B/var_$n/gemm8.o $B/var_$n/gemm8_narrow.o
echo $R/tools/dbg/lib_gemm_$n.so
