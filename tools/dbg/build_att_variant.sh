#!/bin/bash
# A diagnostic / A-B build of csrc/attention.hip linked with the product's other objects:
#   tools/dbg/build_att_variant.sh <name> [-DLR_ATT_PIPE=3 -DLR_ATT_DIAG=16 ...]   ->  tools/dbg/lib_att_<name>.so
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$R/llava-reward_amd/csrc/build
n=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c $R/llava-reward_amd/csrc/attention.hip -o $B/attention_$n.o 2>&1 | grep -v "occupancy\|warnings generated" || true
objs=$(ls $B/*.o | grep -v "attention" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/dbg/lib_att_$n.so $objs $B/attention_$n.o
echo $R/tools/dbg/lib_att_$n.so
