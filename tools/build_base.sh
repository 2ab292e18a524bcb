#!/bin/bash
# Builds yolo_amd/libyolo_amd_base.so from the conv.hip of a git revision (default HEAD)
# and the other sources of the working tree, for A/B timing against the working tree's
# library:   YOLO_AMD_LIB=$PWD/yolo_amd/libyolo_amd_base.so python tools/conv_bench.py ...
set -e
REV=${1:-HEAD}
cd "$(dirname "$0")/../yolo_amd/csrc"
mkdir -p build/base
git show "$REV:yolo_amd/csrc/conv.hip" > build/base/conv.hip
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -I."
/opt/rocm/bin/hipcc $FLAGS -x hip -c build/base/conv.hip -o build/base/conv.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyolo_amd_base.so build/abi.cpp.o build/nms.hip.o \
    build/decode.hip.o build/base/conv.o build/bn.hip.o build/post.hip.o build/pre.hip.o
echo built ../libyolo_amd_base.so from $REV
