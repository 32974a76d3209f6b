#!/bin/bash
# Variant build of one source file for A/B experiments: scripts/build_var.sh conv3x3s p12 "-DW2_PRIO_G1=2 -DW2_TRACE=1"
# -> build/ko/libdfmir_hip_<tag>.so (same C ABI; select with DFMIR_HIP_LIB=...).  build/ is git-ignored but travels
# to the GPU box.
set -e
cd "$(dirname "$0")/../dfmir_amd/csrc"
F=$1; TAG=$2; DEFS=$3
mkdir -p ../../build/ko
make -s -j8 > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result"
[ "$F" = conv3x3s -o "$F" = in_blurdown ] && FLAGS="$FLAGS -Xclang -target-feature -Xclang -packed-fp32-ops"
/opt/rocm/bin/hipcc $FLAGS $DEFS -c $F.hip -o ../../build/ko/$F.$TAG.o 2> >(grep -v "is not a recognized feature" | grep -E "error" >&2)
OBJS=$(ls *.o | grep -v "^$F.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ko/libdfmir_hip_$TAG.so $OBJS ../../build/ko/$F.$TAG.o
echo built build/ko/libdfmir_hip_$TAG.so
