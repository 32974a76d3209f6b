#!/bin/bash
# Knock-out builds of one source file for timing experiments: scripts/build_ko.sh conv3ds W3T_KO 1 2 4 8 ...
# -> build/ko/libdfmir_hip_<macro><n>.so (same C ABI; select with DFMIR_HIP_LIB=...).  build/ is git-ignored but travels
# to the GPU box.
set -e
cd "$(dirname "$0")/../dfmir_amd/csrc"
F=$1; M=$2; shift 2
mkdir -p ../../build/ko
make -s -j8 > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result"
[ "$F" = conv3x3s ] && FLAGS="$FLAGS -Xclang -target-feature -Xclang -packed-fp32-ops"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -D$M=$n -c $F.hip -o ../../build/ko/$F.$M$n.o 2> >(grep -v "is not a recognized feature" | grep -E "error" >&2)
  OBJS=$(ls *.o | grep -v "^$F.o$")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ko/libdfmir_hip_$M$n.so $OBJS ../../build/ko/$F.$M$n.o
  echo built build/ko/libdfmir_hip_$M$n.so
done
