#!/bin/bash
# Knock-out builds of the parity-class weight-gradient kernel (timing only): build/ko/libdfmir_hip_uwko<bits>.so for each
# argument (bits of UW_KO, csrc/conv3duw.hip, or a list of -D definitions NAME=V,NAME=V); the other objects are the in-tree ones.
set -e
cd "$(dirname "$0")/../dfmir_amd/csrc"
mkdir -p ../../build/ko
OBJS=$(ls *.o | grep -v conv3duw.o)
for ko in "$@"; do
  case "$ko" in
    [0-9]*) defs="-DUW_KO=$ko"; tag=$ko;;
    *) defs=$(echo "$ko" | sed 's/^/-D/; s/,/ -D/g'); tag=$(echo "$ko" | sed 's/UW_//g; s/=//g; s/,/_/g');;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $defs -c conv3duw.hip -o /tmp/conv3duw_ko$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ko/libdfmir_hip_uwko$tag.so $OBJS /tmp/conv3duw_ko$tag.o
  echo built uwko$tag
done
