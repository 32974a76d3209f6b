#!/bin/bash
# Knock-out builds of the z-marching 3-D conv kernel (timing only): build/ko/libdfmir_hip_m3ko<bits>.so for each argument
# (bits of M3_KO, csrc/conv3dm.hip); the other objects are the in-tree ones.
set -e
cd "$(dirname "$0")/../dfmir_amd/csrc"
mkdir -p ../../build/ko
OBJS=$(ls *.o | grep -v conv3dm.o)
for ko in "$@"; do
  if [ "$ko" = trace ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DM3_TRACE -c conv3dm.hip -o /tmp/conv3dm_trace.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ko/libdfmir_hip_m3trace.so $OBJS /tmp/conv3dm_trace.o
    echo built m3trace; continue
  fi
  case "$ko" in
    [0-9]*) defs="-DM3_KO=$ko";;
    *) defs=$(echo "$ko" | sed 's/^/-D/; s/,/ -D/g');;          # e.g. M3_PRIO=1,M3_ADEPTH=2
  esac
  tag=$(echo "$ko" | tr -d 'M3_=' | tr ',' '_')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $defs -c conv3dm.hip -o /tmp/conv3dm_ko$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ko/libdfmir_hip_m3ko$tag.so $OBJS /tmp/conv3dm_ko$tag.o
  echo built m3ko$tag
done
