#!/bin/bash
# GPU box: knock-out timings of conv3d_upwgrad_k: scripts/gpu_upwgrad_ko.sh <tag> ...   ("-" = the in-tree library)
mkdir -p gpurun_out
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = "-" ]; then export DFMIR_HIP_LIB=$PWD/dfmir_amd/libdfmir_hip.so; else export DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_uwko$v.so; fi
  TAG=ko$v LEVELS=${LEVELS:-1} timeout 200 python scripts/bench_upwgrad.py 2>/dev/null | tee -a gpurun_out/upwgrad_ko.txt
done; done
