#!/bin/bash
# GPU box: knock-out timings of conv3d_march_k: scripts/gpu_march_ko.sh <ko bits> ...   ("-" = the in-tree library)
mkdir -p gpurun_out
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = "-" ]; then export DFMIR_HIP_LIB=$PWD/dfmir_amd/libdfmir_hip.so; else export DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_m3ko$v.so; fi
  TAG=ko$v python scripts/bench_march.py 2>/dev/null | tee -a gpurun_out/march_ko.txt
done; done
