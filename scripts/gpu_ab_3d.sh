#!/bin/bash
# A/B of the two 3-D steps between libraries: scripts/gpu_ab_3d.sh <outdir> <tag> ...   ("-" = in-tree)
out=gpurun_out/$1; shift; mkdir -p $out
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = "-" ]; then export DFMIR_HIP_LIB=$PWD/dfmir_amd/libdfmir_hip.so; else export DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_$v.so; fi
  python scripts/bench_3d.py 2>/dev/null | cut -c1-64 | sed "s/^/$v  /" | tee -a $out/ab3d.txt
done; done
