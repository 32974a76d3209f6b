#!/bin/bash
mkdir -p gpurun_out/s3l
for v in base3d - m16ko1 m16ko2 m16ko3 m16ko4 m16ko7 m16ko8 -; do
  if [ "$v" = "-" ]; then export DFMIR_HIP_LIB=$PWD/dfmir_amd/libdfmir_hip.so; else export DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_$v.so; fi
  echo "== $v"; ONLY=32-16,16-16 python scripts/bench_conv3d.py 2>&1 | grep "fwd" | cut -c1-40
done > gpurun_out/s3l/ko.txt
cat gpurun_out/s3l/ko.txt
