#!/bin/bash
mkdir -p gpurun_out/s3m
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv3d" 2>&1 | tail -3
for v in base3d - base3d -; do
  if [ "$v" = "-" ]; then export DFMIR_HIP_LIB=$PWD/dfmir_amd/libdfmir_hip.so; else export DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_$v.so; fi
  echo "== $v"; ONLY=32-16,16-16 python scripts/bench_conv3d.py 2>&1 | grep "fwd" | cut -c1-48
done > gpurun_out/s3m/m16.txt
cat gpurun_out/s3m/m16.txt
