#!/bin/bash
mkdir -p gpurun_out/s3o
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "warp" 2>&1 | tail -3
for v in basew - basew -; do
  if [ "$v" = "-" ]; then export DFMIR_HIP_LIB=$PWD/dfmir_amd/libdfmir_hip.so; else export DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_$v.so; fi
  python scripts/bench_warp_roofline.py 2>&1 | grep SUMMARY
done | tee gpurun_out/s3o/warp.txt
