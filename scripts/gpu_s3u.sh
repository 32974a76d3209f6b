#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "stem7" 2>&1 | tail -2
for v in bases - bases -; do
  if [ "$v" = "-" ]; then export DFMIR_HIP_LIB=$PWD/dfmir_amd/libdfmir_hip.so; else export DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_$v.so; fi
  python scripts/bench_stem7.py 2>&1 | grep fwd
done
