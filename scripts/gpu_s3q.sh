#!/bin/bash
mkdir -p gpurun_out/s3q
for v in - prio1 prio2 prio3 -; do
  if [ "$v" = "-" ]; then export DFMIR_HIP_LIB=$PWD/dfmir_amd/libdfmir_hip.so; else export DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_$v.so; fi
  echo "== $v"; python scripts/bench_conv3d.py 2>&1 | grep "fwd" | cut -c1-52,80-120
  python scripts/bench_3d.py 2>/dev/null | cut -c1-64
done > gpurun_out/s3q/prio.txt
cat gpurun_out/s3q/prio.txt
