#!/bin/bash
mkdir -p gpurun_out/s3d
(for v in cstrace_old cstrace; do DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_$v.so python scripts/bench_wgrad2d.py fwd; done) 2>&1 | grep -v "Warn\|amdgpu.ids" > gpurun_out/s3d/fwd.txt
cat gpurun_out/s3d/fwd.txt
