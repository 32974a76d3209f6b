#!/bin/bash
mkdir -p gpurun_out/r4d
python -m pytest tests/test_gpu_ops.py -x -q -k "conv2d or conv3x3 or plane_maxima or deferred or reflect_conv" 2>&1 | tail -5
(python scripts/bench_wgrad2d.py wgrad; DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_chl.so python scripts/bench_wgrad2d.py wgrad; DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_trace.so python scripts/bench_wgrad2d.py wgrad) 2>&1 | grep -v "Warn\|amdgpu.ids" > gpurun_out/r4d/wgrad.txt
DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_cstrace.so python scripts/bench_wgrad2d.py fwd 2>&1 | grep -v "Warn\|amdgpu.ids" > gpurun_out/r4d/fwd_trace.txt
cat gpurun_out/r4d/wgrad.txt gpurun_out/r4d/fwd_trace.txt
