#!/bin/bash
mkdir -p gpurun_out/r4l
export DFMIR_WGRAD_NO_RING=1
(python scripts/bench_wgrad2d.py wgrad; for v in alt ko1 ko6 ko7 ko7alt ko7eq; do DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_$v.so python scripts/bench_wgrad2d.py wgrad; done; python scripts/bench_wgrad2d.py wgrad) 2>&1 | grep -v "Warn\|amdgpu.ids" > gpurun_out/r4l/wgrad.txt
cat gpurun_out/r4l/wgrad.txt
