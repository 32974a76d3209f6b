#!/bin/bash
mkdir -p gpurun_out/r4k
python -m pytest tests/test_gpu_ops.py -x -q -k "conv2d or conv3x3 or plane_maxima or deferred or reflect_conv" 2>&1 | tail -3
(python scripts/bench_wgrad2d.py wgrad; DFMIR_WGRAD_NO_RING=1 python scripts/bench_wgrad2d.py wgrad; DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_phased.so python scripts/bench_wgrad2d.py wgrad) 2>&1 | grep -v "Warn\|amdgpu.ids" > gpurun_out/r4k/wgrad.txt
cat gpurun_out/r4k/wgrad.txt
