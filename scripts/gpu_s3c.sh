#!/bin/bash
mkdir -p gpurun_out/s3c
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv2d or split or reflect_conv or resblock" > gpurun_out/s3c/pytest.txt 2>&1
tail -4 gpurun_out/s3c/pytest.txt
(python scripts/bench_wgrad2d.py fwd) 2>&1 | grep -v "Warn\|amdgpu.ids" > gpurun_out/s3c/fwd.txt
cat gpurun_out/s3c/fwd.txt
