#!/bin/bash
mkdir -p gpurun_out/s2a
python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "full_size_gradients" -s 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -60 > gpurun_out/s2a/grad.txt
cat gpurun_out/s2a/grad.txt
