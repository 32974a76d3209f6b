#!/bin/bash
mkdir -p gpurun_out/r4h
python -m pytest tests/test_gpu_ops.py -x -q -k "persistent or conv2d or reflect_conv" 2>&1 | tail -8
python scripts/bench_wgrad2d.py fwd 2>&1 | grep -v "Warn\|amdgpu.ids" > gpurun_out/r4h/fwd.txt
cat gpurun_out/r4h/fwd.txt
