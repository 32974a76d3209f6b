#!/bin/bash
mkdir -p gpurun_out/s3i
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv2d or split or reflect_conv" 2>&1 | tail -2
bash scripts/gpu_ab_step.sh s3i - epi
