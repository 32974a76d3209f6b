#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -x -q -k "conv2d or vxm or resblock or generator_golden or conv_taps" 2>&1 | tail -2
DFMIR_CONV_FP32=1 timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv2d" 2>&1 | tail -2
bash scripts/gpu_ab_step.sh s3v - base3
