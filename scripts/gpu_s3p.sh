#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -x -q -k "reflect_conv or resblock or conv2d or generator_golden" 2>&1 | tail -2
bash scripts/gpu_ab_step.sh s3p - base2
