#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -x -q -k "conv2d or split or reflect_conv or resblock or generator_golden or whole_step_golden" 2>&1 | tail -2
bash scripts/gpu_ab_step.sh s3r - base2
