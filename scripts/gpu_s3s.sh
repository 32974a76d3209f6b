#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "reflect_conv_skip" 2>&1 | tail -3
python __graft_entry__.py --smoke 2>&1 | tail -2
