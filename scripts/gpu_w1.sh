#!/bin/bash
DFMIR_CONV_W1=1 timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv2d or split or reflect_conv" 2>&1 | tail -2
for rep in 1 2; do
python scripts/bench_wgrad2d.py fwd 2>&1 | grep "^fwd"
DFMIR_CONV_W1=1 python scripts/bench_wgrad2d.py fwd 2>&1 | grep "^fwd\|^lib"
done
DFMIR_CONV_W1=1 DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_w1trace.so python scripts/bench_wgrad2d.py fwd 2>&1 | grep "w1 trace\|wave"
