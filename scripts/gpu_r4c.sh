#!/bin/bash
# one-off GPU batch of round 4 (wgrad trace, forward-error diagnostic, the two new tests)
mkdir -p gpurun_out/r4c
(python scripts/bench_wgrad2d.py wgrad; DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_trace.so python scripts/bench_wgrad2d.py wgrad) 2>&1 | grep -v "Warn\|amdgpu.ids" > gpurun_out/r4c/wgrad.txt
python scripts/diag_forward_error.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -40 > gpurun_out/r4c/fwd_err.txt
DFMIR_CONV_FP32=1 python scripts/diag_forward_error.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -40 > gpurun_out/r4c/fwd_err_fp32.txt
python -m pytest tests/test_gpu_models.py tests/test_next_rows.py -x -q -k "failed_capture or inference_driver" 2>&1 | tail -5
cat gpurun_out/r4c/wgrad.txt gpurun_out/r4c/fwd_err.txt gpurun_out/r4c/fwd_err_fp32.txt
