#!/bin/bash
# GPU box: one iteration of the march kernel's tuning loop: tests, timings (march on / off), per-phase trace
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "march or conv3d or chain" 2>&1 | tail -4
TAG=march python scripts/bench_march.py 2>/dev/null
TAG=march python scripts/bench_march.py 2>/dev/null
if [ -z "$NO_TILED" ]; then DFMIR_CONV3D_NO_MARCH=1 TAG=tiled python scripts/bench_march.py 2>/dev/null; fi
if [ -f build/ko/libdfmir_hip_m3trace.so ]; then DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_m3trace.so python scripts/march_trace.py 2>&1 | grep -v amdgpu.ids | grep -E 'actg|wave [04]'; fi
} > gpurun_out/march_iter.txt 2>&1
cat gpurun_out/march_iter.txt
