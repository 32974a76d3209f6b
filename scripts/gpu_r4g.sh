#!/bin/bash
mkdir -p gpurun_out/r4g
(for v in 0 1 2 4 8 3 7 9 15; do DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_tko$v.so python scripts/bench_wgrad2d.py wgrad; done) 2>&1 | grep -v "Warn\|amdgpu.ids\|^  run \|iteration lengths\|n=48" > gpurun_out/r4g/wgrad_ko.txt
cat gpurun_out/r4g/wgrad_ko.txt
