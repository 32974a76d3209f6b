#!/bin/bash
mkdir -p gpurun_out/r4f
(python scripts/bench_wgrad2d.py wgrad; for v in pc1 pc2 pc3 pc3hl pc2hl tracepc3; do DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_$v.so python scripts/bench_wgrad2d.py wgrad; done) 2>&1 | grep -v "Warn\|amdgpu.ids\|^  run \|iteration lengths" > gpurun_out/r4f/wgrad.txt
cat gpurun_out/r4f/wgrad.txt
